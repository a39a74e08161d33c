"""What a plain fill of the dense observation buffer (4096 x 123 008 B) costs on this card: the practical ceiling of k_observe's write."""
import torch

n = 4096 * 30752
x = torch.empty(n, dtype=torch.float32, device="cuda")
y = torch.empty(n, dtype=torch.float32, device="cuda")
for name, fn in (("zero_", lambda: x.zero_()), ("fill_(1)", lambda: x.fill_(1.0)), ("copy_", lambda: x.copy_(y))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(20):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 20
    print("%s: %.4f ms for %.0f MB written = %.2f TB/s" % (name, ms, n * 4 / 1e6, n * 4 / ms / 1e9))
