"""Times the MFMA GEMM kernel alone on the shapes of the network's layers (dense A, no im2col gather)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from strikeforce_amd import policy

pb = policy.PolicyBatch(policy.init_parameters(0), 4)
gemm = pb.gemm_split if os.environ.get("GEMM_SPLIT") == "1" else pb.gemm  # GEMM_SPLIT=1: the bf16-split kernel
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for name, M, N, K in [("conv0-shape", B * 225, 160, 288), ("conv1-shape", B * 49, 160, 1440), ("conv2-shape", B * 9, 160, 1440),
                      ("conv3-shape", B, 160, 1440), ("gru", B, 480, 160), ("lin", B, 160, 160)]:
    if os.environ.get("GEMM_ONLY") and os.environ["GEMM_ONLY"] != name:
        continue
    if os.environ.get("GEMM_DATA") == "const":
        a = torch.full((M, K), 0.5, device="cuda")
        w = torch.full((N, K), 0.25, device="cuda")
    elif os.environ.get("GEMM_DATA") == "zero":
        a = torch.zeros((M, K), device="cuda")
        w = torch.zeros((N, K), device="cuda")
    else:
        a = torch.randn((M, K), device="cuda")
        w = torch.randn((N, K), device="cuda")
    c = torch.zeros((M, N), device="cuda")
    torch.cuda.synchronize()
    for _ in range(2):
        gemm(a.data_ptr(), K, w.data_ptr(), None, c.data_ptr(), N, M, N, K)
    pb.kernel_time(True)
    for _ in range(5):
        gemm(a.data_ptr(), K, w.data_ptr(), None, c.data_ptr(), N, M, N, K)
    ms, fl, n = pb.kernel_time(False)
    print(json.dumps({"shape": name, "M": M, "N": N, "K": K, "us": ms / n * 1e3, "tflops": fl / (ms * 1e-3) / 1e12}))
    del a, w, c
