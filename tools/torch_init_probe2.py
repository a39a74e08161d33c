import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
if mode in ("import_first", "avail_first"):
    import torch
    if mode == "avail_first":
        torch.cuda.is_available()
from strikeforce_amd import config, env
w = config.baseline_workload("C1", arenas=8); g = env.ArenaBatch(w); tb, sr = w.seeds(); g.reset(tb, sr)
print(mode, "ArenaBatch OK")
import torch
try:
    torch.cuda.init(); x = torch.zeros(4, device="cuda"); print(mode, "torch init OK")
except Exception as e:
    print(mode, "torch init FAILED:", e)
os.system("grep -E 'libamdhip64|libhsa-runtime' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
