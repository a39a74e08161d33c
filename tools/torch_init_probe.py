import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from strikeforce_amd import config, env
mode = sys.argv[1]
def mk(n=8):
    w = config.baseline_workload("C1", arenas=n); g = env.ArenaBatch(w); tb, sr = w.seeds(); g.reset(tb, sr); return w, g
if mode == "observe":
    w, g = mk(); g.observe()
elif mode == "subproc":
    subprocess.check_call([sys.executable, "-c", "import sys; sys.path.insert(0,'.'); from strikeforce_amd import config, env; w=config.baseline_workload('C1',arenas=2); g=env.ArenaBatch(w)"])
    w, g = mk()
elif mode == "plain":
    w, g = mk()
elif mode == "destroy":
    w, g = mk(); g.close(); w, g = mk()
elif mode == "gxx":
    subprocess.check_call(["g++", "--version"], stdout=subprocess.DEVNULL)
    w, g = mk()
import torch
try:
    torch.cuda.init(); print(mode, "torch init OK", torch.cuda.device_count())
except Exception as e:
    print(mode, "torch init FAILED:", e)
