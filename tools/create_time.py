import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from bokego_amd.bkw import load_bkw
from bokego_amd.engine import LeafEngine
from bokego_amd import nnet
g = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests", "golden")
pw, vw = load_bkw(os.path.join(g, "policy_19.bkw")), load_bkw(os.path.join(g, "value_synth.bkw"))
torch.cuda.init()
for mb in (1024, 8192):
    for _ in range(3):
        t = time.perf_counter(); e = LeafEngine(pw, vw, device_id=0, max_batch=mb); t1 = time.perf_counter() - t
        t = time.perf_counter(); e.close(); t2 = time.perf_counter() - t
        print(f"max_batch {mb}: create {t1*1e3:.1f} ms, close {t2*1e3:.1f} ms")
pi = nnet.HipPolicyNet(pw)
x = np.zeros((1, 27, 9, 9), np.float32)
pi(torch.from_numpy(x))
for _ in range(3):
    t = time.perf_counter(); pi.load_state_dict(pw); pi(torch.from_numpy(x)); print(f"HipPolicyNet.load_state_dict + first call: {(time.perf_counter()-t)*1e3:.1f} ms")
