"""Rate of the fp32 (or f16x2) leaf kernel at B = 4,096 for experiment builds whose results are wrong on purpose
(make -C bokego_amd/csrc exp EXP=1|2|3: no LDS reads / no weight loads in the conv loops), next to the shipped library:
    BK_LIB_PATH=bokego_amd/libbokego_amd_exp1.so python tools/exp_rate.py [f32|f16x2]
What an experiment build gains is the most that removing that traffic's cost (e.g. LDS bank conflicts) could gain."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bokego_amd.bkw import load_bkw  # noqa: E402
from bokego_amd.engine import LeafEngine  # noqa: E402

g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
eng = LeafEngine(load_bkw(os.path.join(g, "policy_19.bkw")), load_bkw(os.path.join(g, "value_synth.bkw")), max_batch=4096, precision=prec)
x = torch.from_numpy(np.random.default_rng(0).integers(0, 2, size=(4096, 27, 9, 9)).astype(np.float32)).cuda()
for _ in range(5):
    eng.eval_device(x, logits=True, probs=True, value=True)
torch.cuda.synchronize()
eng.set_profiling(True)
s0 = eng.stats()
for _ in range(40):
    eng.eval_device(x, logits=True, probs=True, value=True)
torch.cuda.synchronize()
s1 = eng.stats()
ms = (s1["kernel_ms_sum"] - s0["kernel_ms_sum"]) / (s1["kernel_ms_count"] - s0["kernel_ms_count"])
print(f"{os.environ.get('BK_LIB_PATH', 'shipped library')} {prec}: {ms:.4f} ms per step = {4096 / ms * 1e3:,.0f} leaf-evals/s")
