"""Launch the GPU feature encoder 60 times on 4096 self-play-like records (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from bokego_amd.bkw import load_bkw
from bokego_amd.engine import LeafEngine
from bokego_amd.workload import make_batch
g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
eng = LeafEngine(load_bkw(os.path.join(g, "policy_19.bkw")), load_bkw(os.path.join(g, "value_synth.bkw")), max_batch=4096)
_, recs = make_batch(4096, dtype=np.uint8, with_records=True)
for _ in range(60):
    eng.encode_positions(recs)
