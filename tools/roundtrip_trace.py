"""One small request through the copy-free host path, 300 times, for a rocprofv3 timeline (kernel, memory-copy and HIP API
traces): where the ~20 us around the leaf kernel go.
    rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d gpurun_out/rt -- python3 tools/roundtrip_trace.py
then tools/roundtrip_timeline.py gpurun_out/rt"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from bokego_amd.bkw import load_bkw
from bokego_amd.engine import LeafEngine
from bokego_amd.workload import make_batch
g = os.path.join(os.getcwd(), "tests", "golden")
eng = LeafEngine(load_bkw(os.path.join(g, "policy_19.bkw")), load_bkw(os.path.join(g, "value_synth.bkw")), max_batch=256)
x8, recs = make_batch(64, seed_base=3, dtype=np.uint8, with_records=True)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 63
for _ in range(300):
    eng.wait(eng.submit_positions(recs[:B], probs=True, value=True, n_policy=1))
eng.close()
