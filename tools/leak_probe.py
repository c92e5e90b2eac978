"""Engine / pool life cycles in one process: device memory, host RSS and open file descriptors must come back to where they were.
    python tools/leak_probe.py [cycles]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bokego_amd import selfplay  # noqa: E402
from bokego_amd.bkw import load_bkw  # noqa: E402
from bokego_amd.engine import LeafEngine  # noqa: E402

g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
pw, vw = load_bkw(os.path.join(g, "policy_19.bkw")), load_bkw(os.path.join(g, "value_synth.bkw"))
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 60
x = np.random.default_rng(0).integers(0, 2, size=(300, 27, 9, 9)).astype(np.uint8)


def state():
    free, _ = torch.cuda.mem_get_info(0)
    rss = int(open("/proc/self/statm").read().split()[1]) * os.sysconf("SC_PAGE_SIZE")
    return free, rss, len(os.listdir("/proc/self/fd")), len(os.listdir("/proc/self/task"))


def cycle(i):
    eng = LeafEngine(pw, vw, device_id=0, max_batch=2048, precision="f16x2" if i % 2 else "f32")
    eng.eval(x[: 1 + i % 300])
    t = [eng.submit(x[:40], n_policy=1) for _ in range(3)]
    for k in t:
        eng.wait(k)
    xd = torch.from_numpy(x[:64]).cuda()
    eng.eval_device(xd)
    ev = selfplay.EngineEvaluator(eng)
    selfplay.self_play(ev, n_games=8, rollouts=30, max_turns=12, threads=2, n_pools=2)
    torch.cuda.synchronize()
    eng.close()


for i in range(5):
    cycle(i)
s0 = state()
for i in range(cycles):
    cycle(i)
s1 = state()
print(f"{cycles} cycles: device free {s0[0] / 2**20:.0f} -> {s1[0] / 2**20:.0f} MiB, host RSS {s0[1] / 2**20:.0f} -> {s1[1] / 2**20:.0f} MiB, fds {s0[2]} -> {s1[2]}, threads {s0[3]} -> {s1[3]}")
ok = s0[0] - s1[0] < 64 << 20 and s1[1] - s0[1] < 256 << 20 and s1[2] - s0[2] < 8 and s1[3] - s0[3] < 4
print("ok" if ok else "LEAK")
sys.exit(0 if ok else 1)
