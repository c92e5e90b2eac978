"""How many CPUs a self-play generation really occupies (process CPU time / wall, threads by name), against the cgroup's quota.
    python tools/cpu_use_probe.py [precision] [threads]"""
import os
import resource
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402,F401
from bokego_amd import selfplay  # noqa: E402
from bokego_amd.bkw import load_bkw  # noqa: E402
from bokego_amd.engine import LeafEngine  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x2"
threads = int(sys.argv[2]) if len(sys.argv) > 2 else None
g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
eng = LeafEngine(load_bkw(os.path.join(g, "policy_19.bkw")), load_bkw(os.path.join(g, "value_synth.bkw")), device_id=0, max_batch=8192, precision=prec)
ev = selfplay.EngineEvaluator(eng)
selfplay.self_play(ev, n_games=64, rollouts=100, threads=threads)          # warm: threads exist


def per_thread():
    out = {}
    for t in os.listdir("/proc/self/task"):
        try:
            f = open(f"/proc/self/task/{t}/stat").read().rsplit(")", 1)[1].split()
            out[t] = (int(f[11]) + int(f[12])) / os.sysconf("SC_CLK_TCK")
        except OSError:
            pass
    return out


def throttled():
    try:
        return {k: int(v) for k, v in (ln.split() for ln in open("/sys/fs/cgroup/cpu.stat")) if k in ("nr_throttled", "throttled_usec")}
    except OSError:
        return {}


for rep in range(4):
    r0, t0, p0, th0 = resource.getrusage(resource.RUSAGE_SELF), time.perf_counter(), per_thread(), throttled()
    local, total = selfplay.self_play(ev, n_games=512, rollouts=400, threads=threads)
    r1, t1, p1, th1 = resource.getrusage(resource.RUSAGE_SELF), time.perf_counter(), per_thread(), throttled()
    cpu = (r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime)
    busy = sorted(((p1[k] - p0.get(k, 0.0)) for k in p1), reverse=True)
    print(f"{prec} threads {threads}: generation {local['seconds']:.3f} s (wall {t1 - t0:.3f}), CPU {cpu:.2f} s = {cpu / (t1 - t0):.1f} CPUs; "
          f"threads with > 5 % of the wall: {sum(b > 0.05 * (t1 - t0) for b in busy)} of {len(p1)}; busiest {[round(b / (t1 - t0), 2) for b in busy[:16]]}; "
          f"throttled periods +{th1.get('nr_throttled', 0) - th0.get('nr_throttled', 0)}", flush=True)
