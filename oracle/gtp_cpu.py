"""oracle/gtp_cpu.py -- TEST / BASELINE INFRASTRUCTURE, NOT PRODUCT CODE.

The CPU-backend engine of BASELINE configs[4] ("... win-rate and ms/move vs CPU baseline"): the same GTP front-end and
the same native PUCT search as `python -m bokego_amd.gtp`, but every leaf is evaluated by oracle/torch_ref.py -- the
reference's torch operators on the host's cores -- instead of the HIP engine.  Used as a subprocess opponent:

    python -m bokego_amd.match --games 100 -r 1600 --opponent "python -m oracle.gtp_cpu -r 1600 --threads 16"
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-r", type=int, default=1600)
    ap.add_argument("--threads", type=int, default=0,
                    help="torch CPU threads (0: the cgroup CPU quota minus one for the match runner, else torch's default; "
                         "torch's own default is every visible core -- 256 on a GPU box whose share is 16, i.e. 10x slower)")
    args = ap.parse_args(argv)
    import torch
    threads = args.threads
    if threads <= 0:
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            if q != "max":
                threads = max(1, int(int(q) / int(per)) - 1)
        except (OSError, ValueError):
            pass
    if threads > 0:
        torch.set_num_threads(threads)
    from bokego_amd.bkw import load_bkw
    from bokego_amd.gtp import NativeGTP
    from bokego_amd.mcts_native import Position
    from oracle.torch_ref import TorchPolicy, TorchValue
    g = os.path.join(REPO, "tests", "golden")
    P, V = TorchPolicy(load_bkw(os.path.join(g, "policy_19.bkw"))), TorchValue(load_bkw(os.path.join(g, "value_synth.bkw")))

    class Net:   # the duck-typed net interface MCTS expects (mcts.py:54-76)
        def __init__(self, fn, value=False):
            self.fn, self.value = fn, value

        def to(self, d):
            return self

        def __call__(self, x):
            o = self.fn(x)
            return o.reshape(-1, 1) if self.value else o

    gtp = NativeGTP(Position(), Net(P), Net(V, True), no_sim=True, time_lim=None, n_rollouts=args.r)
    gtp.start()


if __name__ == "__main__":
    main()
