import json, os, sys
sys.path.insert(0, "/root/repo")
import torch
from bokego_amd import match, nnet
from bokego_amd.bkw import load_bkw
from bokego_amd.gtp import NativeGTP
from bokego_amd.mcts_native import Position
g = "/root/repo/tests/golden"
pi, val = nnet.HipPolicyNet(load_bkw(g + "/policy_19.bkw")), nnet.HipValueNet(load_bkw(g + "/value_synth.bkw"))
for rollouts in ((400,) if "--sixteen" in sys.argv else (400, 1600)):
    for leaves in ((16,) if "--sixteen" in sys.argv else (4, 8)):
        for vo in (0, 1):
            a = match.InProcessEngine(NativeGTP(Position(), pi, val, no_sim=True, time_lim=None, n_rollouts=rollouts, leaves=leaves, leaves_visit_only=vo), name="multi")
            b = match.InProcessEngine(NativeGTP(Position(), pi, val, no_sim=True, time_lim=None, n_rollouts=rollouts), name="one_leaf")
            res = match.play_match(a, b, 100, 5.5, None, opening_plies=4, seed=60_000)
            print(json.dumps({"rollouts": rollouts, "leaves": leaves, "visit_only": vo, "multi_wins": res["multi_wins"], "ms": res["ms_per_move"]}), flush=True)
