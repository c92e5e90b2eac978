"""The GPU feature encoder (bk_encode.hip) against the host encoder (bk_pos_features_u8, itself fuzzed against the reference's
features(): tools/fuzz_rules_vs_reference.py) on positions of random games -- captures, kos, passes, stale liberty caches included.
    python tools/fuzz_encoder.py [seed] [positions]"""
import ctypes
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bokego_amd import go  # noqa: E402
from bokego_amd.bkw import load_bkw  # noqa: E402
from bokego_amd.engine import LeafEngine  # noqa: E402

g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
eng = LeafEngine(load_bkw(os.path.join(g, "policy_19.bkw")), load_bkw(os.path.join(g, "value_synth.bkw")), max_batch=8192)
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
want_n = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
lib = go.golib()
done = games = 0
t0 = time.time()
while done < want_n:
    recs, planes = [], []
    while len(recs) < 8192:
        gm = go.Game(moves=[])
        games += 1
        for ply in range(rng.randint(1, 95)):
            legal = gm.get_legal_moves()
            if rng.random() < 0.03 or not legal:
                gm.play_pass()
            else:
                gm.play_move(rng.choice(sorted(legal)))
            if rng.random() < 0.5:
                # the history-dependent half first (what bk_pool_collect_pos does), then the record and the host planes of that record
                planes.append(gm.features_u8().copy())
                recs.append(np.frombuffer(bytes(gm._pos), np.uint8).copy())
                if len(recs) == 8192:
                    break
    r = np.stack(recs)
    got = eng.encode_positions(r)
    want = np.stack(planes)
    assert np.array_equal(got, want), np.argwhere(got != want)[:5]
    done += len(recs)
    print(f"{done} positions from {games} games: GPU planes equal the host's ({time.time() - t0:.0f} s)", flush=True)
eng.close()
