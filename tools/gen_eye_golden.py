#!/usr/bin/env python3
"""tests/golden/possible_eye.npz: the REFERENCE's go.possible_eye (go.py:470-485, with its DIAGONALS table, go.py:372-373)
on every point of (a) the 536 golden positions, (b) the 256 random-playout finals and (c) 600 seeded random dense boards
(where one-point eyes, false eyes and the table's blind corner all occur).  Data only: boards in, colours out.

    python tools/gen_eye_golden.py        # needs the reference checkout (BOKEGO_REFERENCE, default /root/reference)
"""
import json
import os
import random
import sys

REF = os.environ.get("BOKEGO_REFERENCE", "/root/reference")
if not os.path.isdir(os.path.join(REF, "bokego")):
    sys.exit(f"reference checkout not found at {REF}; set BOKEGO_REFERENCE")
random.seed(0)
sys.path.insert(0, REF)
import numpy as np  # noqa: E402

import bokego.go as go  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
boards = [p["board"] for p in json.load(open(os.path.join(OUT, "positions.json")))["positions"]]
boards += [f["board"] if isinstance(f, dict) else f for f in json.load(open(os.path.join(OUT, "playouts.json")))["final"]]
rng = np.random.default_rng(20264)
for i in range(600):
    p_empty = (0.08, 0.15, 0.3)[i % 3]
    cells = rng.choice(3, size=81, p=[p_empty, (1 - p_empty) / 2, (1 - p_empty) / 2])
    boards.append("".join(".XO"[c] for c in cells))
code = {None: 0, go.BLACK: 1, go.WHITE: 2}
eyes = np.array([[code[go.possible_eye(b, s)] for s in range(81)] for b in boards], np.int8)
as_u8 = np.array([[".XO".index(c) for c in b] for b in boards], np.uint8)
np.savez_compressed(os.path.join(OUT, "possible_eye.npz"), boards=as_u8, eyes=eyes)
print(f"{len(boards)} boards, {int((eyes > 0).sum())} eyes ({int((eyes == 1).sum())} black, {int((eyes == 2).sum())} white)")
