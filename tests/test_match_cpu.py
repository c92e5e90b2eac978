"""The GTP match harness on CPU: in-process engines, the raw-policy opponent, and a subprocess engine."""
import os
import sys

import numpy as np
import torch

from bokego_amd import go, match
from bokego_amd.gtp import GTP
from bokego_amd.mcts import Go_MCTS

from conftest import REPO
from test_selfplay_cpu import FakeNets, _Wrap


def _gtp(f, rollouts):
    return GTP(Go_MCTS(), _Wrap(f.policy), _Wrap(f.value, True), no_sim=True, time_lim=None, n_rollouts=rollouts,
               expand_thresh=6)


def test_match_mcts_vs_policy_and_sgf(tmp_path):
    f = FakeNets()
    a = match.InProcessEngine(_gtp(f, 40), name="mcts40")
    b = match.PolicyEngine(_Wrap(f.policy), name="policy")
    res = match.play_match(a, b, n_games=2, out_sgf=str(tmp_path / "m"))
    assert res["games"] == 2 and res["mcts40_wins"] + res["policy_wins"] == 2
    assert res["ms_per_move"]["mcts40"] > 0
    for i, rec in enumerate(res["records"]):
        assert rec["a_black"] == (i % 2 == 0)
        assert go.get_moves(str(tmp_path / f"m_{i + 1}.sgf")) == rec["moves"]
        r = go.Game()
        for m in rec["moves"]:
            r.play_move(m)                      # the referee accepted only legal moves
        if rec["resigned"] is None:
            assert (r.area_score() > 0) == (rec["result"] == 1)
        else:                                   # the GTP engine resigns below 10 % win-rate after move 50
            assert rec["result"] == (-1 if rec["resigned"] == 0 else 1)


def test_subprocess_engine_speaks_gtp(tmp_path):
    # a tiny GTP engine in a subprocess (always passes): exercises the pipe protocol of SubprocessEngine
    script = tmp_path / "passer.py"
    script.write_text(
        "import sys\n"
        "for line in sys.stdin:\n"
        "    c = line.split()\n"
        "    if not c: continue\n"
        "    out = 'pass' if c[0] == 'genmove' else ''\n"
        "    sys.stdout.write('= ' + out + '\\n\\n'); sys.stdout.flush()\n"
        "    if c[0] == 'quit': break\n")
    f = FakeNets()
    a = match.InProcessEngine(_gtp(f, 20), name="mcts20")
    b = match.SubprocessEngine(f"{sys.executable} {script}", name="passer")
    g = match.play_game(a, b, max_moves=12)
    # like the reference, the engine treats a position whose last move was a pass as over and passes back
    assert len(g["moves"]) == 3 and g["moves"][1:] == [go.PASS, go.PASS] and g["result"] == 1
    b.close()


def test_random_openings_and_cpu_backend_engine_as_subprocess():
    """Seeded openings give two deterministic engines different games (a colour-swapped pair shares its opening), and
    the CPU-backend GTP engine (oracle/gtp_cpu.py: the same search on the reference's torch-CPU arithmetic, BASELINE
    configs[4]'s "CPU baseline") plays a legal game through the subprocess protocol."""
    assert match.random_opening(4, 7) == match.random_opening(4, 7) != match.random_opening(4, 8)
    f = FakeNets()
    a = match.InProcessEngine(_gtp(f, 20), name="mcts20")
    env_cmd = f"{sys.executable} -m oracle.gtp_cpu -r 30 --threads 2"
    cwd = os.getcwd()
    os.chdir(REPO)
    try:
        b = match.SubprocessEngine(env_cmd, name="boke-cpu")
        res = match.play_match(a, b, n_games=2, opening_plies=4, seed=3)
        b.close()
    finally:
        os.chdir(cwd)
    assert res["games"] == 2 and res["mcts20_wins"] + res["boke-cpu_wins"] == 2
    op = match.random_opening(4, 3)
    assert all(rec["moves"][:4] == op for rec in res["records"])
    assert res["ms_per_move"]["boke-cpu"] > 0


def test_match_cli_with_two_subprocess_engines_and_json_out(tmp_path, capsys):
    """`python -m bokego_amd.match --engine ... --opponent ... --json-out ...` (what tools/run_cfg4.sh drives for the
    CPU-backend leg): both sides behind pipes, no HIP engine is constructed, the summary is printed and written."""
    import json
    script = tmp_path / "passer.py"
    script.write_text(
        "import sys\n"
        "for line in sys.stdin:\n"
        "    c = line.split()\n"
        "    if not c: continue\n"
        "    out = 'pass' if c[0] == 'genmove' else ''\n"
        "    sys.stdout.write('= ' + out + '\\n\\n'); sys.stdout.flush()\n"
        "    if c[0] == 'quit': break\n")
    out = tmp_path / "res" / "m.json"
    match.main(["--games", "2", "--engine", f"{sys.executable} {script}", "--engine-name", "a", "--opponent",
                f"{sys.executable} {script}", "--opponent-name", "b", "--json-out", str(out)])
    printed = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    saved = json.load(open(out))
    assert printed == saved and saved["games"] == 2 and saved["a_wins"] + saved["b_wins"] == 2 and "records" not in saved
    assert saved["opponent"].endswith("passer.py") and saved["komi"] == 5.5
    # tools/run_cfg4.sh refuses to start without a gnugo binary (exit 3) instead of failing half-way
    import shutil
    import subprocess
    if shutil.which("gnugo") is None:
        r = subprocess.run([os.path.join(REPO, "tools", "run_cfg4.sh")], capture_output=True, text=True, env={**os.environ, "GNUGO": ""})
        assert r.returncode == 3 and "no gnugo" in r.stdout
