"""-m gpu: the two-engine GTP match runner on the HIP engine (BASELINE configs[4]).  The reference's own
runner (GTPprocess / GTP_match, bokego/gtp.py:450-604) is broken, boke.py:15-45 is the launcher it would start."""
import json
import os
import shutil
import subprocess
import sys

import pytest

from bokego_amd import go, match
from bokego_amd.bkw import load_bkw

from conftest import GOLDEN, REPO

pytestmark = pytest.mark.gpu


def _hip_engine(rollouts, name):
    from bokego_amd import nnet
    from bokego_amd.gtp import NativeGTP
    from bokego_amd.mcts_native import Position
    pi = nnet.HipPolicyNet(load_bkw(os.path.join(GOLDEN, "policy_19.bkw")))
    val = nnet.HipValueNet(load_bkw(os.path.join(GOLDEN, "value_synth.bkw")))
    return match.InProcessEngine(NativeGTP(Position(), pi, val, no_sim=True, time_lim=None, n_rollouts=rollouts), name=name)


def _run(tmp_path, tag, games=4, rollouts=200):
    a = _hip_engine(rollouts, "boke-hip")
    cwd = os.getcwd()
    os.chdir(REPO)                                         # `python -m oracle.gtp_cpu` resolves from the repo root
    try:
        b = match.SubprocessEngine(f"{sys.executable} -m oracle.gtp_cpu -r {rollouts} --threads 8", name="boke-cpu")
        res = match.play_match(a, b, n_games=games, opening_plies=4, seed=11, out_sgf=str(tmp_path / tag))
        b.close()
    finally:
        os.chdir(cwd)
    a.gtp.close()
    return res


def test_match_hip_engine_vs_cpu_backend_engine(tmp_path):
    """play_match with the HIP engine in-process against the CPU-backend engine (the same search on the reference's
    torch-CPU operators) in a subprocess: 4 games at 200 rollouts per move from seeded openings.  Every move is legal on
    the referee board, the SGFs re-read to the same games, ms/move is reported for both sides, and a second run of the
    match gives the same games (both engines are deterministic; HIP and torch-CPU outputs differ by < 1e-4, which the
    search does not feel: SURVEY 6.2 "move stability")."""
    first, second = _run(tmp_path, "a"), _run(tmp_path, "b")
    assert first["games"] == 4 and first["boke-hip_wins"] + first["boke-cpu_wins"] == 4
    assert 0 < first["ms_per_move"]["boke-hip"] < first["ms_per_move"]["boke-cpu"]
    op = [match.random_opening(4, 11), match.random_opening(4, 12)]
    for i, rec in enumerate(first["records"]):
        assert rec["a_black"] == (i % 2 == 0) and rec["moves"][:4] == op[i // 2]
        assert go.get_moves(str(tmp_path / f"a_{i + 1}.sgf")) == rec["moves"]
        ref = go.Game()
        for m in rec["moves"]:
            ref.play_move(m)                               # raises on an illegal move
        if rec["resigned"] is None:
            assert (ref.area_score() > 0) == (rec["result"] == 1)
        assert len(rec["moves"]) > 20
    assert [r["moves"] for r in first["records"]] == [r["moves"] for r in second["records"]]
    assert [r["result"] for r in first["records"]] == [r["result"] for r in second["records"]]
    print(f"\\nmatch: HIP {first['ms_per_move']['boke-hip']:.2f} ms/move, CPU backend {first['ms_per_move']['boke-cpu']:.1f} ms/move, "
          f"HIP wins {first['boke-hip_wins']}/4")


@pytest.mark.timeout(3300, method="thread")
@pytest.mark.skipif(shutil.which("gnugo") is None, reason="no gnugo binary on this box (BASELINE configs[4] needs one)")
def test_config4_vs_gnugo_runs_the_moment_gnugo_is_there(tmp_path):
    """BASELINE configs[4] as written.  Skips itself without a `gnugo`; with one it runs tools/run_cfg4.sh (100 games at
    -r 1600 for the HIP backend, 20 for the CPU backend) and checks what it wrote."""
    tag = "r03"
    out = subprocess.run([os.path.join(REPO, "tools", "run_cfg4.sh"), tag], capture_output=True, text=True, timeout=3000, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    summary = json.loads(out.stdout.strip().splitlines()[-1])
    h = json.load(open(os.path.join(REPO, "profiles", f"{tag}_match_vs_gnugo.json")))
    assert h["games"] == 100 and h["boke-hip-r1600_wins"] + h["gnugo_wins"] == 100
    assert summary["hip"]["ms_per_move"]["boke-hip-r1600"] < summary["cpu_backend"]["ms_per_move"]["boke-cpu-r1600"]
