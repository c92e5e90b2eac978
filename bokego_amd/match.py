"""Two-engine GTP match runner (BASELINE config 5; replaces the reference's broken
GTPprocess / GTP_match, bokego/gtp.py:450-604).

Engines speak GTP either in-process (any object with `send(cmd) -> reply`, e.g. bokego_amd.gtp.GTP)
or as a subprocess (`gnugo --mode gtp --level 10`, another `python -m bokego_amd.gtp ...`).
A referee board (bokego_amd.go) validates every move and scores the final position by area.

    python -m bokego_amd.match --games 20 -r 400 --opponent policy
    python -m bokego_amd.match --games 100 -r 1600 --opponent "gnugo --mode gtp --chinese-rules"
    python -m bokego_amd.match --games 100 -r 1600 --opening-plies 4 --opponent "python -m oracle.gtp_cpu -r 1600"   # CPU-backend baseline
    tools/run_cfg4.sh            # BASELINE configs[4] as written (needs a `gnugo` binary): both backends against GNU Go
"""
import argparse
import json
import os
import shlex
import subprocess
import sys
import time

import numpy as np

from . import go


class InProcessEngine:
    def __init__(self, gtp, name="engine"):
        self.gtp, self.name = gtp, name
        gtp.running = True

    def send(self, cmd):
        out = self.gtp.send(cmd)
        self.gtp.running = True  # a finished game must not stop the match
        if out is None or out[0] != "=":
            raise RuntimeError(f"{self.name}: '{cmd}' -> {out!r}")
        return out[1:].strip()

    def close(self):
        pass


class SubprocessEngine:
    """A GTP engine behind a pipe.  An engine that leaves its loop when a game ends -- the reference's does (a resignation
    or `quit` clears `running`, gtp.py:110-118), and so does `python -m bokego_amd.gtp` -- is started again for the
    next game: a command that finds the pipe closed is re-sent once to a fresh process."""

    def __init__(self, command, name=None):
        self.command = command
        self.name = name or command.split()[0]
        self.restarts = 0
        self._spawn()

    def _spawn(self):
        self.p = subprocess.Popen(shlex.split(self.command), stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, bufsize=1)

    def _exchange(self, cmd):
        self.p.stdin.write(cmd + "\n")
        self.p.stdin.flush()
        lines = []
        while True:
            line = self.p.stdout.readline()
            if line == "":
                return None                       # the engine closed the pipe
            if line.strip() == "" and lines:
                break
            if line.strip():
                lines.append(line.rstrip("\n"))
        return lines

    def send(self, cmd):
        try:
            lines = self._exchange(cmd)
        except BrokenPipeError:
            lines = None
        if lines is None:
            try:
                self.p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                self.p.kill()
            if cmd.split()[0] == "quit":
                return ""
            # a fresh process has an empty board: only the commands that open a game may be replayed to it
            if cmd.split()[0] not in ("boardsize", "clear_board", "name", "version", "protocol_version"):
                raise RuntimeError(f"{self.name}: engine died during '{cmd}' (exit code {self.p.poll()})")
            self.restarts += 1
            self._spawn()
            lines = self._exchange(cmd)
            if lines is None:
                raise RuntimeError(f"{self.name}: engine closed the pipe during '{cmd}'")
        if not lines[0].startswith("="):
            raise RuntimeError(f"{self.name}: '{cmd}' -> {lines}")
        return "\n".join(lines)[1:].strip()

    def close(self):
        try:
            self.send("quit")
        except Exception:
            pass
        try:
            self.p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            self.p.kill()


class PolicyEngine:
    """No-search opponent: plays the raw policy's best legal, non-eye-filling move (passes when none)."""

    def __init__(self, policy_net, name="policy", sample_seed=None):
        self.policy, self.name = policy_net, name
        self.rng = np.random.default_rng(sample_seed) if sample_seed is not None else None
        self.g = go.Game()

    def send(self, cmd):
        from . import nnet
        import ctypes
        c = cmd.lower().split()
        if c[0] == "clear_board":
            self.g = go.Game(komi=self.g.komi)
        elif c[0] == "komi":
            self.g.komi = float(c[1])
        elif c[0] == "play":
            mv = go.squash(c[2])
            if (0 if "b" in c[1] else 1) != self.g.turn % 2:
                self.g.play_move(go.PASS)
            self.g.play_move(mv)
        elif c[0] == "genmove":
            if (0 if "b" in c[1] else 1) != self.g.turn % 2:
                self.g.play_move(go.PASS)
            probs = nnet.policy_dist(self.policy, self.g).probs.numpy().copy()
            color = 1 if self.g.turn % 2 == 0 else 2
            ok = [m for m in self.g.get_legal_moves() if not go.golib().bk_pos_eye_like(ctypes.byref(self.g._pos), m, color)]
            if not ok:
                mv = go.PASS
            elif self.rng is not None:
                p = probs[ok] + 1e-9
                mv = int(self.rng.choice(ok, p=p / p.sum()))
            else:
                mv = max(ok, key=lambda m: probs[m])
            self.g.play_move(mv)
            return go.unsquash(mv)
        return ""

    def close(self):
        pass


def random_opening(plies, seed):
    """`plies` uniformly random legal, non-eye-filling moves from the empty board (seeded): two deterministic engines
    would otherwise repeat the same game; a colour-swapped pair of games shares its opening."""
    import ctypes
    rng = np.random.default_rng(seed)
    g = go.Game(moves=[])
    for _ in range(plies):
        color = 1 if g.turn % 2 == 0 else 2
        ok = [m for m in g.get_legal_moves() if not go.golib().bk_pos_eye_like(ctypes.byref(g._pos), m, color)]
        if not ok:
            break
        g.play_move(int(ok[int(rng.integers(0, len(ok)))]))
    return list(g.moves)


def play_game(black, white, komi=5.5, max_moves=162, opening=()):
    """One game; returns dict(result=+1 black / -1 white, score, moves, seconds per colour)."""
    ref = go.Game(moves=[], komi=komi)
    for e in (black, white):
        e.send("boardsize 9")
        e.send("clear_board")
        e.send(f"komi {komi}")
    for mv in opening:
        col = "bw"[ref.turn % 2]
        ref.play_move(mv)
        for e in (black, white):
            e.send(f"play {col} {go.unsquash(mv)}")
    secs, n = [0.0, 0.0], [0, 0]
    passes, resigned = 0, None
    while passes < 2 and ref.turn < max_moves:
        side = ref.turn % 2
        mover, other = (black, white)[side], (white, black)[side]
        col = "bw"[side]
        t0 = time.perf_counter()
        ans = mover.send(f"genmove {col}").strip().upper()
        secs[side] += time.perf_counter() - t0
        n[side] += 1
        if ans == "RESIGN":
            resigned = side
            break
        mv = go.squash(ans)
        ref.play_move(mv)  # raises go.IllegalMove on an illegal engine move
        passes = passes + 1 if mv == go.PASS else 0
        other.send(f"play {col} {ans}")
    score = ref.area_score()
    result = (-1 if resigned == 0 else 1) if resigned is not None else (1 if score > 0 else -1)
    return {"result": result, "score": score, "moves": list(ref.moves), "resigned": resigned,
            "ms_per_move": [1e3 * secs[i] / max(1, n[i]) for i in (0, 1)]}


def play_match(a, b, n_games=10, komi=5.5, out_sgf=None, opening_plies=0, seed=0, progress=None):
    """a and b alternate colours; returns win counts and mean ms/move of each.  progress: a file that gets one JSON
    line of running totals per finished game (a match cut short still leaves its figures)."""
    wins, ms, games = [0, 0], [[], []], []
    t_start = time.perf_counter()
    for gidx in range(n_games):
        a_black = gidx % 2 == 0
        op = random_opening(opening_plies, seed + gidx // 2) if opening_plies else ()
        g = play_game(a, b, komi, opening=op) if a_black else play_game(b, a, komi, opening=op)
        a_won = (g["result"] == 1) == a_black
        wins[0 if a_won else 1] += 1
        ms[0].append(g["ms_per_move"][0 if a_black else 1])
        ms[1].append(g["ms_per_move"][1 if a_black else 0])
        games.append({"a_black": a_black, **g})
        if progress is not None:
            print(json.dumps({"games": gidx + 1, a.name + "_wins": wins[0], b.name + "_wins": wins[1],
                              "ms_per_move": {a.name: float(np.mean(ms[0])), b.name: float(np.mean(ms[1]))},
                              "elapsed_s": time.perf_counter() - t_start}), file=progress, flush=True)
        if out_sgf:
            os.makedirs(os.path.dirname(os.path.abspath(out_sgf)), exist_ok=True)
            go.write_sgf(g["moves"], f"{out_sgf}_{gidx + 1}.sgf", komi=komi, B=a.name if a_black else b.name,
                         W=b.name if a_black else a.name, result=("B+" if g["score"] > 0 else "W+") + f"{abs(g['score'])}")
    return {"games": n_games, a.name + "_wins": wins[0], b.name + "_wins": wins[1], "win_rate": wins[0] / n_games,
            "ms_per_move": {a.name: float(np.mean(ms[0])), b.name: float(np.mean(ms[1]))}, "records": games}


def main(argv=None):
    import os
    golden = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    ap = argparse.ArgumentParser(description="GTP match: the HIP MCTS engine vs an opponent")
    ap.add_argument("--games", type=int, default=10)
    ap.add_argument("-r", type=int, default=400, help="rollouts per move of the HIP engine")
    ap.add_argument("-p", default=os.path.join(golden, "policy_19.bkw"))
    ap.add_argument("-v", default=os.path.join(golden, "value_synth.bkw"))
    ap.add_argument("--opponent", default="policy", help='"policy" (raw policy, no search) or a GTP command line')
    ap.add_argument("--komi", type=float, default=5.5)
    ap.add_argument("--sgf", default=None, help="prefix for SGF records")
    ap.add_argument("--opening-plies", type=int, default=0, help="seeded random opening moves per game pair (variety between deterministic engines)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--precision", choices=["f32", "f16x2"], default=None)
    ap.add_argument("--engine", default=None, help="GTP command line of side A instead of the in-process HIP engine "
                                                   "(e.g. the CPU-backend engine: \"python -m oracle.gtp_cpu -r 1600\")")
    ap.add_argument("--engine-name", default=None)
    ap.add_argument("--opponent-name", default=None)
    ap.add_argument("--json-out", default=None, help="also write the result (without the move records) to this file")
    args = ap.parse_args(argv)
    pi = None
    if args.engine is None or args.opponent == "policy":
        from . import nnet
        from .gtp import NativeGTP, load_state_dict
        from .mcts_native import Position
        pi = nnet.HipPolicyNet(load_state_dict(args.p), precision=args.precision)
    if args.engine is None:
        val = nnet.HipValueNet(load_state_dict(args.v), precision=args.precision)
        a = InProcessEngine(NativeGTP(Position(), pi, val, no_sim=True, time_lim=None, n_rollouts=args.r),
                            name=args.engine_name or f"boke-hip-r{args.r}")
    else:
        a = SubprocessEngine(args.engine, name=args.engine_name)
    if args.opponent == "policy":
        b = PolicyEngine(pi)
    else:
        b = SubprocessEngine(args.opponent, name=args.opponent_name)
    res = play_match(a, b, args.games, args.komi, args.sgf, args.opening_plies, args.seed, progress=sys.stderr)
    res.pop("records")
    res.update(engine=args.engine or f"in-process HIP engine, -r {args.r}, precision {args.precision or 'f32'}",
               opponent=args.opponent, komi=args.komi, opening_plies=args.opening_plies, seed=args.seed)
    print(json.dumps(res))
    if args.json_out:
        os.makedirs(os.path.dirname(os.path.abspath(args.json_out)), exist_ok=True)
        with open(args.json_out, "w") as f:
            json.dump(res, f, indent=1)
    a.close()
    b.close()


if __name__ == "__main__":
    main()
