"""GTP v2 front-end on the batched MCTS, with the command set and replies of the reference's
bokego/gtp.py (GTP class, gtp.py:16-399) and a launcher that replaces boke.py.

    python -m bokego_amd.gtp -r 1600 -p policy.pt -v value.pt        # fixed rollouts per move
    python -m bokego_amd.gtp -t 5                                     # 5 seconds per move

Differences from the reference, on purpose: `-r` is honoured (the reference parses it and drops
it, boke.py:17 vs 40-44); `clear_cache` answers "= " instead of failing; pondering is off unless
asked for (the reference ponders in a busy loop while a thread waits on stdin).
Replies are pinned against a transcript recorded from the reference (tests/golden/gtp_transcript.json).
"""
import argparse
import os
import re
import sys
from timeit import default_timer

from . import go
from .mcts import MCTS, Go_MCTS

FLOWERS9 = (20, 60, 24, 56, 40)


class _GTPProtocol:
    """Go Text Protocol on a tree searcher (mixed in before MCTS or NativeMCTS).
    kwargs as the reference: pondering, time_lim (20.0), n_rollouts."""

    _node = Go_MCTS   # node type of fresh roots

    colors = ("black", "b", "w", "white")
    commands = ("name", "boardsize", "clear_board", "komi", "play", "genmove", "reg_genmove", "final_score",
                "quit", "version", "showboard", "clear_cache", "last_move", "move_history", "undo", "help",
                "known_command", "protocol_version", "list_commands", "set_fixed_handicap", "printsgf", "loadsgf",
                "analyze", "pondering")

    def __init__(self, root, policy_net, value_net=None, **kwargs):
        self.time_lim = kwargs.pop("time_lim", 20.0)
        self.n_rollouts = kwargs.pop("n_rollouts", None)
        self.pondering = kwargs.pop("pondering", False)
        kwargs.pop("connection", None)
        super().__init__(root, policy_net, value_net, **kwargs)
        self.running = False
        self._move_history = []
        self._last_root = None
        self._undid = False
        self.genmove_seconds = []

    # ---- main loop ---------------------------------------------------------------------------------
    def start(self, stream_in=None, stream_out=None):
        stream_in, stream_out = stream_in or sys.stdin, stream_out or sys.stdout
        self.running = True
        for line in stream_in:
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            out = self.send(line)
            if out is None:
                break
            if not isinstance(out, str):      # analyze generator: one snapshot, then the terminator
                out = "".join([next(out), next(out), "\n"])
            stream_out.write(out)
            stream_out.flush()
            if not self.running:
                break

    def stop(self):
        self.running = False

    # ---- protocol ------------------------------------------------------------------------------------
    def send(self, cmd):
        """One GTP command -> reply string "=id text\\n\\n" / "?id text\\n\\n" (reference gtp.py:110-330)."""
        if not self.running or not cmd:
            return None
        valid, out, cmd_id = False, "", ""
        cmd = cmd.lower().split()
        if re.match(r"\d+", cmd[0]):
            cmd_id, cmd = cmd[0], cmd[1:]
        this_turn = self.root.turn
        c = cmd[0]

        if c not in _GTPProtocol.commands:
            out = f"unknown command '{c}'"
        elif c == "protocol_version":
            out, valid = "2", True
        elif c == "version":
            out, valid = "0.3", True
        elif c == "name":
            out, valid = "boke", True
        elif c == "known_command":
            if len(cmd) == 2:
                out, valid = ("true" if cmd[1] in _GTPProtocol.commands else "false"), True
        elif c == "boardsize":
            if len(cmd) != 2 or cmd[1] != "9":
                out = "boke only plays on 9x9 board"
            else:
                valid = True
        elif c == "clear_board":
            self.set_root(self._node())
            valid = True
        elif c == "komi":
            if len(cmd) < 2:
                out = "usage: komi <num-komi>"
            else:
                try:
                    self.root.komi = float(cmd[1])
                    valid = True
                except ValueError:
                    out = "invalid komi value"
        elif c == "play":
            if len(cmd) < 3 or cmd[1] not in _GTPProtocol.colors:
                out = "usage: play <color> <vertex>"
            elif cmd[2] == "resign":
                valid, self.running = True, False
            else:
                try:
                    mv = go.squash(cmd[2])
                except (ValueError, IndexError):
                    mv, out = None, "invalid coordinate"
                if mv is not None:
                    turn = 0 if "b" in cmd[1] else 1
                    if turn != this_turn % 2:      # same colour twice in a row: a pass is inserted
                        new = self.root.make_move(go.PASS)
                        if new.is_legal(mv):
                            self._last_root = self.root
                            self.set_root(new.make_move(mv))
                            self._move_history.append(mv)
                            self._undid = False
                            valid = True
                        else:
                            out = "illegal move"
                    else:
                        try:
                            self.input_move(mv)
                            valid = True
                        except go.IllegalMove:
                            out = "illegal move"
        elif c == "showboard":
            out, valid = "\n" + str(self.root), True
        elif c in ("genmove", "reg_genmove"):
            if len(cmd) != 2 or cmd[1] not in _GTPProtocol.colors:
                out = f"usage: {c} <color>"
            else:
                turn = 0 if "b" in cmd[1] else 1
                if turn != this_turn % 2:
                    self.input_move(go.PASS)
                    self._undid = True
                mv = self.genmove(False if c == "reg_genmove" else None)
                if mv == go.RESIGN:
                    out, self.running = "resign", False
                else:
                    out = go.unsquash(mv)
                valid = True
        elif c == "undo":
            if self._undid or self._last_root is None:
                out = "cannot undo"
            else:
                self.set_root(self._last_root)
                self._move_history.pop()
                self._last_root, self._undid, valid = None, True, True
        elif c == "last_move":
            mv = self.root.last_move
            if mv is None:
                out = "no previous move known"
            else:
                out, valid = ("black " if this_turn % 2 == 1 else "white ") + go.unsquash(mv), True
        elif c == "quit":
            self.running, valid = False, True
        elif c in ("help", "list_commands"):
            out, valid = "\n".join(_GTPProtocol.commands), True
        elif c == "clear_cache":
            self.clear_cache()
            self._undid, valid = True, True
        elif c == "final_score":
            score = self.root.score()
            out = "0" if abs(score) < 1e-4 else (f"B+{score}" if score > 0 else f"W+{-score}")
            valid = True
        elif c == "move_history":
            out, valid = "\n".join(go.unsquash(self._move_history)), True
        elif c == "set_fixed_handicap":
            if len(cmd) != 2 or not cmd[1].isnumeric():
                out = "usage: set_fixed_handicap <num-handicaps>"
            elif self.root.board != go.EMPTY_BOARD:
                out = "board is not empty"
            elif not 1 < int(cmd[1]) <= 5:
                out = "invalid number of handicaps"
            else:
                stones = FLOWERS9[:int(cmd[1])]
                board = "".join(go.BLACK if i in stones else go.EMPTY for i in range(81))
                self.set_root(self._node(board=board, turn=1))
                out, valid = " ".join(go.unsquash(list(stones))), True
        elif c == "printsgf":
            path = cmd[1] if len(cmd) == 2 else os.path.join(os.getcwd(), "bokego.sgf")
            out, valid = go.write_sgf(self._move_history, path, komi=self.root.komi), True
        elif c == "loadsgf":
            if len(cmd) != 3 or not cmd[2].isnumeric():
                out = "usage: loadsgf <path-to-sgf> <move-number>"
            else:
                try:
                    for mv in go.get_moves(cmd[1]):
                        self.input_move(mv)
                    out, valid = ("black" if (int(cmd[2]) - 1) % 2 == 0 else "white"), True
                except IOError as e:
                    out = str(e)
                except go.IllegalMove:
                    out = "illegal move in sgf"
        elif c == "analyze":
            if len(cmd) != 3 or cmd[1] not in _GTPProtocol.colors or not cmd[2].isnumeric():
                out = "usage: analyze <color> <interval>"
            elif (0 if "b" in cmd[1] else 1) != this_turn % 2:
                out = f"it is not {cmd[1]}'s turn"
            elif not hasattr(self, "N"):
                out = "analyze needs the Python tree (start without --native)"
            else:
                return self.analyze(int(cmd[2]))
        elif c == "pondering":
            if len(cmd) != 2 or cmd[1] not in ("on", "off"):
                out = "usage: pondering <on/off>"
            else:
                self.pondering, valid = cmd[1] == "on", True
        return f"{'=' if valid else '?'}{cmd_id} {out}\n\n"

    # ---- engine side ---------------------------------------------------------------------------------
    def input_move(self, sq_c):
        node = self.root.make_move(sq_c)
        self._last_root = self.root
        self.set_root(node)
        self._move_history.append(sq_c)
        self._undid = False

    @property
    def surrender(self):
        return self.winrate() is not None and self.winrate() < 0.1 and self.root.turn > 50

    def genmove(self, resign=None):
        """Search, choose, re-root; returns the squashed move (gtp.py:344-366)."""
        if (resign if resign is not None else self.surrender):
            self.running = False
            return go.RESIGN
        t0 = default_timer()
        if self.time_lim:
            self.timed_rollout(self.time_lim)
        elif self.n_rollouts:
            self.rollout(self.n_rollouts)
        self._last_root = self.root
        # a terminal root (last move was a pass, or turn > MAX_TURNS) has nothing to choose from: the
        # reference returns the root's own last move there, which is PASS only in the first case
        mv = go.PASS if self.root._terminal else self.choose().last_move
        self.genmove_seconds.append(default_timer() - t0)
        self._move_history.append(mv)
        self._undid = False
        return mv

    def timed_rollout(self, time, analyze_dict=None):
        t0 = default_timer()
        while default_timer() < t0 + time:
            self.rollout(16, analyze_dict=analyze_dict)

    def analyze(self, interval, k=3):
        """Sabaki-style analysis lines (gtp.py:374-399): yields "= \\n", then one info line per interval."""
        variations = {}
        yield "= \n"
        while True:
            self.timed_rollout(interval / 200.0, analyze_dict=variations)
            best = sorted(variations, key=lambda n: self.N[n])
            out = ""
            for n in best[-k:]:
                pv = go.unsquash([m.last_move for m in variations[n]])
                prior = self.root.dist.probs[n.last_move]
                out += (f"info move {go.unsquash(n.last_move)} visits {self.N[n]} winrate {10000 * (1 - n.winrate):.0f} "
                        f"prior {10000 * prior:.0f} pv " + " ".join(pv) + " ")
            yield out + "\n"


class GTP(_GTPProtocol, MCTS):
    """The reference's GTP(MCTS) (gtp.py:16) on the batched Python tree."""


from .mcts_native import NativeMCTS, Position  # noqa: E402


class NativeGTP(_GTPProtocol, NativeMCTS):
    """Same protocol on the native tree core: ~10x less host time per genmove."""
    _node = Position


def load_state_dict(path):
    """A reference checkpoint ({"model_state_dict": ...}, boke.py:31-37), a bare state_dict, or a BKW1 file."""
    if path.endswith(".bkw"):
        from .bkw import load_bkw
        return load_bkw(path)
    import torch
    ck = torch.load(path, map_location="cpu")
    return ck["model_state_dict"] if "model_state_dict" in ck else ck


def main(argv=None):
    golden = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    ap = argparse.ArgumentParser(description="BokeGo GTP engine on the MI355X leaf-evaluation engine")
    ap.add_argument("-t", metavar="SEC", type=float, default=10.0, help="time limit in seconds for each move")
    ap.add_argument("-r", type=int, default=None, help="number of rollouts per move (overrides -t)")
    ap.add_argument("-p", metavar="PATH", default=os.path.join(golden, "policy_19.bkw"), help="policy weights (.pt/.bkw)")
    ap.add_argument("-v", metavar="PATH", default=os.path.join(golden, "value_synth.bkw"), help="value weights (.pt/.bkw)")
    ap.add_argument("-g", "--gpu", type=int, default=0, help="GPU index")
    ap.add_argument("--precision", choices=["f16x2", "f32"], default=None)
    ap.add_argument("--ponder", action="store_true")
    ap.add_argument("--python-tree", action="store_true", help="search with the Python tree (needed for `analyze`)")
    args = ap.parse_args(argv)

    from . import nnet
    pi, val = nnet.HipPolicyNet(load_state_dict(args.p), device_id=args.gpu), nnet.HipValueNet(load_state_dict(args.v), device_id=args.gpu)
    cls, root = (GTP, Go_MCTS()) if args.python_tree else (NativeGTP, Position())
    gtp = cls(root, pi, val, no_sim=True, time_lim=None if args.r else args.t, n_rollouts=args.r, pondering=args.ponder)
    if args.precision:
        gtp.evaluator.engine.set_precision(args.precision)
    gtp.start()


if __name__ == "__main__":
    main()
