"""GTP v2 front-end on the batched MCTS, with the command set and replies of the reference's
bokego/gtp.py (GTP class, gtp.py:16-399) and a launcher that replaces boke.py.

    python -m bokego_amd.gtp -r 1600 -p policy.pt -v value.pt        # fixed rollouts per move
    python -m bokego_amd.gtp -t 5                                     # 5 seconds per move

Differences from the reference, on purpose: `-r` is honoured (the reference parses it and drops
it, boke.py:17 vs 40-44); `clear_cache` answers "= " instead of failing; pondering is off unless
asked for (`--ponder` / `pondering on`; the reference's default is on: it searches in a busy loop while a thread waits
on stdin); an `analyze` reply ends with the empty line GTP requires once the next command has arrived.
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
        self._komi = None          # set by the `komi` command; re-applied to every root installed afterwards
        self.genmove_seconds = []

    # ---- main loop ---------------------------------------------------------------------------------
    def start(self, stream_in=None, stream_out=None):
        """The main loop (gtp.py:63-92).  On a live stream (stdin, a pipe: anything with a file descriptor) a thread reads
        the commands, and while none is waiting the engine does what the reference's loop does: with pondering on it keeps
        searching the current position (gtp.py:70-74), and after `analyze` it keeps printing info lines until the next
        command arrives (gtp.py:77-86), then ends that reply with the empty line GTP asks for.  On an in-memory stream
        (tests, replayed sessions) every command is already there: no thread, no pondering, one snapshot per `analyze`."""
        stream_in, stream_out = stream_in or sys.stdin, stream_out or sys.stdout
        self.running = True

        def write(text):
            stream_out.write(text)
            stream_out.flush()

        try:
            stream_in.fileno()
            live = True
        except (AttributeError, OSError, ValueError):
            live = False
        if live:
            import queue
            import threading
            q = queue.Queue()

            def reader():
                for raw in stream_in:
                    q.put(raw)
                q.put(None)

            threading.Thread(target=reader, daemon=True).start()

            def lines():
                while True:
                    ponder = self.pondering and not self.root._terminal
                    try:
                        raw = q.get(block=not ponder, timeout=None if ponder else 0.25)
                    except queue.Empty:
                        if ponder:
                            self.rollout(10)              # think in the opponent's time (gtp.py:72-73)
                        continue
                    if raw is None:
                        return
                    yield raw
            pending = lambda: not q.empty()  # noqa: E731
        else:
            lines = lambda: iter(stream_in)  # noqa: E731
            pending = lambda: True           # noqa: E731
        for line in lines():
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            out = self.send(line)
            if out is None:
                break
            if not isinstance(out, str):      # analyze: "= \n", info lines until a command is waiting, then the terminator
                write(next(out))
                write(next(out))
                while not pending():
                    write(next(out))
                out = "\n"
            write(out)
            if not self.running:
                break

    def stop(self):
        self.running = False

    # ---- protocol ------------------------------------------------------------------------------------
    # Every command is a handler `_c_<name>(args, ctx) -> (ok, text)`; send() does the framing.
    # ctx.turn is the root's turn when the command arrived.
    def send(self, cmd):
        """One GTP command -> reply string "=id text\n\n" / "?id text\n\n" (reference gtp.py:110-330)."""
        if not self.running or not cmd:
            return None
        words = cmd.lower().split()
        cmd_id = ""
        if re.match(r"\d+", words[0]):
            cmd_id, words = words[0], words[1:]
        name, args = words[0], words[1:]
        if name not in self.commands:
            ok, text = False, f"unknown command '{name}'"
        else:
            if self._komi is not None:       # roots installed by moves / search since the last command: same komi
                self._set_komi(self._komi)
            res = getattr(self, "_c_" + name)(args, self.root.turn)
            if not isinstance(res, tuple):
                return res                      # analyze: a generator of info lines
            ok, text = res
        return f"{'=' if ok else '?'}{cmd_id} {text}\n\n"

    @staticmethod
    def _side(word):
        """0 for black, 1 for white ('b', 'black', 'w', 'white')."""
        return 0 if "b" in word else 1

    def _c_protocol_version(self, args, turn):
        return True, "2"

    def _c_version(self, args, turn):
        return True, "0.3"

    def _c_name(self, args, turn):
        return True, "boke"

    def _c_known_command(self, args, turn):
        if len(args) != 1:
            return False, ""
        return True, "true" if args[0] in self.commands else "false"

    def _c_list_commands(self, args, turn):
        return True, "\n".join(self.commands)

    _c_help = _c_list_commands

    def _c_boardsize(self, args, turn):
        if args != ["9"]:
            return False, "boke only plays on 9x9 board"
        return True, ""

    def _c_clear_board(self, args, turn):
        self._install_root(self._node(), keep_komi=False)
        return True, ""

    def _c_komi(self, args, turn):
        if not args:
            return False, "usage: komi <num-komi>"
        try:
            self._set_komi(float(args[0]))
        except ValueError:
            return False, "invalid komi value"
        return True, ""

    def _set_komi(self, komi):
        """The reference writes `self.root.komi` (gtp.py:151-159) and loses it at the next move, because Go_MCTS
        nodes are rebuilt with the default (mcts.py:275-277).  Here the value stays in force for final_score /
        printsgf across moves, undo and handicap roots; `clear_board` resets it to 5.5 exactly as the reference's
        set_root(Go_MCTS()) does (gtp.py:147-149; pinned by tests/golden/gtp_transcript.json)."""
        self._komi = komi
        self.root.komi = komi

    def _install_root(self, node, keep_komi=True):
        self.set_root(node)
        if not keep_komi:
            # (and it stays in force: the Python tree hands back interned nodes -- a position reached again after the
            # clear_board -- that still carry the komi of before; found by a fuzzed session, round 4)
            self._set_komi(5.5)
        elif self._komi is not None:
            self._set_komi(self._komi)

    def _c_play(self, args, turn):
        if len(args) < 2 or args[0] not in self.colors:
            return False, "usage: play <color> <vertex>"
        if args[1] == "resign":
            self.running = False
            return True, ""
        try:
            mv = go.squash(args[1])
        except (ValueError, IndexError):
            return False, "invalid coordinate"
        try:
            if self._side(args[0]) != turn % 2:   # same colour twice in a row: a pass goes in between
                passed = self.root.make_move(go.PASS)
                if not passed.is_legal(mv):
                    return False, "illegal move"
                self._last_root = self.root
                self.set_root(passed.make_move(mv))
                self._move_history.append(mv)
                self._undid = False
            else:
                self.input_move(mv)
        except go.IllegalMove:
            return False, "illegal move"
        return True, ""

    def _c_showboard(self, args, turn):
        return True, "\n" + str(self.root)

    def _c_genmove(self, args, turn, may_resign=True):
        if len(args) != 1 or args[0] not in self.colors:
            return False, "usage: " + ("genmove" if may_resign else "reg_genmove") + " <color>"
        if self._side(args[0]) != turn % 2:
            self.input_move(go.PASS)
            self._undid = True
        mv = self.genmove(None if may_resign else False)
        if mv == go.RESIGN:
            self.running = False
            return True, "resign"
        return True, go.unsquash(mv)

    def _c_reg_genmove(self, args, turn):
        return self._c_genmove(args, turn, may_resign=False)

    def _c_undo(self, args, turn):
        if self._undid or self._last_root is None:     # one level only, like the reference
            return False, "cannot undo"
        self._install_root(self._last_root)
        self._move_history.pop()
        self._last_root, self._undid = None, True
        return True, ""

    def _c_last_move(self, args, turn):
        mv = self.root.last_move
        if mv is None:
            return False, "no previous move known"
        return True, ("black " if turn % 2 == 1 else "white ") + go.unsquash(mv)

    def _c_move_history(self, args, turn):
        return True, "\n".join(go.unsquash(self._move_history))

    def _c_quit(self, args, turn):
        self.running = False
        return True, ""

    def _c_clear_cache(self, args, turn):
        self.clear_cache()
        self._undid = True
        return True, ""

    def _c_final_score(self, args, turn):
        score = self.root.score()
        if abs(score) < 1e-4:
            return True, "0"
        return True, f"B+{score}" if score > 0 else f"W+{-score}"

    def _c_set_fixed_handicap(self, args, turn):
        if len(args) != 1 or not args[0].isnumeric():
            return False, "usage: set_fixed_handicap <num-handicaps>"
        if self.root.board != go.EMPTY_BOARD:
            return False, "board is not empty"
        n = int(args[0])
        if not 1 < n <= 5:
            return False, "invalid number of handicaps"
        stones = FLOWERS9[:n]
        self._install_root(self._node(board="".join(go.BLACK if i in stones else go.EMPTY for i in range(81)), turn=1))
        return True, " ".join(go.unsquash(list(stones)))

    def _c_printsgf(self, args, turn):
        path = args[0] if len(args) == 1 else os.path.join(os.getcwd(), "bokego.sgf")
        return True, go.write_sgf(self._move_history, path, komi=self.root.komi)

    def _c_loadsgf(self, args, turn):
        if len(args) != 2 or not args[1].isnumeric():
            return False, "usage: loadsgf <path-to-sgf> <move-number>"
        try:
            for mv in go.get_moves(args[0]):
                self.input_move(mv)
        except IOError as e:
            return False, str(e)
        except go.IllegalMove:
            return False, "illegal move in sgf"
        return True, "black" if (int(args[1]) - 1) % 2 == 0 else "white"

    def _c_analyze(self, args, turn):
        if len(args) != 2 or args[0] not in self.colors or not args[1].isnumeric():
            return False, "usage: analyze <color> <interval>"
        if self._side(args[0]) != turn % 2:
            return False, f"it is not {args[0]}'s turn"
        return self.analyze(int(args[1]))

    def _c_pondering(self, args, turn):
        if len(args) != 1 or args[0] not in ("on", "off"):
            return False, "usage: pondering <on/off>"
        self.pondering = args[0] == "on"
        return True, ""

    # ---- engine side ---------------------------------------------------------------------------------
    def input_move(self, sq_c):
        node = self.root.make_move(sq_c)
        self._last_root = self.root
        self._install_root(node)
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
            yield self.analyze_line(variations, k)

    def analyze_line(self, variations, k=3):
        """One info line from the variations a search has collected: the k most visited root children, best last, each
        with visits, winrate (from the mover's side, x 10,000), prior and the last line searched through it.  Ties in the
        visit count are ordered by the move, so that the Python and the native tree print the same line."""
        best = sorted(variations, key=lambda n: (self.N[n], -n.last_move))
        probs = self.root.dist.probs
        out = ""
        for n in best[-k:]:
            pv = go.unsquash([m.last_move for m in variations[n]])
            out += (f"info move {go.unsquash(n.last_move)} visits {self.N[n]} winrate {10000 * (1 - n.winrate):.0f} "
                    f"prior {10000 * probs[n.last_move]:.0f} pv " + " ".join(pv) + " ")
        return out + "\n"


class GTP(_GTPProtocol, MCTS):
    """The reference's GTP(MCTS) (gtp.py:16) on the batched Python tree."""


from .mcts_native import NativeMCTS, Position  # noqa: E402


class NativeGTP(_GTPProtocol, NativeMCTS):
    """Same protocol on the native tree core: ~10x less host time per genmove."""
    _node = Position

    def _set_komi(self, komi):
        # NativeMCTS.root is a fresh snapshot on every access: the komi lives on the tree
        self._komi = self.komi = komi


def load_state_dict(path):
    """A reference checkpoint ({"model_state_dict": ...}, boke.py:31-37), a bare state_dict, or a BKW1 file."""
    if path.endswith(".bkw"):
        from .bkw import load_bkw
        return load_bkw(path)
    import torch
    ck = torch.load(path, map_location="cpu")
    return ck["model_state_dict"] if "model_state_dict" in ck else ck


def build_parser():
    """The reference launcher's flags (boke.py:15-26: -t -r -p -v -g/--gpu --simulate) with the same meaning, plus this
    engine's own.  `-g` may be given bare, as the reference's store_true flag is, or with a GPU index."""
    golden = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    ap = argparse.ArgumentParser(description="BokeGo GTP engine on the MI355X leaf-evaluation engine")
    ap.add_argument("-t", metavar="SEC", type=float, default=10.0, help="time limit in seconds for each move")
    ap.add_argument("-r", type=int, default=None, help="number of rollouts per move (overrides -t)")
    ap.add_argument("-p", metavar="PATH", default=os.path.join(golden, "policy_19.bkw"), help="policy weights (.pt/.bkw)")
    ap.add_argument("-v", metavar="PATH", default=os.path.join(golden, "value_synth.bkw"), help="value weights (.pt/.bkw)")
    ap.add_argument("-g", "--gpu", type=int, nargs="?", const=0, default=0, metavar="INDEX",
                    help="GPU index (default 0; the networks always run on the GPU -- a bare -g is accepted as in the reference)")
    ap.add_argument("--simulate", action="store_true",
                    help="enable simulations to end of game (the reference's flag, boke.py:24: no_sim=False -- playouts sampled from the policy)")
    ap.add_argument("--precision", choices=["f32", "f16x2"], default=None,
                    help="conv arithmetic: f32 (default, the reference's width) or the opt-in split-fp16 fast path")
    ap.add_argument("--ponder", action="store_true")
    ap.add_argument("--python-tree", action="store_true", help="search with the Python tree instead of the native one")
    return ap


def main(argv=None):
    args = build_parser().parse_args(argv)

    from . import nnet
    pi = nnet.HipPolicyNet(load_state_dict(args.p), device_id=args.gpu, precision=args.precision)
    val = nnet.HipValueNet(load_state_dict(args.v), device_id=args.gpu, precision=args.precision)
    cls, root = (GTP, Go_MCTS()) if args.python_tree else (NativeGTP, Position())
    gtp = cls(root, pi, val, no_sim=not args.simulate, time_lim=None if args.r else args.t, n_rollouts=args.r, pondering=args.ponder)
    gtp.start()


if __name__ == "__main__":
    main()
