"""9x9 Go rules on the native board (libbkgo.so), with the call surface of the reference's
bokego/go.py (Game, IllegalMove, squash/unsquash, PASS, ...).

Coordinates are "squashed": sq = 9*row + col (reference go.py:4-12).  The state lives in a 192-byte
C struct (include/bokego_go.h); copying a game is a struct copy, not a deepcopy.
"""
import ctypes
import os
import re

import numpy as np

N = 9
WHITE, BLACK, EMPTY = "O", "X", "."
EMPTY_BOARD = EMPTY * (N * N)
PASS = -1
RESIGN = -2
_NO_MOVE = -3

_HERE = os.path.dirname(os.path.abspath(__file__))
GO_LIB_PATH = os.environ.get("BK_GO_LIB_PATH") or os.path.join(_HERE, "libbkgo.so")  # BK_GO_LIB_PATH: sanitizer builds


class Pos(ctypes.Structure):
    """bk_pos (include/bokego_go.h)"""
    _fields_ = [("board", ctypes.c_int8 * 81), ("libs", ctypes.c_uint8 * 81), ("libs_valid", ctypes.c_uint8),
                ("reserved", ctypes.c_uint8), ("ko", ctypes.c_int16), ("last_move", ctypes.c_int16),
                ("reserved2", ctypes.c_int16), ("turn", ctypes.c_int32), ("reserved3", ctypes.c_uint32),
                ("hash", ctypes.c_uint64)]


_PP = ctypes.POINTER(Pos)
_U8P = ctypes.POINTER(ctypes.c_uint8)
GO_SYMBOLS = {
    "bk_go_abi_version": (ctypes.c_int, []),
    "bk_pos_init": (None, [_PP]),
    "bk_pos_from_board": (ctypes.c_int, [_PP, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "bk_pos_board_string": (None, [_PP, ctypes.c_char_p]),
    "bk_pos_play": (ctypes.c_int, [_PP, ctypes.c_int]),
    "bk_pos_is_legal": (ctypes.c_int, [_PP, ctypes.c_int]),
    "bk_pos_legal_moves": (ctypes.c_int, [_PP, _U8P]),
    "bk_pos_liberties": (None, [_PP, _U8P]),
    "bk_pos_score": (ctypes.c_float, [_PP, ctypes.c_float]),
    "bk_pos_area_score": (ctypes.c_float, [_PP, ctypes.c_float]),
    "bk_pos_eye_like": (ctypes.c_int, [_PP, ctypes.c_int, ctypes.c_int]),
    "bk_pos_possible_eye": (ctypes.c_int, [_PP, ctypes.c_int]),
    "bk_pos_features_u8": (None, [_PP, ctypes.c_void_p, ctypes.c_int]),
    "bk_pos_features_f32": (None, [_PP, ctypes.c_void_p, ctypes.c_int]),
    "bk_features_batch_u8": (None, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]),
    "bk_pos_children": (ctypes.c_int, [_PP, ctypes.c_void_p, ctypes.c_void_p]),
    "bk_pos_children_slow": (ctypes.c_int, [_PP, ctypes.c_void_p, ctypes.c_void_p]),
}
_golib = None


GO_ABI_VERSION = 6     # include/bokego_go.h + bokego_tree.h (bk_go_abi_version)


def golib():
    global _golib
    if _golib is None:
        if not os.path.exists(GO_LIB_PATH):
            raise RuntimeError(f"{GO_LIB_PATH} not found: build it with `make -C bokego_amd/csrc`")
        lib = ctypes.CDLL(GO_LIB_PATH)
        for name, (res, args) in GO_SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        assert ctypes.sizeof(Pos) == 192
        if lib.bk_go_abi_version() != GO_ABI_VERSION:
            raise RuntimeError(f"{GO_LIB_PATH} is ABI {lib.bk_go_abi_version()}, this package needs {GO_ABI_VERSION}: "
                               "rebuild it with `make -C bokego_amd/csrc`")
        _golib = lib
    return _golib


_RULES = {-11: "ko", -12: "not_empty", -13: "suicide", -14: "off_board"}


class IllegalMove(Exception):
    """Raised by Game.play_move; rule_type in {"ko", "suicide", "not_empty", "off_board"} (reference go.py:279-319)."""

    def __init__(self, game, rule_type=None, sq_c=None):
        super().__init__()
        self.game, self.rule_type = game, rule_type
        self.move = unsquash(sq_c) if sq_c is not None and rule_type != "off_board" else None

    def __str__(self):
        what = {"ko": "illegally retakes ko", "suicide": "is suicide", "not_empty": "is occupied",
                "off_board": "is not on the board"}.get(self.rule_type, "is illegal")
        return f"\n{self.game}\n Move at {self.move} {what}."


class Game:
    """A game of 9x9 go; same constructor and methods as the reference's go.Game (go.py:33-277)."""

    __slots__ = ("_pos", "moves", "komi", "sgf")

    def __init__(self, board=EMPTY_BOARD, ko=None, last_move=None, turn=0, moves=None, komi=5.5, sgf=None):
        self.sgf = sgf
        self.moves = get_moves(sgf) if sgf else moves
        self.komi = komi
        self._pos = Pos()
        rc = golib().bk_pos_from_board(ctypes.byref(self._pos), board.encode("ascii"),
                                       -1 if ko is None else int(ko),
                                       _NO_MOVE if last_move is None else int(last_move), int(turn))
        if rc:
            raise ValueError("bad board string / ko / last_move")

    # ---- state -------------------------------------------------------------------------
    @property
    def board(self):
        buf = ctypes.create_string_buffer(82)
        golib().bk_pos_board_string(ctypes.byref(self._pos), buf)
        return buf.value.decode("ascii")

    @property
    def ko(self):
        return None if self._pos.ko < 0 else int(self._pos.ko)

    @property
    def last_move(self):
        lm = int(self._pos.last_move)
        return None if lm == _NO_MOVE else lm

    @property
    def turn(self):
        return int(self._pos.turn)

    def copy(self):
        g = object.__new__(type(self))
        g._pos = Pos.from_buffer_copy(self._pos)
        g.moves = list(self.moves) if self.moves is not None else None
        g.komi, g.sgf = self.komi, self.sgf
        return g

    def __copy__(self):
        return self.copy()

    def __deepcopy__(self, memo):
        return self.copy()

    def key(self):
        """(board bytes, ko, last_move, side): equality of positions as the reference's
        Go_MCTS.__eq__ (mcts.py:294-296) plus the side to move."""
        p = self._pos
        return (bytes(p.board), int(p.ko), int(p.last_move), int(p.turn) & 1)

    def __hash__(self):
        return int(self._pos.hash)

    def zobrist_hash(self):
        return int(self._pos.hash)

    def __eq__(self, other):
        return isinstance(other, Game) and self.key() == other.key()

    def __len__(self):
        return len(self.moves) if self.moves else 0

    def __repr__(self):
        return repr((self.board, self.ko, self.last_move))

    def __str__(self):
        b = self.board
        rows = ["\t   " + " ".join("ABCDEFGHJ")]
        for r in range(N):
            cells = [("+" if (9 * r + c) in (20, 24, 40, 56, 60) and b[9 * r + c] == EMPTY else b[9 * r + c])
                     for c in range(N)]
            rows.append(f"\t{r + 1}  " + " ".join(cells))
        return "\n".join(rows)

    def to_numpy(self):
        """(9,9) int8: black 1, white -1, empty 0 (reference go.py:99-107)."""
        a = np.frombuffer(bytes(self._pos.board), dtype=np.int8).reshape(N, N).copy()
        a[a == 2] = -1
        return a

    # ---- rules -------------------------------------------------------------------------
    def play_pass(self):
        golib().bk_pos_play(ctypes.byref(self._pos), PASS)
        if self.moves is not None:
            self.moves.append(PASS)

    def play_move(self, sq_c=None, testing=False):
        """Play for the side to move; sq_c=None replays the next move of an SGF (go.py:123-182)."""
        from_list = sq_c is None
        if from_list:
            if not self.moves or self.turn >= len(self.moves):
                print("No moves to play.")
                return
            sq_c = self.moves[self.turn]
        sq_c = int(sq_c)
        if testing:
            if sq_c != PASS and not golib().bk_pos_is_legal(ctypes.byref(self._pos), sq_c):
                raise IllegalMove(self, rule_type=self._why(sq_c), sq_c=sq_c)
            return
        rc = golib().bk_pos_play(ctypes.byref(self._pos), sq_c)
        if rc:
            raise IllegalMove(self, rule_type=_RULES.get(rc), sq_c=sq_c)
        if self.moves is not None and not from_list and self.sgf is None:
            self.moves.append(sq_c)

    def _why(self, sq_c):
        probe = Pos.from_buffer_copy(self._pos)
        return _RULES.get(golib().bk_pos_play(ctypes.byref(probe), sq_c))

    def is_legal(self, sq_c):
        return bool(golib().bk_pos_is_legal(ctypes.byref(self._pos), int(sq_c)))

    def get_legal_moves(self):
        """All legal moves besides PASS, ascending."""
        buf = (ctypes.c_uint8 * 81)()
        golib().bk_pos_legal_moves(ctypes.byref(self._pos), buf)
        return [i for i in range(81) if buf[i]]

    def get_liberties(self):
        """Reference-compatible (history dependent) liberty cache, go.py:220-243."""
        buf = (ctypes.c_uint8 * 81)()
        golib().bk_pos_liberties(ctypes.byref(self._pos), buf)
        return list(buf)

    def score(self):
        """Black minus white minus komi as the reference computes it (go.py:202-218): Tromp-Taylor area,
        except that stones bordering a neutral empty region are not counted (reference quirk)."""
        return float(golib().bk_pos_score(ctypes.byref(self._pos), ctypes.c_float(self.komi)))

    def area_score(self):
        """Tromp-Taylor area score proper."""
        return float(golib().bk_pos_area_score(ctypes.byref(self._pos), ctypes.c_float(self.komi)))

    def features_u8(self, fresh=False):
        out = np.empty((27, 9, 9), np.uint8)
        golib().bk_pos_features_u8(ctypes.byref(self._pos), out.ctypes.data, int(fresh))
        return out


# ---- helpers with the reference's names -------------------------------------------------
_COLS = "ABCDEFGHJ"


def squash(c):
    """(row, col) pair or alpha-numeric coordinate ("E5", "pass") -> squashed coordinate."""
    if isinstance(c, list):
        return [squash(x) for x in c]
    if isinstance(c, str):
        c = c.upper()
        if c == "PASS":
            return PASS
        m = re.fullmatch(r"([A-HJ])(\d)", c)
        if m is None:
            raise ValueError(c)
        return N * (int(m[2]) - 1) + _COLS.index(m[1])
    return N * c[0] + c[1]


def unsquash(sq_c, alph=True):
    if isinstance(sq_c, list):
        return [unsquash(x, alph) for x in sq_c]
    if sq_c == PASS:
        return "PASS"
    r, c = divmod(sq_c, N)
    return f"{_COLS[c]}{r + 1}" if alph else (r, c)


NEIGHBORS = [[9 * rr + cc for rr, cc in ((r + 1, c), (r - 1, c), (r, c + 1), (r, c - 1)) if 0 <= rr < N and 0 <= cc < N]
             for r in range(N) for c in range(N)]


def get_moves(sgf):
    """Move list of an SGF written by the reference (row-first letters, go.py:499-510)."""
    with open(sgf) as f:
        toks = re.findall(r";[BW]\[(\w*)\]", f.read())
    return [PASS if not t else 9 * (ord(t[0]) - 97) + ord(t[1]) - 97 for t in toks]


def write_sgf(moves, out_path, komi=5.5, B="", W="", result="", handicap=0):
    """SGF in the reference's dialect (row-first coordinates, go.py:528-564)."""
    s = f"(;GM[1]HA[{handicap}]RU[Chinese]"
    if B and W:
        s += f"PB[{B}]PW[{W}]"
    if result:
        s += f"RE[{result}]"
    s += f"SZ[{N}]KM[{komi}]\n"
    for i, mv in enumerate(moves):
        who = "BW"[i & 1]
        s += f";{who}[]\n" if mv == PASS else f";{who}[{chr(mv // 9 + 97)}{chr(mv % 9 + 97)}]\n"
    s += ")"
    if out_path:
        with open(out_path, "w") as f:
            f.write(s)
    return s
