/*
 * bokego_go.h -- C ABI of the native 9x9 board + feature encoder (libbkgo.so, host only).
 *
 * Produces the input of the leaf-evaluation path: the 27 feature planes of the reference's
 * nnet.features() (bokego/nnet.py:182-262) from a position maintained with the observable
 * behaviour of the reference's rules engine go.Game (bokego/go.py:33-277), including the two
 * quirks that the golden vectors pin:
 *   - Game._libs is a lazily refreshed liberty cache (go.py:220-243): it is refreshed only
 *     around the last move, only if the cache entry of the last move is 0, and captured points
 *     keep their old counts; planes 6-12 therefore depend on the move history;
 *   - get_caps (go.py:404-418) counts a captured chain once per point at which it touches the
 *     capturing move; planes 20-26 use that count.
 * `fresh != 0` in the feature calls gives the history-free variant (liberties recomputed from
 * the board), the reference's behaviour for a Game constructed from a board string.
 *
 * A bk_pos is a plain 192-byte value: copy it with memcpy/struct assignment (this replaces the
 * reference's deepcopy in Go_MCTS.make_move, mcts.py:340-346).
 */
#ifndef BOKEGO_GO_H
#define BOKEGO_GO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BK_EMPTY 0
#define BK_BLACK 1 /* 'X', moves first (go.py:23,142) */
#define BK_WHITE 2 /* 'O' */

#define BK_PASS (-1)    /* go.PASS */
#define BK_NO_MOVE (-3) /* last_move is None */
#define BK_NO_KO (-1)   /* ko is None */

#define BK_ILLEGAL_KO (-11)        /* IllegalMove rule_type "ko"        go.py:137-138 */
#define BK_ILLEGAL_NOT_EMPTY (-12) /* IllegalMove rule_type "not_empty" go.py:139-140 */
#define BK_ILLEGAL_SUICIDE (-13)   /* IllegalMove rule_type "suicide"   go.py:155-157 */
#define BK_ILLEGAL_OFF_BOARD (-14)

typedef struct bk_pos {
    int8_t board[81];   /* BK_EMPTY / BK_BLACK / BK_WHITE, index = 9*row + col (go.py:8) */
    uint8_t libs[81];   /* Game._libs                                                   */
    uint8_t libs_valid; /* 0: _libs is None                                             */
    uint8_t reserved;
    int16_t ko;         /* BK_NO_KO or 0..80                                            */
    int16_t last_move;  /* BK_NO_MOVE, BK_PASS or 0..80                                 */
    int16_t reserved2;
    int32_t turn;       /* starts at 0; even = black to move                            */
    uint32_t reserved3;
    uint64_t hash;      /* Zobrist over (stones, ko, side to move)                      */
} bk_pos;

int bk_go_abi_version(void);

/* Game() / Game(board=..., ko=..., last_move=..., turn=...)  go.py:51-66 */
void bk_pos_init(bk_pos *p);
int bk_pos_from_board(bk_pos *p, const char *board81 /* 'X','O','.' */, int ko, int last_move, int turn);
void bk_pos_board_string(const bk_pos *p, char out82[82]);

/* Game.play_move / play_pass  go.py:109-182.  Returns 0 or a BK_ILLEGAL_* code (state untouched). */
int bk_pos_play(bk_pos *p, int move);
/* Game.is_legal  go.py:184-200 */
int bk_pos_is_legal(const bk_pos *p, int move);
/* Game.get_legal_moves go.py:245-260: legal[i] = 1/0; returns the count */
int bk_pos_legal_moves(const bk_pos *p, uint8_t legal[81]);
/* Game.get_liberties go.py:220-243 (refreshes the cache exactly as the reference does) */
void bk_pos_liberties(bk_pos *p, uint8_t out[81]);
/* Game.score go.py:202-218, bug-compatible: the reference also repaints the stones bordering each
 * empty region, so stones next to neutral points are not counted (pinned by the GTP transcript) */
float bk_pos_score(const bk_pos *p, float komi);
/* Tromp-Taylor area score proper: black stones+territory minus white's minus komi (self-play results) */
float bk_pos_area_score(const bk_pos *p, float komi);
/* single-point eye test used by the build's playout generator: all on-board neighbours are `color` */
int bk_pos_eye_like(const bk_pos *p, int sq, int color);
/* go.possible_eye go.py:470-485 with the reference's DIAGONALS table (go.py:372-373: (x-1,y-1) twice, (x-1,y+1) never): the
 * colour (BK_BLACK / BK_WHITE) of the one-point eye at sq, or 0.  Go_MCTS.get_move (mcts.py:354) rejects a sampled move when
 * this is the mover's own colour. */
int bk_pos_possible_eye(const bk_pos *p, int sq);

/* nnet.features() nnet.py:182-262: 27 planes x 81, values 0..7 */
void bk_pos_features_u8(bk_pos *p, uint8_t out[2187], int fresh);
void bk_pos_features_f32(bk_pos *p, float out[2187], int fresh);
/* n positions -> contiguous [n][27][9][9]; positions are `stride` bytes apart (>= sizeof(bk_pos)) */
void bk_features_batch_u8(void *pos, int n, int stride, uint8_t *out, int fresh);

/* children of a node: for every legal move (ascending index) a copy of *p with the move played
 * (Go_MCTS.find_children, mcts.py:309-317).  Returns the number written to out/moves (<= 81). */
int bk_pos_children(const bk_pos *p, bk_pos *out, int16_t *moves);
/* the same by definition -- one bk_pos_play per empty point; bk_pos_children finds the position's chains once instead
 * (tests compare the two record for record) */
int bk_pos_children_slow(const bk_pos *p, bk_pos *out, int16_t *moves);

#ifdef __cplusplus
}
#endif
#endif
