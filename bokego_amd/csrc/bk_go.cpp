// bk_go.cpp -- native 9x9 board + nnet.features() encoder (host only, no HIP).
// Observable behaviour follows the reference's bokego/go.py and bokego/nnet.py:182-262 (see
// include/bokego_go.h for the per-function citations); the implementation is array/flood-fill
// based on a fixed 192-byte value type instead of immutable strings + deepcopy.
#include <cstring>

#include "../../include/bokego_go.h"

namespace {

constexpr int N = 9, NN = 81;

struct Tables {
    int8_t nbr[NN][4];
    int8_t nn[NN];
    uint64_t z[3][NN];
    uint64_t flip;
    Tables() {
        // neighbour order of the reference: (x+1,y) (x-1,y) (x,y+1) (x,y-1), x = row   go.py:370
        for (int r = 0; r < N; ++r)
            for (int c = 0; c < N; ++c) {
                int k = 0, s = r * N + c;
                if (r + 1 < N) nbr[s][k++] = (int8_t)(s + N);
                if (r - 1 >= 0) nbr[s][k++] = (int8_t)(s - N);
                if (c + 1 < N) nbr[s][k++] = (int8_t)(s + 1);
                if (c - 1 >= 0) nbr[s][k++] = (int8_t)(s - 1);
                nn[s] = (int8_t)k;
            }
        uint64_t x = 0x9E3779B97F4A7C15ull;  // fixed-seed splitmix64: hashes are reproducible
        auto next = [&]() {
            uint64_t zz = (x += 0x9E3779B97F4A7C15ull);
            zz = (zz ^ (zz >> 30)) * 0xBF58476D1CE4E5B9ull;
            zz = (zz ^ (zz >> 27)) * 0x94D049BB133111EBull;
            return zz ^ (zz >> 31);
        };
        for (auto& row : z)
            for (auto& v : row) v = next();
        flip = next();
    }
};
const Tables T;

struct Chain {
    int n = 0;
    int nlibs = 0;
    int8_t stones[NN];
};

// flood fill of the chain containing sq on board b; counts distinct empty neighbours
void chain_at(const int8_t* b, int sq, Chain& c) {
    uint8_t seen[NN] = {0};
    const int8_t color = b[sq];
    int8_t stack[NN];
    int sp = 0;
    c.n = 0;
    c.nlibs = 0;
    stack[sp++] = (int8_t)sq;
    seen[sq] = 1;
    while (sp) {
        const int s = stack[--sp];
        c.stones[c.n++] = (int8_t)s;
        for (int k = 0; k < T.nn[s]; ++k) {
            const int t = T.nbr[s][k];
            if (seen[t]) continue;
            seen[t] = 1;
            if (b[t] == color) stack[sp++] = (int8_t)t;
            else if (b[t] == BK_EMPTY) ++c.nlibs;
        }
    }
}

struct MoveResult {
    int8_t board[NN];
    int n_captured_dup;  // len(opp_captured) of the reference, duplicates included (go.py:413-416)
    int first_captured;
    int new_ko;
};

// place + capture + suicide test on a scratch board.  Returns 0 or BK_ILLEGAL_SUICIDE.
int try_move(const bk_pos* p, int sq, MoveResult& r) {
    const int8_t color = (p->turn & 1) ? BK_WHITE : BK_BLACK;
    const int8_t opp = color == BK_WHITE ? BK_BLACK : BK_WHITE;
    // possible_ko (go.py:461-468): every neighbour is a stone of one colour
    bool all_opp = true;
    for (int k = 0; k < T.nn[sq]; ++k)
        if (p->board[T.nbr[sq][k]] != opp) all_opp = false;
    std::memcpy(r.board, p->board, NN);
    r.board[sq] = color;
    r.n_captured_dup = 0;
    r.first_captured = -1;
    Chain ch;
    uint8_t dead[NN] = {0};
    for (int k = 0; k < T.nn[sq]; ++k) {
        const int t = T.nbr[sq][k];
        if (r.board[t] != opp) continue;
        chain_at(r.board, t, ch);  // evaluated on the board before any removal, like the reference
        if (ch.nlibs == 0) {
            if (r.first_captured < 0) r.first_captured = ch.stones[0];
            r.n_captured_dup += ch.n;
            for (int i = 0; i < ch.n; ++i) dead[ch.stones[i]] = 1;
        }
    }
    if (r.n_captured_dup)
        for (int i = 0; i < NN; ++i)
            if (dead[i]) r.board[i] = BK_EMPTY;
    r.new_ko = (r.n_captured_dup == 1 && all_opp) ? r.first_captured : BK_NO_KO;
    chain_at(r.board, sq, ch);
    return ch.nlibs == 0 ? BK_ILLEGAL_SUICIDE : 0;
}

void fresh_libs(const int8_t* b, uint8_t* libs) {
    std::memset(libs, 0, NN);
    uint8_t done[NN] = {0};
    Chain ch;
    for (int s = 0; s < NN; ++s) {
        if (b[s] == BK_EMPTY || done[s]) continue;
        chain_at(b, s, ch);
        for (int i = 0; i < ch.n; ++i) {
            libs[ch.stones[i]] = (uint8_t)ch.nlibs;
            done[ch.stones[i]] = 1;
        }
    }
}

// Game.get_liberties (go.py:220-243), cache semantics preserved
void refresh_libs(bk_pos* p) {
    if (!p->libs_valid) {
        fresh_libs(p->board, p->libs);
        p->libs_valid = 1;
        return;
    }
    const int lm = p->last_move;
    if (lm < 0 || p->libs[lm] != 0) return;
    uint8_t seen[NN] = {0};
    Chain ch;
    for (int k = 0; k <= T.nn[lm]; ++k) {
        const int s = k < T.nn[lm] ? T.nbr[lm][k] : lm;
        if (p->board[s] == BK_EMPTY || seen[s]) continue;
        chain_at(p->board, s, ch);
        for (int i = 0; i < ch.n; ++i) {
            p->libs[ch.stones[i]] = (uint8_t)ch.nlibs;
            seen[ch.stones[i]] = 1;
        }
    }
}

uint64_t full_hash(const bk_pos* p) {
    uint64_t h = 0;
    for (int s = 0; s < NN; ++s)
        if (p->board[s]) h ^= T.z[p->board[s] - 1][s];
    if (p->ko >= 0) h ^= T.z[2][p->ko];
    if (p->turn & 1) h ^= T.flip;
    return h;
}

inline uint8_t sep(int v) { return (uint8_t)(v > 6 ? 7 : v); }  // `separate`, nnet.py:253-258

typedef unsigned __int128 u128;
inline u128 bit(int s) { return (u128)1 << s; }
inline int popcnt(u128 m) { return __builtin_popcountll((uint64_t)m) + __builtin_popcountll((uint64_t)(m >> 64)); }

struct NbrMasks {
    u128 m[NN];
    NbrMasks() {
        for (int s = 0; s < NN; ++s) {
            m[s] = 0;
            for (int k = 0; k < T.nn[s]; ++k) m[s] |= bit(T.nbr[s][k]);
        }
    }
};
const NbrMasks NBR;

// chains of a position as bit masks: stones, all neighbouring points, liberties
struct Groups {
    int8_t gid[NN];
    u128 stones[NN], nbrs[NN], libs[NN];
    int size[NN];
    u128 empty;
    explicit Groups(const int8_t* b) {
        empty = 0;
        for (int s = 0; s < NN; ++s) {
            gid[s] = -1;
            if (b[s] == BK_EMPTY) empty |= bit(s);
        }
        int ng = 0;
        int8_t stack[NN];
        for (int s = 0; s < NN; ++s) {
            if (b[s] == BK_EMPTY || gid[s] >= 0) continue;
            const int g = ng++;
            u128 st = 0, nb = 0;
            int sp = 0, n = 0;
            stack[sp++] = (int8_t)s;
            gid[s] = (int8_t)g;
            while (sp) {
                const int q = stack[--sp];
                st |= bit(q);
                nb |= NBR.m[q];
                ++n;
                for (int k = 0; k < T.nn[q]; ++k) {
                    const int t = T.nbr[q][k];
                    if (b[t] == b[s] && gid[t] < 0) { gid[t] = (int8_t)g; stack[sp++] = (int8_t)t; }
                }
            }
            stones[g] = st;
            nbrs[g] = nb;
            libs[g] = nb & empty;
            size[g] = n;
        }
    }
};

// nnet.features() (nnet.py:182-262).  Per candidate move the reference places the stone, removes
// captured chains and flood-fills the new chain (nnet.py:241-247); here every chain's stone /
// neighbour / liberty set is computed once as a 128-bit mask and a move is a few mask operations:
//   captured   = opponent neighbour chains whose only liberty is the move
//   new chain  = move + own neighbour chains
//   liberties  = (empty neighbours of the new chain, minus the move) + captured points touching it
template <typename O>
void features_impl(bk_pos* p, O* out, int fresh) {
    for (int i = 0; i < 27 * NN; ++i) out[i] = 0;
    const int8_t me = (p->turn & 1) ? BK_WHITE : BK_BLACK;
    const int8_t opp = me == BK_WHITE ? BK_BLACK : BK_WHITE;
    uint8_t libs_fresh[NN];
    const uint8_t* libs;
    if (fresh) {
        fresh_libs(p->board, libs_fresh);
        libs = libs_fresh;
    } else {
        refresh_libs(p);
        libs = p->libs;
    }
    for (int s = 0; s < NN; ++s) {
        const int8_t b = p->board[s];
        if (b == me) out[0 * NN + s] = 1;
        else if (b != BK_EMPTY) out[1 * NN + s] = 1;
        else out[2 * NN + s] = 1;
        if (me == BK_BLACK) out[3 * NN + s] = 1;
        const int lv = libs[s];
        if (lv) out[(6 + (lv > 6 ? 6 : lv - 1)) * NN + s] = sep(lv);
    }
    if (p->last_move >= 0) out[4 * NN + p->last_move] = 1;
    const Groups G(p->board);
    for (int s = 0; s < NN; ++s) {
        if (p->board[s] != BK_EMPTY || s == p->ko) continue;
        const u128 me_bit = bit(s);
        u128 chain_nbrs = NBR.m[s], lib = NBR.m[s] & G.empty, captured = 0;
        int cap_dup = 0;  // the reference counts a captured chain once per point at which it touches the move
        for (int k = 0; k < T.nn[s]; ++k) {
            const int t = T.nbr[s][k];
            const int g = G.gid[t];
            if (p->board[t] == opp) {
                if (G.libs[g] == me_bit) { captured |= G.stones[g]; cap_dup += G.size[g]; }
            } else if (p->board[t] == me) {
                chain_nbrs |= G.nbrs[g];
                lib |= G.libs[g];
            }
        }
        lib = (lib & ~me_bit) | (captured & chain_nbrs);
        const int la = popcnt(lib);
        if (la == 0) continue;  // suicide
        out[5 * NN + s] = 1;
        out[(13 + (la > 6 ? 6 : la - 1)) * NN + s] = sep(la);
        if (cap_dup) out[(20 + (cap_dup > 6 ? 6 : cap_dup - 1)) * NN + s] = sep(cap_dup);
    }
}

}  // namespace

extern "C" {

int bk_go_abi_version(void) { return 6; }  // 6: bk_pool_set_lanes, BK_POOLS_MAX_INFLIGHT; 5: bk_pool_game_stats, bk_pool_snapshot / _restore, bk_pools_run, bk_normalise_rows; 4: bk_search_params.simulate/use_value/value_weight, bk_pool_node_q, bk_pos_possible_eye; 3: bk_search_params.branch_num, bk_pool_set_dedup, bk_team_selftest_concurrent; 2: request_tasks, eager_top, request_steps, tree views

void bk_pos_init(bk_pos* p) {
    std::memset(p, 0, sizeof(*p));
    p->ko = BK_NO_KO;
    p->last_move = BK_NO_MOVE;
}

int bk_pos_from_board(bk_pos* p, const char* s, int ko, int last_move, int turn) {
    bk_pos_init(p);
    for (int i = 0; i < NN; ++i) {
        if (s[i] == 'X') p->board[i] = BK_BLACK;
        else if (s[i] == 'O') p->board[i] = BK_WHITE;
        else if (s[i] == '.' || s[i] == '+') p->board[i] = BK_EMPTY;
        else return BK_ILLEGAL_OFF_BOARD;
    }
    if (ko < -1 || ko >= NN || last_move >= NN || (last_move < 0 && last_move != BK_PASS && last_move != BK_NO_MOVE))
        return BK_ILLEGAL_OFF_BOARD;
    p->ko = (int16_t)ko;
    p->last_move = (int16_t)last_move;
    p->turn = turn;
    p->hash = full_hash(p);
    return 0;
}

void bk_pos_board_string(const bk_pos* p, char out[82]) {
    for (int i = 0; i < NN; ++i) out[i] = p->board[i] == BK_BLACK ? 'X' : p->board[i] == BK_WHITE ? 'O' : '.';
    out[NN] = 0;
}

int bk_pos_play(bk_pos* p, int move) {
    if (move == BK_PASS) {  // play_pass, go.py:109-121 (does not touch the liberty cache)
        if (p->ko >= 0) p->hash ^= T.z[2][p->ko];
        p->hash ^= T.flip;
        p->turn += 1;
        p->ko = BK_NO_KO;
        p->last_move = BK_PASS;
        return 0;
    }
    if (move < 0 || move >= NN) return BK_ILLEGAL_OFF_BOARD;
    if (move == p->ko) return BK_ILLEGAL_KO;
    if (p->board[move] != BK_EMPTY) return BK_ILLEGAL_NOT_EMPTY;
    MoveResult r;
    const int rc = try_move(p, move, r);
    if (rc) return rc;
    refresh_libs(p);  // go.py:160: the cache is refreshed on the position BEFORE the move
    const int8_t color = (p->turn & 1) ? BK_WHITE : BK_BLACK;
    uint64_t h = p->hash ^ T.z[color - 1][move];
    if (p->ko >= 0) h ^= T.z[2][p->ko];
    if (r.new_ko >= 0) h ^= T.z[2][r.new_ko];
    if (r.n_captured_dup)
        for (int i = 0; i < NN; ++i)
            if (p->board[i] != BK_EMPTY && r.board[i] == BK_EMPTY) h ^= T.z[p->board[i] - 1][i];
    h ^= T.flip;
    std::memcpy(p->board, r.board, NN);
    p->hash = h;
    p->last_move = (int16_t)move;
    p->ko = (int16_t)r.new_ko;
    p->turn += 1;
    return 0;
}

int bk_pos_is_legal(const bk_pos* p, int move) {
    if (move == BK_PASS) return 1;
    if (move < 0 || move >= NN || p->board[move] != BK_EMPTY || move == p->ko) return 0;
    MoveResult r;
    return try_move(p, move, r) == 0;
}

int bk_pos_legal_moves(const bk_pos* p, uint8_t legal[81]) {
    int n = 0;
    for (int s = 0; s < NN; ++s) {
        legal[s] = (uint8_t)bk_pos_is_legal(p, s);
        n += legal[s];
    }
    return n;
}

void bk_pos_liberties(bk_pos* p, uint8_t out[81]) {
    refresh_libs(p);
    std::memcpy(out, p->libs, NN);
}

float bk_pos_area_score(const bk_pos* p, float komi) {
    // Tromp-Taylor area: stones + empty regions bordered by one colour only
    int black = 0, white = 0;
    uint8_t seen[NN] = {0};
    for (int s = 0; s < NN; ++s) {
        if (p->board[s] == BK_BLACK) ++black;
        else if (p->board[s] == BK_WHITE) ++white;
        else if (!seen[s]) {
            int8_t stack[NN];
            int sp = 0, size = 0;
            bool tb = false, tw = false;
            stack[sp++] = (int8_t)s;
            seen[s] = 1;
            while (sp) {
                const int q = stack[--sp];
                ++size;
                for (int k = 0; k < T.nn[q]; ++k) {
                    const int t = T.nbr[q][k];
                    if (p->board[t] == BK_BLACK) tb = true;
                    else if (p->board[t] == BK_WHITE) tw = true;
                    else if (!seen[t]) { seen[t] = 1; stack[sp++] = (int8_t)t; }
                }
            }
            if (tb && !tw) black += size;
            else if (tw && !tb) white += size;
        }
    }
    return (float)black - ((float)white + komi);
}

float bk_pos_score(const bk_pos* p, float komi) {
    // Game.score (go.py:202-218) restated literally, including what it does beyond Tromp-Taylor:
    // regions are processed in board order on a board that is repainted as it goes, and a
    // region's BORDER stones are repainted too -- with '?' when the region touches both colours,
    // so stones next to neutral points drop out of the count and later regions see the '?'.
    enum { Q = 3 };
    int8_t b[NN];
    std::memcpy(b, p->board, NN);
    for (;;) {
        int first = -1;
        for (int s = 0; s < NN; ++s)
            if (b[s] == BK_EMPTY) { first = s; break; }
        if (first < 0) break;
        uint8_t in_region[NN] = {0}, is_border[NN] = {0};
        int8_t stack[NN];
        int sp = 0;
        stack[sp++] = (int8_t)first;
        in_region[first] = 1;
        bool tb = false, tw = false;
        while (sp) {
            const int q = stack[--sp];
            for (int k = 0; k < T.nn[q]; ++k) {
                const int t = T.nbr[q][k];
                if (b[t] == BK_EMPTY) {
                    if (!in_region[t]) { in_region[t] = 1; stack[sp++] = (int8_t)t; }
                } else {
                    is_border[t] = 1;
                    if (b[t] == BK_BLACK) tb = true;
                    else if (b[t] == BK_WHITE) tw = true;
                }
            }
        }
        const int8_t paint = (tb && !tw) ? BK_BLACK : (tw && !tb) ? BK_WHITE : (int8_t)Q;
        for (int s = 0; s < NN; ++s)
            if (in_region[s] || is_border[s]) b[s] = paint;
    }
    int black = 0, white = 0;
    for (int s = 0; s < NN; ++s) {
        black += b[s] == BK_BLACK;
        white += b[s] == BK_WHITE;
    }
    return (float)black - ((float)white + komi);
}

int bk_pos_eye_like(const bk_pos* p, int sq, int color) {
    if (sq < 0 || sq >= NN || p->board[sq] != BK_EMPTY) return 0;
    for (int k = 0; k < T.nn[sq]; ++k)
        if (p->board[T.nbr[sq][k]] != color) return 0;
    return 1;
}

// go.possible_eye (go.py:470-485) as the reference computes it, table bug included: DIAGONALS (go.py:372-373) lists
// (x+1,y+1), (x+1,y-1), (x-1,y-1) and (x-1,y-1) AGAIN -- (x-1,y+1) is never looked at and (x-1,y-1) counts twice.
// Returns the colour of the one-point eye at sq (BK_BLACK / BK_WHITE) or 0.
int bk_pos_possible_eye(const bk_pos* p, int sq) {
    if (sq < 0 || sq >= NN || p->board[sq] == BK_BLACK || p->board[sq] == BK_WHITE) return 0;
    const int8_t color = p->board[T.nbr[sq][0]];           // possible_ko (go.py:461-468): one colour all around, no empties
    if (color != BK_BLACK && color != BK_WHITE) return 0;
    for (int k = 1; k < T.nn[sq]; ++k)
        if (p->board[T.nbr[sq][k]] != color) return 0;
    const int x = sq / N, y = sq % N;
    static const int dx[4] = {1, 1, -1, -1}, dy[4] = {1, -1, -1, -1};
    int on_board = 0, faults = 0;
    for (int k = 0; k < 4; ++k) {
        const int a = x + dx[k], b = y + dy[k];
        if (a < 0 || a >= N || b < 0 || b >= N) continue;
        ++on_board;
        const int8_t c = p->board[a * N + b];
        if (c == BK_BLACK || c == BK_WHITE) faults += c != color;
    }
    if (on_board < 4) ++faults;
    return faults > 1 ? 0 : color;
}

void bk_pos_features_u8(bk_pos* p, uint8_t out[2187], int fresh) { features_impl(p, out, fresh); }
void bk_pos_features_f32(bk_pos* p, float out[2187], int fresh) { features_impl(p, out, fresh); }

void bk_features_batch_u8(void* pos, int n, int stride, uint8_t* out, int fresh) {
    for (int i = 0; i < n; ++i)
        features_impl(reinterpret_cast<bk_pos*>(static_cast<char*>(pos) + (size_t)i * stride), out + (size_t)i * 2187, fresh);
}

// every legal successor, one bk_pos_play each: the definition (and what tests compare bk_pos_children with)
int bk_pos_children_slow(const bk_pos* p, bk_pos* out, int16_t* moves) {
    int n = 0;
    for (int s = 0; s < NN; ++s) {
        if (p->board[s] != BK_EMPTY || s == p->ko) continue;
        out[n] = *p;
        if (bk_pos_play(&out[n], s) == 0) moves[n++] = (int16_t)s;
    }
    return n;
}

// The same successors without a flood fill per candidate and neighbour (an expansion of the tree search builds ~75 of them:
// chain_at was 16 % of the host side of self-play).  The position's chains and their exact liberty counts are found once;
// the liberty cache is refreshed once (every successor inherits the parent's cache as refreshed before the move, go.py:160).
// A move next to no opponent chain in atari captures nothing: it is legal iff the point has an empty neighbour or joins a chain of
// its own colour that has another liberty, the new board is the old one plus the stone, and no ko arises (a ko needs exactly one
// captured stone).  Only moves that may capture take the general path.
int bk_pos_children(const bk_pos* p, bk_pos* out, int16_t* moves) {
    bk_pos base = *p;
    refresh_libs(&base);
    int8_t cid[NN];
    uint8_t clibs[NN];
    std::memset(cid, -1, NN);
    int nc = 0;
    Chain ch;
    for (int s = 0; s < NN; ++s) {
        if (base.board[s] == BK_EMPTY || cid[s] >= 0) continue;
        chain_at(base.board, s, ch);
        for (int i = 0; i < ch.n; ++i) cid[ch.stones[i]] = (int8_t)nc;
        clibs[nc++] = (uint8_t)ch.nlibs;
    }
    const int8_t color = (base.turn & 1) ? BK_WHITE : BK_BLACK;
    const int8_t opp = color == BK_WHITE ? BK_BLACK : BK_WHITE;
    uint64_t h0 = base.hash ^ T.flip;
    if (base.ko >= 0) h0 ^= T.z[2][base.ko];
    int n = 0;
    for (int s = 0; s < NN; ++s) {
        if (base.board[s] != BK_EMPTY || s == base.ko) continue;
        bool may_capture = false, alive = false;
        for (int k = 0; k < T.nn[s]; ++k) {
            const int t = T.nbr[s][k];
            const int8_t b = base.board[t];
            if (b == BK_EMPTY) alive = true;
            else if (b == opp) may_capture = may_capture || clibs[cid[t]] == 1;
            else alive = alive || clibs[cid[t]] >= 2;
        }
        if (may_capture) {
            out[n] = base;
            if (bk_pos_play(&out[n], s) == 0) moves[n++] = (int16_t)s;
            continue;
        }
        if (!alive) continue;                       // suicide
        bk_pos& c = out[n];
        c = base;
        c.board[s] = color;
        c.hash = h0 ^ T.z[color - 1][s];
        c.last_move = (int16_t)s;
        c.ko = BK_NO_KO;
        c.turn += 1;
        moves[n++] = (int16_t)s;
    }
    return n;
}

}  // extern "C"
