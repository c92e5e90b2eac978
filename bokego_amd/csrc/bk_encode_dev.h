// bk_encode_dev.h -- the feature encoder's device code (see bk_encode.hip for what it computes and how), shared by the encoder kernel
// and by the leaf kernels' staging of SMALL requests: there the planes are computed from the 192-byte position records inside the
// leaf kernel itself (bk_kernels.hip, stage_positions) -- one launch and one dependent dispatch less on the critical path of a
// one-tree search's request (round 6).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bk_internal.h"

#ifndef BK_ENC_EXP
#define BK_ENC_EXP 0   // timing experiments only (wrong results): 1 = one plane stored, 2 = one flood round
#endif

namespace bk_enc {


constexpr int NN = 81;
constexpr int PPW = 3;                    // positions per workgroup
constexpr unsigned M27 = 0x7FFFFFFu;      // one word = 3 board rows
constexpr unsigned NC0 = 0x1FEu | (0x1FEu << 9) | (0x1FEu << 18);  // points whose column is not 0
constexpr unsigned NC8 = 0x0FFu | (0x0FFu << 9) | (0x0FFu << 18);  // points whose column is not 8

struct BB {  // 81-point set: word k = rows 3k..3k+2, bit = 9*(row%3) + col
    unsigned w[3];
};
__device__ __forceinline__ BB operator|(BB a, BB b) { return {{a.w[0] | b.w[0], a.w[1] | b.w[1], a.w[2] | b.w[2]}}; }
__device__ __forceinline__ BB operator&(BB a, BB b) { return {{a.w[0] & b.w[0], a.w[1] & b.w[1], a.w[2] & b.w[2]}}; }
__device__ __forceinline__ BB operator~(BB a) { return {{~a.w[0] & M27, ~a.w[1] & M27, ~a.w[2] & M27}}; }
__device__ __forceinline__ bool operator!=(BB a, BB b) { return ((a.w[0] ^ b.w[0]) | (a.w[1] ^ b.w[1]) | (a.w[2] ^ b.w[2])) != 0; }
__device__ __forceinline__ int popc(BB a) { return __popc(a.w[0]) + __popc(a.w[1]) + __popc(a.w[2]); }
// all points adjacent to a point of x (go.py:375-383)
__device__ __forceinline__ BB dilate(BB x) {
    BB d;
    d.w[0] = ((x.w[0] << 9) & M27) | (x.w[0] >> 9) | ((x.w[1] << 18) & M27) | ((x.w[0] << 1) & NC0) | ((x.w[0] >> 1) & NC8);
    d.w[1] = ((x.w[1] << 9) & M27) | (x.w[0] >> 18) | (x.w[1] >> 9) | ((x.w[2] << 18) & M27) | ((x.w[1] << 1) & NC0) | ((x.w[1] >> 1) & NC8);
    d.w[2] = ((x.w[2] << 9) & M27) | (x.w[1] >> 18) | (x.w[2] >> 9) | ((x.w[2] << 1) & NC0) | ((x.w[2] >> 1) & NC8);
    return d;
}
__device__ __forceinline__ BB single(int k, unsigned bit) {  // no dynamic register indexing
    return {{k == 0 ? bit : 0u, k == 1 ? bit : 0u, k == 2 ? bit : 0u}};
}
__device__ __forceinline__ unsigned word_of(BB a, int k) { return k == 0 ? a.w[0] : k == 1 ? a.w[1] : a.w[2]; }

struct EncLds {
    unsigned bal[2][8];          // black / white ballots of the 4 waves = 256 bits each
    uint4 chain[PPW][NN];        // per stone: its chain's mask (x,y,z) and the chain's only liberty (w; 255: not exactly one)
};


// The 27 plane values of board point q of position p of a workgroup's (up to) three positions, for the threads tid < 243 of the
// workgroup's first 256 (`live`: this thread has a point of one of the nb positions that exist); every thread of the workgroup
// must call it (two workgroup barriers inside).  pos: the record of the workgroup's first position.
__device__ __forceinline__ void encode_points(const unsigned char* __restrict__ pos, int nb, int tid, EncLds& S, unsigned char (&v)[27],
                                              bool& live, int& p, int& q) {
    const bool worker = tid < 256;
    p = tid / NN;                        // position inside the workgroup (3: the 13 spare threads; >= 3: the waves beyond the fourth)
    q = tid - NN * p;                    // board point
    live = worker && p < PPW && p < nb;
    const int r = q / 9, c = q - 9 * r;
    const int k = r / 3;                 // bitboard word and bit of this point
    const unsigned bit = 1u << ((q - 27 * k) & 31);

    int me_board = 0, my_libs = 0, ko = -1, last_move = -3, turn = 0;
    if (live) {
        const unsigned char* src = pos + (size_t)p * BK_POS_BYTES;
        me_board = (signed char)src[q];
        my_libs = src[81 + q];
        const unsigned kl = *reinterpret_cast<const unsigned*>(src + 164);   // ko | last_move << 16 (records are 192-B aligned)
        ko = (short)(kl & 0xffffu);
        last_move = (short)(kl >> 16);
        turn = *reinterpret_cast<const int*>(src + 172);
    }
    // ---- 1. bitboards from wave ballots ----
    const unsigned long long bb = __ballot(me_board == 1), bw = __ballot(me_board == 2);
    if (worker && (tid & 63) == 0) {
        const int w = tid >> 6;
        S.bal[0][2 * w] = (unsigned)bb; S.bal[0][2 * w + 1] = (unsigned)(bb >> 32);
        S.bal[1][2 * w] = (unsigned)bw; S.bal[1][2 * w + 1] = (unsigned)(bw >> 32);
    }
    __syncthreads();
    BB black, white;
    {
        const int pp = p < PPW ? p : 0;
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
            const int off = NN * pp + 27 * kk, i = off >> 5, sh = off & 31;     // off + 27 <= 243: i + 1 <= 7
            const unsigned long long b2 = ((unsigned long long)S.bal[0][i + 1] << 32) | S.bal[0][i];
            const unsigned long long w2 = ((unsigned long long)S.bal[1][i + 1] << 32) | S.bal[1][i];
            black.w[kk] = (unsigned)(b2 >> sh) & M27;
            white.w[kk] = (unsigned)(w2 >> sh) & M27;
        }
    }
    const BB empty = ~(black | white);

    // ---- 2. this stone's chain, grown in registers ----
    const BB seed = single(k, bit);
    BB x = seed;
    {
        const BB own = me_board == 1 ? black : white;
        bool changed = live && me_board != 0;
        BB d = dilate(x);                  // kept in step with x: after the loop it is the chain's neighbourhood
        for (;;) {                         // wave-uniform exit: at most 81 rounds, normally a handful
            const BB nx = (x | d) & own;
            changed = changed && (nx != x);
            if (!__any(changed)) break;
            if (changed) x = nx;
            d = dilate(x);
#if BK_ENC_EXP == 2
            break;
#endif
        }
        if (live && me_board != 0) {
            const BB lib = d & empty;
            unsigned only = 255u;
            if (popc(lib) == 1)
                only = lib.w[0] ? __ffs(lib.w[0]) - 1 : lib.w[1] ? 27 + __ffs(lib.w[1]) - 1 : 54 + __ffs(lib.w[2]) - 1;
            S.chain[p][q] = make_uint4(x.w[0], x.w[1], x.w[2], only);
        }
    }
    __syncthreads();

    // ---- 3. this point's 27 plane values ----
#pragma unroll
    for (int i = 0; i < 27; ++i) v[i] = 0;
    if (live) {
        const int nbr[4] = {q + 9, q - 9, q + 1, q - 1};
        const bool nv[4] = {r + 1 < 9, r >= 1, c + 1 < 9, c >= 1};
        const int me = (turn & 1) ? 2 : 1, opp = 3 - me;
        if (me_board == me) v[0] = 1;
        else if (me_board != 0) v[1] = 1;
        else v[2] = 1;
        if (me == 1) v[3] = 1;
        if (q == last_move) v[4] = 1;
        int p_lib = -1, lib_val = 0, p_la = -1, la_val = 0, p_cap = -1, cap_val = 0;
        if (my_libs) { p_lib = 6 + (my_libs > 6 ? 6 : my_libs - 1); lib_val = my_libs > 6 ? 7 : my_libs; }
        bool legal = false;
        if (me_board == 0 && q != ko) {
            BB newchain = seed, cap{{0, 0, 0}};
            int cap_dup = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (!nv[j]) continue;
                const int t = nbr[j], tk = t / 27;
                const unsigned tb = 1u << (t - 27 * tk);
                const int bt = (word_of(black, tk) & tb) ? 1 : (word_of(white, tk) & tb) ? 2 : 0;
                if (bt == 0) continue;
                const uint4 ch = S.chain[p][t];
                const BB cx{{ch.x, ch.y, ch.z}};
                if (bt == opp) {
                    if (ch.w == (unsigned)q) {   // captured iff the chain's only liberty is this point
                        cap = cap | cx;
                        cap_dup += popc(cx);     // once per touching point, as the reference counts (go.py:413-416)
                    }
                } else {
                    newchain = newchain | cx;
                }
            }
            const BB d = dilate(newchain);
            const BB lib = (d & empty & ~seed) | (cap & d);
            const int la = popc(lib);
            if (la) {  // la == 0: suicide
                legal = true;
                p_la = 13 + (la > 6 ? 6 : la - 1);
                la_val = la > 6 ? 7 : la;
                if (cap_dup) { p_cap = 20 + (cap_dup > 6 ? 6 : cap_dup - 1); cap_val = cap_dup > 6 ? 7 : cap_dup; }
            }
        }
        if (legal) v[5] = 1;
#pragma unroll
        for (int i = 6; i < 27; ++i) {
            if (i == p_lib) v[i] = (unsigned char)lib_val;
            if (i == p_la) v[i] = (unsigned char)la_val;
            if (i == p_cap) v[i] = (unsigned char)cap_val;
        }
    }
}

}  // namespace bk_enc
