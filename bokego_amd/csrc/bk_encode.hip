// bk_encode.hip -- the 27 feature planes of nnet.features() (bokego/nnet.py:182-262) computed on the GPU
// from 192-byte position records (include/bokego_go.h: bk_pos), so that the host ships 192 B per leaf
// instead of 2,187 B of planes and spends no time encoding.
//
// Division of labour with the host (libbkgo): the reference's liberty planes 6-12 come from Game._libs, a
// lazily refreshed, history-dependent cache (go.py:220-243).  That refresh is the stateful part and stays on
// the host (bk_pos_liberties / bk_pool_collect_pos run it before the record is copied); everything else is a
// pure function of (board, libs, ko, last_move, turn) and is computed here.
//
// Mapping: one workgroup = one position, 128 threads, one thread per board point (81 active).  The workgroup
// needs 2.8 KB of LDS and few registers on purpose: it is launched on the engine's copy-in stream, so it runs
// *under* the previous request's leaf kernel, whose 3-board workgroups leave 4.6 KB of LDS per CU free.
//   1. chains by label propagation in LDS (label = smallest point index of the chain, min over same-colour
//      neighbours + pointer jumping, until no label changes);
//   2. per chain: stone mask, neighbour mask (96-bit, three LDS words) and size, by LDS atomics;
//   3. per empty point: the move's captures / new-chain liberties as mask operations on the <= 4 neighbouring
//      chains (the same algebra as features_impl in bk_go.cpp, including the reference's per-touching-point
//      double count of captured chains, go.py:413-416);
//   4. 27 byte planes per position, written plane-major ([27][81], coalesced over the points).
// Integer/byte work: HBM-bound by 192 B in + 2,187 B out per position; no MFMA.
#include "bk_internal.h"

namespace {

constexpr int NN = 81;

struct Mask {
    unsigned w[3];
};

__device__ __forceinline__ void mask_set(Mask& m, int s) {  // no dynamic register indexing
    const unsigned bit = 1u << (s & 31);
#pragma unroll
    for (int i = 0; i < 3; ++i) m.w[i] |= (s >> 5) == i ? bit : 0u;
}
__device__ __forceinline__ int mask_pop(const Mask& m) { return __popc(m.w[0]) + __popc(m.w[1]) + __popc(m.w[2]); }

struct PosLds {
    signed char board[84];
    unsigned char libs[84];
    int label[NN];
    unsigned stones[NN][3];
    unsigned nbrs[NN][3];
    int size[NN];
    unsigned empty[3];
    int ko, last_move, turn;
};

__global__ void __launch_bounds__(128) bk_encode_kernel(const unsigned char* __restrict__ pos, int B,
                                                       unsigned char* __restrict__ planes) {
    __shared__ PosLds P;
    const int q = threadIdx.x;          // board point (lanes 81..127 idle, they only take part in the barriers)
    const int b = blockIdx.x;
    const bool live = q < NN && b < B;
    const int r = q / 9, c = q - 9 * r;

    // neighbours (go.py:375-383 order is irrelevant here: only sums / unions are formed); fixed slots with a
    // validity mask so that nothing is indexed dynamically
    const int nb[4] = {q + 9, q - 9, q + 1, q - 1};
    const bool nv[4] = {r + 1 < 9, r >= 1, c + 1 < 9, c >= 1};
    constexpr int nn = 4;

    int me_board = 0, my_libs = 0;
    if (live) {
        const unsigned char* src = pos + (size_t)b * 192;
        me_board = (signed char)src[q];
        my_libs = src[81 + q];
        P.board[q] = (signed char)me_board;
        P.libs[q] = (unsigned char)my_libs;
        P.label[q] = me_board ? q : -1;
        P.stones[q][0] = P.stones[q][1] = P.stones[q][2] = 0;
        P.nbrs[q][0] = P.nbrs[q][1] = P.nbrs[q][2] = 0;
        P.size[q] = 0;
        if (q < 3) P.empty[q] = 0;
        if (q == 0) {
            P.ko = (short)(src[164] | (src[165] << 8));
            P.last_move = (short)(src[166] | (src[167] << 8));
            P.turn = (int)(src[172] | (src[173] << 8) | (src[174] << 16) | ((unsigned)src[175] << 24));
        }
    }
    __syncthreads();

    // ---- 1. chain labels ----
    for (;;) {
        int changed = 0;
        if (live && me_board) {
            int m = P.label[q];
#pragma unroll
            for (int k = 0; k < nn; ++k)
                if (nv[k] && P.board[nb[k]] == me_board) m = min(m, P.label[nb[k]]);
            m = min(m, P.label[m]);  // pointer jumping: labels only ever point to smaller indices of the same chain
            if (m < P.label[q]) { P.label[q] = m; changed = 1; }
        }
        // racy reads of neighbouring labels only ever see values that are valid (smaller-or-equal, same chain)
        if (!__syncthreads_or(changed)) break;
    }

    // ---- 2. per-chain masks ----
    if (live) {
        if (me_board) {
            const int g = P.label[q];
            atomicOr(&P.stones[g][q >> 5], 1u << (q & 31));
#pragma unroll
            for (int k = 0; k < nn; ++k)
                if (nv[k]) atomicOr(&P.nbrs[g][nb[k] >> 5], 1u << (nb[k] & 31));
            atomicAdd(&P.size[g], 1);
        } else {
            atomicOr(&P.empty[q >> 5], 1u << (q & 31));
        }
    }
    __syncthreads();

    // ---- 3/4. planes ----
    if (!live) return;
    unsigned char* out = planes + (size_t)b * 2187 + q;
    const int me = (P.turn & 1) ? 2 : 1, opp = 3 - me;
    unsigned char v[27];
#pragma unroll
    for (int i = 0; i < 27; ++i) v[i] = 0;
    if (me_board == me) v[0] = 1;
    else if (me_board != 0) v[1] = 1;
    else v[2] = 1;
    if (me == 1) v[3] = 1;
    if (q == P.last_move) v[4] = 1;
    int p_lib = -1, lib_val = 0, p_la = -1, la_val = 0, p_cap = -1, cap_val = 0;
    if (my_libs) { p_lib = 6 + (my_libs > 6 ? 6 : my_libs - 1); lib_val = my_libs > 6 ? 7 : my_libs; }
    bool legal = false;
    if (me_board == 0 && q != P.ko) {
        Mask nbm{{0, 0, 0}}, lib{{0, 0, 0}}, cap{{0, 0, 0}};
#pragma unroll
        for (int k = 0; k < nn; ++k)
            if (nv[k]) mask_set(nbm, nb[k]);
        for (int i = 0; i < 3; ++i) lib.w[i] = nbm.w[i] & P.empty[i];
        Mask chain_nbrs = nbm;
        int cap_dup = 0;
#pragma unroll
        for (int k = 0; k < nn; ++k) {
            if (!nv[k]) continue;
            const int t = nb[k], bt = P.board[t];
            if (bt == 0) continue;
            const int g = P.label[t];
            if (bt == opp) {
                // captured iff the chain's only liberty is this point
                Mask gl;
                for (int i = 0; i < 3; ++i) gl.w[i] = P.nbrs[g][i] & P.empty[i];
                bool only_me = true;
                for (int i = 0; i < 3; ++i) only_me &= gl.w[i] == ((q >> 5) == i ? 1u << (q & 31) : 0u);
                if (only_me) {
                    for (int i = 0; i < 3; ++i) cap.w[i] |= P.stones[g][i];
                    cap_dup += P.size[g];
                }
            } else {
                for (int i = 0; i < 3; ++i) {
                    chain_nbrs.w[i] |= P.nbrs[g][i];
                    lib.w[i] |= P.nbrs[g][i] & P.empty[i];
                }
            }
        }
        for (int i = 0; i < 3; ++i) {
            const unsigned mebit = (q >> 5) == i ? 1u << (q & 31) : 0u;
            lib.w[i] = (lib.w[i] & ~mebit) | (cap.w[i] & chain_nbrs.w[i]);
        }
        const int la = mask_pop(lib);
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
#pragma unroll
    for (int i = 0; i < 27; ++i) out[i * NN] = v[i];
}

}  // namespace

hipError_t bk_launch_encode(const void* d_pos, int B, uint8_t* d_planes, hipStream_t stream) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(bk_encode_kernel, dim3(B), dim3(128), 0, stream, static_cast<const unsigned char*>(d_pos), B,
                       d_planes);
    return hipGetLastError();
}
