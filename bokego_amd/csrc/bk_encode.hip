// bk_encode.hip -- the 27 feature planes of nnet.features() (bokego/nnet.py:182-262) computed on the GPU
// from 192-byte position records (include/bokego_go.h: bk_pos), so that the host ships 192 B per leaf
// instead of 2,187 B of planes and spends no time encoding.
//
// Division of labour with the host (libbkgo): the reference's liberty planes 6-12 come from Game._libs, a
// lazily refreshed, history-dependent cache (go.py:220-243).  That refresh is the stateful part and stays on
// the host (bk_pos_liberties / bk_pool_collect_pos run it before the record is copied); everything else is a
// pure function of (board, libs, ko, last_move, turn) and is computed here.
//
// Mapping: one workgroup = THREE positions, 256 threads, one thread per board point (243 active) -- the same 3-board
// packing as the leaf kernel.  No LDS atomics, no data-dependent workgroup barriers (two fixed ones):
//   1. bitboards.  Every wave ballots "black here" / "white here"; each thread then cuts its position's 81 bits out
//      of the workgroup's 256-bit ballots as three 27-bit words (3 board rows per word: +-9 is a shift by 9 with a
//      carry between words, +-1 a shift by 1 under a column mask);
//   2. chains by flood fill IN REGISTERS: a stone's thread grows its own chain from its own bit
//      (x |= dilate(x) & own_colour until nothing changes; the loop is wave-uniform via __any) and publishes
//      {chain mask, the chain's single liberty or none} as one 16-byte LDS record;
//   3. per empty point: the move's captures / new-chain liberties as mask operations on the <= 4 neighbouring
//      chains' records (the same algebra as features_impl in bk_go.cpp, including the reference's
//      per-touching-point double count of captured chains, go.py:413-416);
//   4. 27 byte planes per position, written plane-major ([27][81], coalesced over the points).
// Integer/byte work: HBM-bound by 192 B in + 2,187 B out per position; no MFMA.  4 KB of LDS and < 64 VGPRs per
// workgroup, so it still fits beside the leaf kernel's 3-board workgroups when launched on the copy-in stream.
#include "bk_encode_dev.h"

namespace {

using namespace bk_enc;

__global__ void __launch_bounds__(256) bk_encode_kernel(const unsigned char* __restrict__ pos, int B,
                                                       unsigned char* __restrict__ planes) {
    __shared__ EncLds S;
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * PPW;
    unsigned char v[27];
    bool live;
    int p, q;
    encode_points(pos + (size_t)b0 * BK_POS_BYTES, B - b0, tid, S, v, live, p, q);

    // ---- 4. planes out, plane-major: one byte per point and plane (a wave's store covers 64 consecutive bytes).
    //         Measured alternatives, both slower: staging the planes in LDS and writing aligned 16-byte chunks, one
    //         position per round in 4 KB (14.7 us per 4,096 positions) or all three at once in 6.6 KB (14.6 us), against
    //         13.0 us for these direct byte stores (the 27 ds_write_b8 + barriers cost more than the 27 store instructions).
    if (live) {
        unsigned char* out = planes + (size_t)(b0 + p) * 2187 + q;
#pragma unroll
        for (int i = 0; i < 27; ++i) {
#if BK_ENC_EXP == 1
            if (i > 0) break;   // experiment: one plane only
#endif
            out[i * NN] = v[i];
        }
    }
}

}  // namespace

hipError_t bk_launch_encode(const void* d_pos, int B, uint8_t* d_planes, hipStream_t stream) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(bk_encode_kernel, dim3((B + PPW - 1) / PPW), dim3(256), 0, stream, static_cast<const unsigned char*>(d_pos), B,
                       d_planes);
    return hipGetLastError();
}
