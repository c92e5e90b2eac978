// bk_kernels_f16.hip -- split-precision ("f16x2") variant of the fused leaf-evaluation kernel.
//
// Same fusion as bk_kernels.hip (NB boards of one net stay in LDS through all 7 conv layers and
// the head), but the convolutions run on v_mfma_f32_32x32x16_f16, which issues 16x the MACs per
// cycle of the fp32 matrix instruction.  fp32 accuracy is kept by splitting every operand into
// an fp16 pair:  x = (x_hi + x_lo) / s  with  x_hi = fp16(s x),  x_lo = fp16(s x - x_hi)
// (22 significant bits, s a power of two), and accumulating, in fp32 inside the matrix unit,
//      w_hi x_hi + w_lo x_hi + w_hi x_lo        (the w_lo x_lo term is < 2^-22 and dropped)
// = 3 MFMAs of 32 cycles per 16-deep K step instead of 8 fp32 MFMAs of 64 cycles.
// An offline float64 emulation over the 536 golden positions puts the logit
// error this split adds at <= 1e-5, below the 5.7e-5 of the exact-fp32 kernel's summation order.
//
// Orientation is transposed w.r.t. bk_kernels.hip: the weights are the MFMA "A" operand (rows =
// couts) and the activations the "B" operand (columns = positions), so a lane's accumulator
// registers hold 4 CONSECUTIVE couts of ONE position and the epilogue writes them as one 8-byte
// hi and one 8-byte lo store.
//
// LDS layout: 512 B per position as before, now [hi plane: 16 groups x 8 fp16][lo plane: same];
// 16-byte groups XOR-swizzled by the position within each plane; lo = hi address + 256.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "bk_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

#ifndef BK_EXP
#define BK_EXP 0
#endif
#ifdef BK_STAMPS
#define STAMP(k)                                                                                   \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        unsigned long long _t;                                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");               \
        if (a.stamps && lane == 0) a.stamps[((size_t)blockIdx.x * 4 + wave) * 32 + (k)] = _t;      \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } while (0)
#else
#define STAMP(k) do {} while (0)
#endif

namespace {

// v - (float)h[0] / v - (float)h[1] as one mixed-precision FMA (fp16 source operand, fp32 result)
__device__ __forceinline__ float sub_f16_lo(float v, f16x2 h) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(v));
    return r;
}
__device__ __forceinline__ float sub_f16_hi(float v, f16x2 h) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(v));
    return r;
}

__device__ __forceinline__ int pos3(int b, int y, int x) { return (10 * b + 1 + y) * 10 + x + 1; }
__device__ __forceinline__ int pos5(int b, int y, int x) { return (11 * b + 2 + y) * 11 + x + 2; }

template <int NB>
struct Geo {
    static constexpr int MT = (81 * NB + 31) / 32;
    static constexpr int NPOS = 100 * NB + 11;
    static constexpr int NP0 = 121 * NB + 24;
    static constexpr int WM = (MT % 2 == 0) ? 2 : 1;  // wave grid over [position tiles x cout tiles]
    static constexpr int WN = 4 / WM;
    static constexpr int NT = 4 / WN;
    static constexpr int MTW = MT / WM;
    static constexpr int DUMMY_BYTES = 2048;  // 64 x 16 B slots for the hi store, the lo store lands 256 B further
    static constexpr int HS_FLOATS = NB * 96;
    static constexpr int LDS_BYTES = NPOS * 512 + DUMMY_BYTES + HS_FLOATS * 4;
};

// LDS swizzle key of an activation record: its 16-B chunks are stored at chunk ^ key.  A reader of tap
// (dy,dx) addresses the neighbour's record, whose key is the reader's own key + 9dy + dx, so a lane group
// whose rows have 16 distinct keys reads 16 distinct bank slots at every tap.
template <int NB>
__device__ __forceinline__ int act_key(int b, int y, int x) {
    return NB == 3 ? 6 * b + 9 * (y - 1) + x + 16 : 81 * b + 9 * y + x;  // only the low 4 bits are used
}

// Rows of the two edge tiles of the NB == 3 order: (board, x) of the y=0 / y=8 point handled by each of
// the 32 lanes (-1: padding row) and the row's swizzle key.  The 27 points are dealt to the lanes so that
// each ds_read_b128 lane group {0-3,12-15,20-27} / {4-11,16-19,28-31} holds 16 distinct keys, padding rows
// taking the keys left over (tools/lds_layout.py generates and checks these tables).
__device__ constexpr signed char kEdgeB[2][32] = {
    {1, 1, 1, 0, 2, 2, 2, 2, 2, 2, 2, 2, 0, 0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 1, 1, 0, -1, -1, 2, -1, -1, -1},
    {0, 0, 0, 0, 2, 2, 2, 2, 1, 1, 1, 2, 0, 0, 0, 1, 2, 2, 2, 1, 1, 1, 0, 0, 1, 2, -1, -1, 1, -1, -1, -1}};
__device__ constexpr signed char kEdgeX[2][32] = {
    {6, 7, 8, 0, 0, 1, 2, 4, 5, 6, 7, 8, 1, 2, 3, 4, 0, 1, 2, 4, 6, 7, 8, 3, 5, 5, 0, 0, 3, 0, 0, 0},
    {1, 2, 3, 4, 5, 6, 7, 8, 0, 1, 2, 0, 6, 7, 8, 6, 1, 2, 4, 3, 7, 8, 0, 5, 4, 3, 0, 0, 5, 0, 0, 0}};
__device__ constexpr signed char kEdgeKey[2][32] = {
    {3, 4, 5, 7, 3, 4, 5, 7, 8, 9, 10, 11, 8, 9, 10, 11, 13, 14, 15, 1, 13, 14, 15, 0, 2, 12, 1, 6, 6, 0, 2, 12},
    {0, 1, 2, 3, 0, 1, 2, 3, 5, 6, 7, 11, 5, 6, 7, 11, 12, 13, 15, 8, 12, 13, 15, 4, 9, 14, 8, 10, 10, 4, 9, 14}};

// Flat row index r of the workgroup's GEMM -> board point and swizzle key.  NB == 3 uses a y-major order so
// that tile 0 holds exactly the y=0 points of the three boards and tile 7 the y=8 points (27 + 5 padding
// rows each): a 3x3 tap with dy=-1 reads only zero halo for tile 0 and one with dy=+1 only zero halo for
// tile 7, so those MFMAs are skipped.  Tiles 1..6 hold the points with y in 1..7, two tiles per board
// (63 points + 1 padding row), in (y,x) order: consecutive rows have consecutive keys.
template <int NB>
__device__ __forceinline__ bool row_decode(int r, int& b, int& y, int& x, int& key) {
    bool valid;
    if (NB == 3) {
        if (r < 32 || r >= 224) {
            const int e = r >= 224, l = r & 31;
            b = kEdgeB[e][l];
            x = kEdgeX[e][l];
            key = kEdgeKey[e][l];
            y = e ? 8 : 0;
            valid = b >= 0;
        } else {
            const int i = r - 32, j = i & 63;
            b = i >> 6;
            valid = j < 63;
            y = 1 + j / 9;
            x = j - 9 * (y - 1);
            key = act_key<NB>(b, y, x);  // the padding row j = 63 continues the sequence
        }
    } else {
        valid = r < 81 * NB;
        b = r / 81;
        const int q = r - 81 * b;
        y = q / 9;
        x = q - 9 * y;
        key = act_key<NB>(b, y, x);
    }
    if (!valid) { b = 0; y = NB == 3 ? 4 : 0; x = 0; }  // padding rows compute from a valid address; never stored
    return valid;
}

// Issue order inside one K step: NMFMA matrix instructions with NREADS LDS reads (2 address VALU
// each) and NLOADS weight loads dealt out between them, so the matrix pipe never waits for a
// burst of loads.  Masks: 0x8 MFMA, 0x100 DS read, 0x20 VMEM read, 0x2 VALU.
template <int NMFMA, int NREADS, int NLOADS, int... I>
__device__ __forceinline__ void sched_pattern(std::integer_sequence<int, I...>) {
    ((__builtin_amdgcn_sched_group_barrier(0x008, NMFMA / NREADS + (I < NMFMA % NREADS ? 1 : 0), 0),
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0), __builtin_amdgcn_sched_group_barrier(0x100, 1, 0),
      __builtin_amdgcn_sched_group_barrier(0x020, I < NLOADS ? 1 : 0, 0)),
     ...);
}

// One conv layer for one wave.  acc[mt][nt]: 32 couts (rows) x 32 positions (columns).
// wl: the layer's fp16 fragments [k16 step][cout tile (4)][piece hi/lo][lane][8], 8 KiB per step.
// NPROD = 3: w_hi x_hi + w_lo x_hi + w_hi x_lo.  NPROD = 2 drops the x_lo product (and its LDS reads):
// only for layer 0 when every input plane value of the workgroup is exactly an fp16 (features()
// planes are small integers, so in practice always).
template <int NB, bool FIRST, int NPROD = 3>
__device__ __forceinline__ void conv_layer16(const char* actb, const _Float16* __restrict__ wl,
                                             f32x16 (&acc)[Geo<NB>::MTW][Geo<NB>::NT], int lane, int wm, int wn,
                                             const int (&rpos3)[Geo<NB>::MTW], const int (&rpos5)[Geo<NB>::MTW],
                                             const int (&rkey)[Geo<NB>::MTW]) {
    constexpr int MTW = Geo<NB>::MTW, NT = Geo<NB>::NT;
    constexpr int KW = FIRST ? 5 : 3;
    constexpr int TAPS = KW * KW;
    constexpr int S = FIRST ? 2 : 8;                      // k16 steps per tap (32 / 128 input channels)
    constexpr int NSTEPS = FIRST ? BK16_L0_STEPS : TAPS * S;  // layer 0 is zero-padded to a multiple of 4
    constexpr int ROWB = FIRST ? 128 : 512;               // bytes per position
    constexpr int LO = FIRST ? 64 : 256;                  // hi plane -> lo plane
    const int h = lane >> 5;

    int pbase[MTW], rrow[MTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        pbase[mt] = FIRST ? rpos5[mt] : rpos3[mt];
        rrow[mt] = rkey[mt];
    }
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

    // weights: wave-uniform base + lane offset (scalar pointer arithmetic per step)
    const _Float16* wv = wl + (wn * NT) * 1024;  // 2 pieces x 512 halfs per cout tile
    const int lane8 = lane * 8;
    auto load_w = [&](f16x8 (&W)[NT][2], int ks) {
        const _Float16* wb = wv + (size_t)ks * 4096;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc)
                W[nt][pc] = *reinterpret_cast<const f16x8*>(wb + (nt * 2 + pc) * 512 + lane8);
    };

    // activations: byte address = ab + ((group<<4) ^ (key<<4)); lo plane at +LO (immediate)
    int ab[MTW], swb[MTW];
    auto tap_setup = [&](int t) {
        t = t < TAPS ? t : TAPS - 1;  // the zero-weight padding steps of layer 0 re-read the last tap
        const int ky = t / KW, kx = t - ky * KW;
        const int off = FIRST ? (ky - 2) * 11 + (kx - 2) : (ky - 1) * 10 + (kx - 1);
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            const int pa = pbase[mt] + off;
            ab[mt] = pa * ROWB;
            // swizzle key of the position being read = this row's key + 9dy + dx (= the writer's key,
            // act_key): the 16 lanes of a ds_read_b128 group then hit 16 distinct slots.
            // Out-of-board neighbours land in all-zero halo rows, where any key reads zeros.
            swb[mt] = (FIRST ? ((pa >> 1) & 3) : ((rrow[mt] + 9 * (ky - 1) + (kx - 1)) & 15)) << 4;
        }
    };
    auto read_x = [&](f16x8 (&X)[MTW][2], int s) {
        const int gb = (2 * s + h) << 4;  // this lane half's 8-channel group
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            const char* p = actb + (ab[mt] + (gb ^ swb[mt]));
            X[mt][0] = *reinterpret_cast<const f16x8*>(p);
            if (NPROD == 3) X[mt][1] = *reinterpret_cast<const f16x8*>(p + LO);
        }
    };

    f16x8 X0[MTW][2], X1[MTW][2];
    f16x8 W0[NT][2], W1[NT][2], W2[NT][2], W3[NT][2];

    // MLO..MHI: the position tiles of this wave that take part (tap skipping, see row_decode)
    auto mma = [&](auto MLO, auto MHI, const f16x8 (&X)[MTW][2], const f16x8 (&W)[NT][2]) {
        // product-major order: consecutive MFMAs write different accumulator tiles (no back-to-back dependence)
#pragma unroll
        for (int pr = 0; pr < NPROD; ++pr)
#pragma unroll
            for (int mt = decltype(MLO)::value; mt < decltype(MHI)::value; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[nt][pr == 1 ? 1 : 0], X[mt][pr == 2 ? 1 : 0],
                                                                         acc[mt][nt], 0, 0, 0);
    };

    // step ks: X(ks+1) is read from LDS and W(ks+3) fetched from L2 while step ks's MFMAs run
    auto step = [&](auto MLO, auto MHI, int ks, const f16x8 (&Xc)[MTW][2], f16x8 (&Xn)[MTW][2],
                    const f16x8 (&Wc)[NT][2], f16x8 (&Wn3)[NT][2], bool may_cross) {
        constexpr int NMFMA = NPROD * (decltype(MHI)::value - decltype(MLO)::value) * NT;
#if BK_EXP & 2   // timing experiment: no weight traffic in the loop (results are wrong)
        (void)Wn3;
#else
        load_w(Wn3, ks + 3);
#endif
        const int kn = ks + 1;
        if (may_cross && (kn % S) == 0) tap_setup(kn / S);
#if BK_EXP & 1   // timing experiment: no LDS reads in the loop (results are wrong)
        (void)Xn;
#else
        read_x(Xn, kn % S);
#endif
        mma(MLO, MHI, Xc, Wc);
        sched_pattern<NMFMA, (NPROD == 3 ? 2 : 1) * MTW, 2 * NT>(std::make_integer_sequence<int, (NPROD == 3 ? 2 : 1) * MTW>{});
        __builtin_amdgcn_sched_barrier(0);
    };

    // steps [k0, k1) (multiples of 4) with one tile range; the operand pipelines run across calls
    auto run = [&](auto MLO, auto MHI, int k0, int k1) {
#pragma unroll 1
        for (int ks = k0; ks < k1; ks += 4) {
            step(MLO, MHI, ks + 0, X0, X1, W0, W3, S < 2);
            step(MLO, MHI, ks + 1, X1, X0, W1, W0, S <= 2);
            step(MLO, MHI, ks + 2, X0, X1, W2, W1, S < 2);
            step(MLO, MHI, ks + 3, X1, X0, W3, W2, true);
        }
    };

    load_w(W0, 0);
    load_w(W1, 1);
    load_w(W2, 2);
    tap_setup(0);
    read_x(X0, 0);
    using I0 = std::integral_constant<int, 0>;
    using IM = std::integral_constant<int, MTW>;
    if constexpr (NB == 3 && !FIRST) {
        // taps ky=0 are steps [0,24), ky=1 [24,48), ky=2 [48,72)
        if (wm == 0) {
            run(std::integral_constant<int, 1>{}, IM{}, 0, 24);   // tile 0 = y=0 points: dy=-1 is all halo
            run(I0{}, IM{}, 24, 72);
        } else {
            run(I0{}, IM{}, 0, 48);
            run(I0{}, std::integral_constant<int, MTW - 1>{}, 48, 72);  // tile 7 = y=8 points: dy=+1 is all halo
        }
    } else if constexpr (NB == 3 && FIRST) {
        // 5x5: taps ky=0,1 are steps [0,20), ky=3,4 steps [30,50); ranges must be multiples of 4
        if (wm == 0) {
            run(std::integral_constant<int, 1>{}, IM{}, 0, 20);   // dy=-2,-1 read only halo for the y=0 rows
            run(I0{}, IM{}, 20, NSTEPS);
        } else {
            run(I0{}, IM{}, 0, 32);
            run(I0{}, std::integral_constant<int, MTW - 1>{}, 32, NSTEPS);  // dy=+1,+2: only halo for the y=8 rows
        }
    } else {
        run(I0{}, IM{}, 0, NSTEPS);
    }
}

template <int NB>
__global__ void __launch_bounds__(256) bk_leaf_eval_f16_kernel(const bk_eval_args a) {
    using G = Geo<NB>;
    constexpr int MTW = G::MTW, NT = G::NT;
    extern __shared__ __attribute__((aligned(16))) char smem16[];
    char* actb = smem16;
    const int dummy_addr = G::NPOS * 512 + (threadIdx.x & 63) * 16;
    float* hs = reinterpret_cast<float*>(smem16 + G::NPOS * 512 + G::DUMMY_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l32 = lane & 31;

    int net, task;
    const int bid = blockIdx.x;
    const int m4 = min(a.tasks_p, a.tasks_v) & ~3;
    if (bid < 2 * m4) {
        net = (bid >> 2) & 1;
        task = ((bid >> 3) << 2) | (bid & 3);
    } else {
        const int r = bid - 2 * m4;
        if (r < a.tasks_p - m4) { net = 0; task = m4 + r; }
        else { net = 1; task = m4 + r - (a.tasks_p - m4); }
    }
    const bk_net_params& P = a.net[net];
    const int b0 = (net ? a.off_v : a.off_p) + task * NB;  // this launch covers boards [off, B) of each net
    const int nb = min(NB, (net ? a.B_value : a.B_policy) - b0);

    STAMP(0);
    // ---- stage the feature planes as fp16 hi/lo: NCHW global -> [pos][hi 4x8 | lo 4x8] ----
    for (int i = tid; i < G::NP0 * 8; i += 256) reinterpret_cast<f32x4*>(actb)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    bool inexact = false;
    if (tid < nb * 81) {
        // one thread per position: 27 channel loads (coalesced across threads: consecutive points of a
        // plane), split into fp16 hi/lo, stored as 4 groups x (16 B hi + 16 B lo)
        const int b = tid / 81, q = tid - 81 * b, y = q / 9, x = q - 9 * y;
        const int p = pos5(b, y, x);
        float v[32];
        if (a.feats_dtype == BK_FEATS_F32_) {
            const float* src = static_cast<const float*>(a.feats) + (size_t)(b0 + b) * 2187 + q;
#pragma unroll
            for (int c = 0; c < 27; ++c) v[c] = __builtin_nontemporal_load(src + c * 81);  // streamed once: keep L2 for the weights
        } else {
            const uint8_t* src = static_cast<const uint8_t*>(a.feats) + (size_t)(b0 + b) * 2187 + q;
#pragma unroll
            for (int c = 0; c < 27; ++c) v[c] = (float)__builtin_nontemporal_load(src + c * 81);
        }
#pragma unroll
        for (int c = 27; c < 32; ++c) v[c] = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f16x8 hi, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                hi[j] = (_Float16)v[8 * g + j];
                lo[j] = (_Float16)(v[8 * g + j] - (float)hi[j]);
                inexact |= lo[j] != (_Float16)0.f;
            }
            char* d = actb + p * 128 + ((g ^ ((p >> 1) & 3)) << 4);
            *reinterpret_cast<f16x8*>(d) = hi;
            *reinterpret_cast<f16x8*>(d + 64) = lo;
        }
    }
    // barrier + workgroup-wide OR: does any input need its lo half?
    const bool need_lo = __syncthreads_or(inexact ? 1 : 0) != 0;

    STAMP(1);
    f32x16 acc[MTW][NT];
    const int wm = wave / G::WN, wn = wave - wm * G::WN;

    // this lane's GEMM rows (one per position tile of the wave): decoded once, the edge tiles read tables
    int rpos3[MTW], rpos5[MTW], rkey[MTW], rstore[MTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        int b, y, x, k;
        const bool valid = row_decode<NB>((wm * MTW + mt) * 32 + l32, b, y, x, k);
        rpos3[mt] = pos3(b, y, x);
        rpos5[mt] = pos5(b, y, x);
        rkey[mt] = k & 15;
        rstore[mt] = valid ? rpos3[mt] * 512 : -1;
    }

    // epilogue: acc register 4q+j of (mt, nt) = cout 32*(wn*NT+nt) + 8q + 4h + j at this lane's position
    float umax = 0.f;  // largest pre-clamp activation seen by this lane (inf if anything overflowed)
    // the layer's folded biases for this lane's couts (sa_out * bias) and its output scale: fetched before the
    // barrier that precedes the epilogue, so the loads fly while the wave waits for the others
    f32x4 bvec[NT * 4];
    float cs = 0.f;
    auto load_bias = [&](int L) {
        cs = P.cscale16[L];                             // sa_out / (sa_in * sw_L): a power of two
        const float* bias = P.bias16 + L * 128;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                bvec[nt * 4 + q] = *reinterpret_cast<const f32x4*>(bias + 32 * (wn * NT + nt) + 8 * q + 4 * h);
    };
    auto store = [&]() {
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            const int rowb = rstore[mt];
            const int key = rkey[mt] << 4;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c0 = 32 * (wn * NT + nt) + 8 * q + 4 * h;
                    const f32x4 bv = bvec[nt * 4 + q];
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float u = fmaf(acc[mt][nt][4 * q + j], cs, bv[j]);
                        v[j] = __builtin_amdgcn_fmed3f(u, 0.f, 65504.f);  // ReLU + fp16 range
                        umax = fmaxf(umax, u);
                    }
                    f16x4 hi, lo;
#pragma unroll
                    for (int j = 0; j < 4; ++j) hi[j] = (_Float16)v[j];
                    // lo = fp16(v - hi): v - hi is exact in fp32; v_fma_mix_f32 reads hi straight from its packed
                    // fp16 register (one VALU op instead of convert + subtract)
                    const f16x2 h01 = {hi[0], hi[1]}, h23 = {hi[2], hi[3]};
                    lo[0] = (_Float16)sub_f16_lo(v[0], h01);
                    lo[1] = (_Float16)sub_f16_hi(v[1], h01);
                    lo[2] = (_Float16)sub_f16_lo(v[2], h23);
                    lo[3] = (_Float16)sub_f16_hi(v[3], h23);
                    const int addr = rowb >= 0 ? rowb + ((((c0 >> 3) << 4) ^ key) | (8 * h)) : dummy_addr;
                    *reinterpret_cast<f16x4*>(actb + addr) = hi;
                    *reinterpret_cast<f16x4*>(actb + addr + 256) = lo;
                }
        }
    };

    // ---- layer 0: 5x5, 27(32) -> 128 ----
    if (need_lo) conv_layer16<NB, true, 3>(actb, P.wfrag16, acc, lane, wm, wn, rpos3, rpos5, rkey);
    else conv_layer16<NB, true, 2>(actb, P.wfrag16, acc, lane, wm, wn, rpos3, rpos5, rkey);
    STAMP(2);
    load_bias(0);
    __syncthreads();
    for (int i = tid; i < G::NPOS * 32; i += 256) reinterpret_cast<f32x4*>(actb)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    STAMP(3);
    store();
    STAMP(4);
    __syncthreads();
    STAMP(5);

    // ---- layers 1..6: 3x3, 128 -> 128, in place ----
#pragma unroll 1
    for (int L = 1; L < 7; ++L) {
        conv_layer16<NB, false>(actb, P.wfrag16 + BK16_L0_HALFS + (size_t)(L - 1) * BK16_L3_HALFS, acc, lane, wm, wn, rpos3, rpos5, rkey);
        STAMP(2 + 4 * L);
        load_bias(L);
        __syncthreads();
        STAMP(3 + 4 * L);
        store();
        STAMP(4 + 4 * L);
        __syncthreads();
        STAMP(5 + 4 * L);
    }

    // an activation beyond the fp16 range was clamped: tell the host, which re-runs the batch in fp32
    if (__any(!(umax < 65504.f)) && lane == 0 && a.overflow) atomicMax(a.overflow, a.overflow_tag);

    // ---- heads.  1x1 conv (untied bias): one thread per board point over all 4 waves, results to LDS;
    //      then wave w finishes board w (activations = (hi + lo) * inv_sa) ----
    if (tid < 81 * nb) {
        const int bb = tid / 81, q = tid - 81 * bb;
        const int y = q / 9, x = q - 9 * y, p = pos3(bb, y, x), key = act_key<NB>(bb, y, x) & 15;
        const f32x4* hw = reinterpret_cast<const f32x4*>(P.head_w);
        float d = 0.f;
#pragma unroll 4
        for (int g = 0; g < 16; ++g) {
            const char* src = actb + p * 512 + ((g ^ key) << 4);
            const f16x8 vh = *reinterpret_cast<const f16x8*>(src);
            const f16x8 vl = *reinterpret_cast<const f16x8*>(src + 256);
            const f32x4 w0 = hw[2 * g], w1 = hw[2 * g + 1];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                d += ((float)vh[j] + (float)vl[j]) * w0[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                d += ((float)vh[4 + j] + (float)vl[4 + j]) * w1[j];
            }
        }
        hs[bb * 96 + q] = d * P.inv_sa16 + P.head_b[q];
    }
    __syncthreads();
    if (wave < nb) {
        float s[2];
        s[0] = hs[wave * 96 + lane];
        s[1] = lane + 64 < 81 ? hs[wave * 96 + lane + 64] : 0.f;
        const int bg = b0 + wave;
        if (net == 0) {
            float m = fmaxf(s[0], (lane + 64 < 81) ? s[1] : -INFINITY);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
            const float e0 = expf(s[0] - m);
            const float e1 = (lane + 64 < 81) ? expf(s[1] - m) : 0.f;
            float sum = e0 + e1;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
            const float inv = 1.f / sum;
            if (a.logits) {
                a.logits[(size_t)bg * 81 + lane] = s[0];
                if (lane + 64 < 81) a.logits[(size_t)bg * 81 + lane + 64] = s[1];
            }
            if (a.probs) {
                a.probs[(size_t)bg * 81 + lane] = e0 * inv;
                if (lane + 64 < 81) a.probs[(size_t)bg * 81 + lane + 64] = e1 * inv;
            }
        } else {
            float* hv = hs + wave * 96;
            hv[lane] = fmaxf(s[0], 0.f);
            if (lane + 64 < 81) hv[lane + 64] = fmaxf(s[1], 0.f);
            __builtin_amdgcn_wave_barrier();
            float z = P.lin1_b[lane];
#pragma unroll 9
            for (int q = 0; q < 81; ++q) z += P.lin1_wt[q * 64 + lane] * hv[q];
            z = fmaxf(z, 0.f);
            float v = z * P.lin2_w[lane];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            v += P.lin2_b;
            if (lane == 0 && a.values) a.values[bg] = tanhf(v);
        }
    }
    STAMP(30);
}

template <int NB>
hipError_t launch16_nb(const bk_eval_args& a, hipStream_t stream) {
    static bool attr_set_dev[64] = {false};  // the attribute is per device: one flag per device ordinal
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    bool& attr_set = attr_set_dev[dev];
    auto kern = bk_leaf_eval_f16_kernel<NB>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, Geo<NB>::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    bk_eval_args args = a;
    args.tasks_p = (a.B_policy - a.off_p + NB - 1) / NB;
    args.tasks_v = (a.B_value - a.off_v + NB - 1) / NB;
    const int grid = args.tasks_p + args.tasks_v;
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), Geo<NB>::LDS_BYTES, stream, args);
    return hipGetLastError();
}

}  // namespace

hipError_t bk_launch_leaf_eval_f16(const bk_eval_args& a, int nb, hipStream_t stream) {
    switch (nb) {
        case 1: return launch16_nb<1>(a, stream);
        case 2: return launch16_nb<2>(a, stream);
        default: return launch16_nb<3>(a, stream);
    }
}
