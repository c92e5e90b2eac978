// bk_kernels.hip -- the fused leaf-evaluation kernel for gfx950 (MI355X / CDNA4).
//
// Replaces, for a batch of positions, the reference's two forward passes
//   PolicyNet.forward  (bokego/nnet.py:31-57)   + SOFT (nnet.py:16,273)
//   ValueNet.forward   (bokego/nnet.py:73-113)
// One workgroup (8 waves, two per SIMD) owns NB whole boards of ONE net and runs the whole network on them without
// leaving the CU:
//   * the NB x 81 x 128 fp32 activations live in LDS, position-major, in a "shared halo" layout (10-column rows: the
//     zero column between two board rows is the right halo of one and the left halo of the next), so every 3x3 tap is
//     a constant address offset and needs no bounds test; records are padded (528 B per position) so that the 16 lanes
//     of a ds_read_b128 group hit 16 distinct LDS slots without any address swizzling;
//   * each conv layer is an implicit GEMM  [16-position tiles] x [128 couts] x [taps*cin]  on the exact-fp32 matrix
//     instruction v_mfma_f32_16x16x4_f32 (bit-identical to an fmaf chain), computed transposed (weights = A operand);
//     the board points are grouped into tiles by edge class so that taps which point off the board for a whole tile are
//     skipped; every weight fragment comes straight from L2 into registers in a host-prepacked order (coalesced 1 KiB
//     buffer loads, scalar offsets) and the conv loops contain no vector-ALU instruction besides the MFMAs;
//   * the layer output stays in the accumulators until every wave has finished reading the layer input, then is written
//     back IN PLACE (bias + ReLU fused, 16-byte stores): no ping-pong buffer, which is what lets 3 boards (147 KB) fit
//     the 160 KB LDS;
//   * the untied-bias 1x1 head, the 81-way softmax and the value MLP + tanh are wave-level reductions at the end of the
//     same kernel.
// HBM traffic is therefore the compulsory 8,748 B in + 652 B out per position (+ weights, L2 resident).  BatchNorm (eval
// mode) is folded into the weights on the host in fp64.
//   * a small request of position records (192 B each) has no planes at all: the kernel computes them from the records while it
//     stages (stage_positions: the feature encoder's device code, bk_encode_dev.h) -- one launch per request instead of two.
// Batches of at most 128 (net, board) tasks take the cooperative form further down instead: 2 .. 12 workgroups on as
// many CUs share one board, each computing a slice of every layer and exchanging slices through L2 (same bits); requests
// between the whole-board forms' ranges (129..192 and 257..384 tasks) run as groups of THREE boards shared by 4 resp. 2 CUs.
//   * every conv dot product is summed in TWO fp32 chains (the halves of the kernel window), the heads in four: as close to
//     the float64 evaluation of the network as the reference's own fp32 is (conv_layer below; tools/emu/kernel_emu.c
//     reproduces the order on the CPU bit for bit).
// History: round 1 used 32-row tiles (v_mfma_f32_32x32x2_f32), an XOR-swizzled LDS layout and one wave per
// SIMD; rounds 2-3 were checked bit-identical to it on the GPU (tools/ab_bits.py), which is why the channel slots of
// layers 0..5 are kept in the permuted order bk_slot_perm (it reproduced that kernel's one-chain summation order); round 4
// changed the order on purpose (two chains) and is checked against the reference and the emulation instead.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "bk_encode_dev.h"
#include "bk_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// pointers that are LDS pointers by TYPE (32 bits, ds_* instructions): pointers kept in arrays across layers, or in a struct, are
// beyond what the compiler's address-space inference follows -- it falls back to generic (flat_*) accesses, which wait on both
// counters and halved the kernel's rate when that happened during development
typedef __attribute__((address_space(3))) const char lds_cchar;
typedef __attribute__((address_space(3))) const f32x4 lds_cf32x4;


#ifndef BK_EXP
#define BK_EXP 0
#endif
#ifdef BK_STAMPS
// Diagnostic build only: lane 0 of every wave records the shader clock at phase boundaries into a
// buffer nothing else reads (cdna_hip_programming.md "In-kernel stamps").  Never in the shipped .so.
#define STAMP(k)                                                                                   \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        unsigned long long _t;                                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");               \
        if (a.stamps && lane == 0 && wave < 4) a.stamps[((size_t)blockIdx.x * 4 + wave) * 32 + (k)] = _t; \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } while (0)
// the clock into a variable (packed differences go into a free slot: stage_input)
#define STAMP_T(var)                                                                               \
    unsigned long long var;                                                                        \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");               \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } while (0)
#else
#define STAMP(k) do {} while (0)
#define STAMP_T(var) do {} while (0)
#endif

namespace {

// ---- LDS layouts -------------------------------------------------------------------------
// Every address the conv loops use is (per-lane base) + (compile-time constant), so the loops contain NO vector-ALU
// address arithmetic (the fp32 MFMA shares the vector ALU: every VALU instruction between two MFMAs is lost MFMA
// time).  Bank conflicts are avoided by PADDING, not by swizzling.
//
// 128-channel activations: one record per position = 512 B of data + 16 B pad (REC3 = 528); a board row is 10
// records (column 0 = the zero halo shared by the end of one row and the start of the next) at a pitch of
// RP3 = 10*528 - 16: the never-touched pad of a row's last record overlaps the first 16 B of the next row's
// halo record.  The 16-byte slot of chunk c of (row R, column C) is (9R + C + c) mod 16, so the 16 lanes of a
// ds_read_b128 lane group -- 16 consecutive board points in (y,x) order -- read 16 distinct slots.
// NB == 3: boards are stacked without separator rows (R = 1 + 9b + y): the only points that could read across a
// board edge are the y=0 / y=8 points, which sit alone in their tiles and skip those taps (Tiles below).
// NB < 3: a zero row after every board (R = 1 + 10b + y).
constexpr int REC3 = 528, RP3 = 10 * REC3 - 16;
// layer-0 input (27 channels in 32 slots = 128 B + 16 B pad, 5x5 taps): 11 records per row (columns 0,1 = halo),
// pitch 1808 B (slot = (R + 9C + c) mod 16: consecutive points stay on distinct slots across row ends), two zero
// rows on top; NB == 3: one zero row after every board (y=0 / y=8 rows skip their out-of-board taps), else two.
constexpr int REC0 = 144, RP0 = 1808;

template <int NB>
struct Geo {
    static constexpr int RB3 = NB == 3 ? 9 : 10, NROWS3 = 1 + NB * RB3;
    static constexpr int RB0 = NB == 3 ? 10 : 11, NROWS0 = 2 + NB * RB0;
    // + one record: the right-halo reads of the last row (column 10 resp. 11,12) land just behind it
    static constexpr int L3_BYTES = NROWS3 * RP3 + REC3;
    static constexpr int L0_BYTES = NROWS0 * RP0 + REC0;
    static_assert(L0_BYTES <= L3_BYTES, "the layer-0 input lives inside the activation region");
        // 8 waves per workgroup = two per SIMD: while one wave issues its loads and waits, the other's MFMA keeps the matrix
    // pipe busy (measured on the earlier 32-row-tile loop: 65.06 cycles per 64-cycle MFMA with one wave per SIMD, +1.1 % with two)
    static constexpr int NW = 8;
    static constexpr int THREADS = 64 * NW;
    static constexpr int DUMMY_FLOATS = 256;        // sink for the padding rows' stores (one shared record: never read)
    static constexpr int HS_FLOATS = NB * 96;
    static constexpr int LDS_BYTES = L3_BYTES + (DUMMY_FLOATS + HS_FLOATS) * 4;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static constexpr int addr3(int b, int y, int x) { return (1 + RB3 * b + y) * RP3 + (x + 1) * REC3; }  // bytes
    static constexpr int addr0(int b, int y, int x) { return (2 + RB0 * b + y) * RP0 + (x + 2) * REC0; }
};
// ---- tiles ---------------------------------------------------------------------------------------------------------
// 16-position tiles (v_mfma_f32_16x16x4_f32: 16 positions x 16 couts x 4 input channels, 32 cycles).  The board points
// are grouped into tiles by edge class, and a tap skips an edge tile when it points off the board for every point of the
// tile (wave-uniform branch):
//   3 boards (243 points, 16 tiles): two position groups (wm = 0 / 1) x four cout groups; a wave = 8 tiles x 2 cout tiles
//       [ x-edge | 5 interior | y-edge a | y-edge b ]    x-edge: 16 of the 21 points with x = 0 (wm 0) / x = 8 (wm 1), y in 1..7
//                                                         y-edge: the 27 points with y = 0 / y = 8 (+ 5 padding rows)
//                                                         interior: the rest (+ 3 padding rows at the very end)
//       3x3 layers: 63 of 72 tile-taps, layer 0: 170 of 200
//   2 boards (162 points, 11 tiles): one position group x eight cout groups; a wave = 11 tiles x 1 cout tile
//       [ x=0 | x=8 | 7 interior | y=0 | y=8 ]   (14 / 14 / 98 + 4 left over / 16 / 16 points); 3x3 layers: 87 of 99 tile-taps
//   1 board (81 points, 6 tiles): no edge class fills a tile, every tile runs every tap
// Channels: an MFMA k-step consumes 4 input channels, one per lane quad kq, and a lane's accumulator holds 4
// consecutive output slots.  Outputs of layers 0..5 are kept in the slot order bk_slot_perm (bk_internal.h), which makes
// the k order of every dot product equal to the round-1 kernel's (8-channel groups, pairs (j, j+4)): bit-identical
// results.  Layer 6 writes natural channel order for the heads.  The host packs weights and biases accordingly.
// Scheduling of a channel group, measured on one box (leaf-evals/s at B = 4,096): hard sched_barrier fences between the
// phases of a channel group 506.9 k; none (the compiler's own order) 510.4 k; none + the pattern hint below 511.9 k.
#ifdef BK_PHASE_FENCES
#define PHASE_FENCE __builtin_amdgcn_sched_barrier(0)
#else
#define PHASE_FENCE do {} while (0)
#endif
// scheduling hint for one channel group: NMFMA matrix instructions with NREADS LDS reads and NLOADS weight loads dealt out
// between them.  Masks: 0x8 MFMA, 0x100 DS read, 0x20 VMEM read.
template <int NMFMA, int NREADS, int NLOADS, int... I>
__device__ __forceinline__ void sched_pattern(std::integer_sequence<int, I...>) {
    ((__builtin_amdgcn_sched_group_barrier(0x008, NMFMA / NREADS + (I < NMFMA % NREADS ? 1 : 0), 0),
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0), __builtin_amdgcn_sched_group_barrier(0x020, I < NLOADS ? 1 : 0, 0)),
     ...);
}
// the same with the reads up front: one behind each of the first NREADS MFMAs, then the rest of the MFMAs.  For the cooperative
// forms whose waves hold ONE tile (12, 8 and 6 CUs per board): dealt out "evenly", a channel group's only read came behind all four
// of its MFMAs and was waited for at once -- the whole LDS latency, once per group (round 6: requests of up to 32 tasks 83 -> 79 us,
// two tasks 60 -> 56, 33 tasks 90 -> 87; forms whose waves hold three or four tiles measured the same either way and keep the even deal)
template <int NMFMA, int NREADS, int NLOADS, int... I>
__device__ __forceinline__ void sched_front(std::integer_sequence<int, I...>) {
    ((__builtin_amdgcn_sched_group_barrier(0x008, 1, 0), __builtin_amdgcn_sched_group_barrier(0x100, 1, 0),
      __builtin_amdgcn_sched_group_barrier(0x020, I < NLOADS ? 1 : 0, 0)),
     ...);
    __builtin_amdgcn_sched_group_barrier(0x008, NMFMA - NREADS, 0);
}
template <int NB>
struct Tiles {
    static_assert(NB >= 1 && NB <= 3, "tiles are laid out for 1-, 2- and 3-board workgroups");
    static constexpr int WM = NB == 3 ? 2 : 1;          // position groups
    static constexpr int WN = 8 / WM;                   // cout groups
    static constexpr int RT = NB == 3 ? 8 : NB == 2 ? 11 : 6;   // 16-position tiles per wave
    static constexpr int CTW = 8 / WN;                  // 16-cout tiles per wave
    // tile classes of a wave: [A0, A1) always; X0 skips when dx < 0 resp. (NB == 3, wm == 1) dx > 0; X1 (NB == 2) when
    // dx > 0; [Y0a, Y0b) when dy < 0 resp. (NB == 3, wm == 1) dy > 0; Y1 (NB == 2) when dy > 0; -1: no such tile
    // 1 board (81 points, 6 tiles): [3 interior | x-edges + the 49th interior point | y=0 | y=8]: the two y-edge tiles (9 points
    // + 7 padding rows each) skip their outward taps -- 48 of 54 tile-taps (round 3: the points in board order, nothing to skip)
    static constexpr int A0 = NB == 3 ? 1 : NB == 2 ? 2 : 0, A1 = NB == 3 ? 6 : NB == 2 ? 9 : 4;
    static constexpr int X0 = NB == 1 ? -1 : 0, X1 = NB == 2 ? 1 : -1;
    static constexpr int Y0a = NB == 3 ? 6 : NB == 2 ? 9 : 4, Y0b = NB == 3 ? 8 : NB == 2 ? 10 : 5, Y1 = NB == 2 ? 10 : NB == 1 ? 5 : -1;
    static constexpr bool WM_EDGES = NB == 3;           // the two position groups hold opposite edges
    static constexpr bool FRONT = false;                // (conv_layer, sched_front)
    static constexpr bool DB2 = NB != 3;                // second chain: double-buffered fragments too (conv_layer) where registers allow
    static constexpr int RING = 4;                      // slots of the weight ring (groups fetched RING - 1 ahead)
};
struct TileRow { int b, y, x; bool valid; };
template <int NB>
constexpr TileRow tile_row(int wm, int rt, int p16) {
    TileRow r{0, 4, 0, true};
    if constexpr (NB == 1) {
        if (rt <= 2) {                       // interior: the first 48 of the 49 points (y, x = 1..7)
            const int i = rt * 16 + p16;
            r.y = 1 + i / 7; r.x = 1 + i % 7;
        } else if (rt == 3) {                // x = 0 and x = 8 (y = 1..7), the 49th interior point (7,7), one padding row
            if (p16 < 14) { r.y = 1 + p16 % 7; r.x = p16 < 7 ? 0 : 8; }
            else if (p16 == 14) { r.y = 7; r.x = 7; }
            else r.valid = false;
        } else {                             // y = 0 (tile 4) / y = 8 (tile 5): 9 points + 7 padding rows
            if (p16 < 9) { r.y = rt == 4 ? 0 : 8; r.x = p16; } else r.valid = false;
        }
    } else if constexpr (NB == 3) {
        if (rt == 0) {                       // x-edge tile: the first 16 of the 21 points (b, y = 1..7, x = 0 or 8)
            r.b = p16 / 7; r.y = 1 + p16 % 7; r.x = wm ? 8 : 0;
        } else if (rt <= 5) {                // interior pool: 147 interior points, then the 5 + 5 x-edge points left over
            const int i = wm * 80 + (rt - 1) * 16 + p16;
            if (i < 147) { r.b = i / 49; const int rem = i - 49 * r.b; r.y = 1 + rem / 7; r.x = 1 + rem % 7; }
            else if (i < 157) { const int jx = 16 + (i - 147) % 5; r.b = jx / 7; r.y = 1 + jx % 7; r.x = i < 152 ? 0 : 8; }
            else r.valid = false;
        } else {                             // y-edge tiles: (b, y = 0 or 8, x = 0..8)
            const int i = (rt - 6) * 16 + p16;
            if (i < 27) { r.b = i / 9; r.x = i % 9; r.y = wm ? 8 : 0; }
            else r.valid = false;
        }
    } else {
        if (rt <= 1) {                       // x = 0 / x = 8 tiles: the 14 points (b, y = 1..7) + 2 padding rows
            if (p16 < 14) { r.b = p16 / 7; r.y = 1 + p16 % 7; r.x = rt ? 8 : 0; } else r.valid = false;
        } else if (rt <= 8) {                // interior pool: 98 interior points, then the 2 + 2 y-edge points left over
            const int i = (rt - 2) * 16 + p16;
            if (i < 98) { r.b = i / 49; const int rem = i - 49 * r.b; r.y = 1 + rem / 7; r.x = 1 + rem % 7; }
            else if (i < 102) { r.b = 1; r.x = 7 + (i & 1); r.y = i < 100 ? 0 : 8; }   // entries 16, 17 of the y-edge lists
            else r.valid = false;
        } else {                             // y = 0 / y = 8 tiles: the first 16 of the 18 points (b, x = 0..8)
            r.b = p16 / 9; r.x = p16 % 9; r.y = rt == 9 ? 0 : 8;
        }
    }
    if (!r.valid) { r.b = 0; r.y = 4; r.x = 0; }   // padding rows compute from a valid address; stored to the dummy record
    return r;
}
// the same as a table in device memory, built at compile time: per (position group, tile, row of the tile) the LDS byte offsets
// of the position's tap-(0,0) neighbour in the layer-0 layout and in the 128-channel layout (bit 0 of the latter: a real board
// point, not a padding row).  The kernels read their eight-odd entries instead of decoding (divisions) in registers.
template <int NB>
struct RowTable {
    static constexpr int N = Tiles<NB>::WM * Tiles<NB>::RT * 16;
    int a0[N], a3v[N];
};
template <int NB>
constexpr RowTable<NB> make_row_table() {
    RowTable<NB> t{};
    for (int wm = 0; wm < Tiles<NB>::WM; ++wm)
        for (int rt = 0; rt < Tiles<NB>::RT; ++rt)
            for (int p = 0; p < 16; ++p) {
                const TileRow r = tile_row<NB>(wm, rt, p);
                const int i = (wm * Tiles<NB>::RT + rt) * 16 + p;
                t.a0[i] = Geo<NB>::addr0(r.b, r.y, r.x) - 2 * RP0 - 2 * REC0;
                t.a3v[i] = (Geo<NB>::addr3(r.b, r.y, r.x) - RP3 - REC3) | (r.valid ? 1 : 0);
            }
    return t;
}
template <int NB>
__device__ const RowTable<NB> g_rows = make_row_table<NB>();
// float slot of input plane c inside a layer-0 record: planes 0..23 by the inverse of bk_slot_perm, planes 24..26 are
// k-step 2 of the second channel group (slot 16 + 4kq + 2 for lane quad kq = c - 24)
__device__ __forceinline__ int in_slot(int c) {
    return c < 24 ? (c & ~15) | ((c & 1) << 3) | (c & 4) | (((c >> 3) & 1) << 1) | ((c >> 1) & 1) : 16 + 4 * (c - 24) + 2;
}

// Summation order (round 4).  Every dot product of a conv layer is TWO fp32 chains -- the taps of the first half of the
// kernel window (row-major taps 0..4 of 9, 0..11 of 25) and those of the second half -- added once at the end, then the
// bias: out = (chain(first taps) + chain(second taps)) + bias.  Within a chain the order is the MFMA's: tap by tap, group of
// 16 input slots by group, k-step j by k-step, lane quad kq ascending (input channel bk_slot_perm(16g + 4kq + j)).  One chain
// of 1,152 terms (rounds 1-3) put the fp32 kernel 5.8e-5 from the float64 evaluation of the network at |logit| ~ 65, the
// reference's own fp32 being at 3.5e-5; the rounding error of a chain grows with its length times the size of its partial
// sums, so two chains of half the length halve it (tools/error_budget.py reproduces the kernel's order on the CPU and prices
// the variants over the 49,152-position sweep: profiles/r04_error_budget.md).  Every form of the kernel -- 1, 2, 3 boards per
// workgroup, the cooperative slices -- sums in this same order: bit-identical outputs.
// The second chain needs a second set of accumulators (64 more registers in the 3-board form, which had 210 of 256 in use):
// they come out of the activation fragments' double buffer, during the second chain only (conv_layer, A0 / A1).
//
// wl: the layer's weights [tap][group of 16 input slots][cout tile (8)][lane][4] (pack_trunk); tile0: the wave's first cout tile.
// ap[rt]: LDS address of this lane's position of tile rt at tap (0,0), + 16 * (lane >> 4); advanced tap by tap and put back at
// the end (the kernel keeps ONE copy of these pointers alive across the layers).
// W: the weight ring of F::RING slots, owned by the kernel so that it lives across layers: group n of the network's stream uses
// slot n % RING and fetches group n + RING - 1 into the slot the group before it has just left; the last groups of a layer so
// fetch the first ones of the NEXT layer (the layers are contiguous in memory).  Layer 0 starts at slot 0, its 50 groups leave
// the 3x3 layers (72 groups each: a whole number of turns) at slot 2.  RING = 4: three groups ahead (two until round 4: with
// one wave per SIMD and 4..12 MFMAs per group -- the cooperative forms -- two groups did not cover the L2 latency of the
// weights: 8 CUs per board 100 -> 85 us).  Eight slots, seven ahead, were measured too (the code takes RING = 8): no further
// gain in any form (86.9 against 84.9 us at 9 tasks, 104.1 against 103.5 at 63, 141.6 against 134.7 at 65) -- three cover it.
// F: the wave's tile set (Tiles<NB>, or CoopTiles<SC, SR, RH> of the cooperative small-batch kernel below).
template <class F, bool FIRST>
__device__ __forceinline__ void conv_layer(const char* actb, const float* __restrict__ wl, f32x4 (&acc)[F::CTW][F::RT], int lane,
                                           int wm, int tile0, lds_cchar* (&ap)[F::RT], f32x4 (&W)[F::RING][F::CTW]) {
    constexpr int RT = F::RT, CTW = F::CTW, RING = F::RING;
    static_assert(RING == 4 || RING == 8, "layer 0's 50 groups leave the ring at slot 2: what the 3x3 layers start from");
    constexpr int PH = FIRST ? 0 : 50 % RING;           // ring slot of the layer's group 0
    constexpr int KW = FIRST ? 5 : 3, TAPS = KW * KW;
    constexpr int HALF = FIRST ? 12 : 5;                // the second chain starts at this tap (layer 0: even, its taps go in pairs)
    constexpr int G = FIRST ? 2 : 8;                    // groups of 16 input slots per tap
    constexpr int REC = FIRST ? REC0 : REC3, RP = FIRST ? RP0 : RP3;
    (void)actb;
    f32x4 part[CTW][RT];                                // the finished first chain
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct) acc[ct][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // this wave's cout tiles (tile0 ..): 1 KiB per tile and group, 8 KiB per group
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wl + tile0 * 256), 0, 0x7ffffff0, 0x00020000);
    const int lane16 = lane * 16;
    int boff = (RING - 1) * 8192;                       // scalar: byte offset of the group being fetched
    auto load_w = [&](f32x4 (&W)[CTW]) {
#if BK_EXP & 2   // timing experiment (make exp EXP=2|3): no weight traffic in the loops -- results are wrong
        (void)W;
#else
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct)
            W[ct] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, lane16 + ct * 1024, boff, 0));
#endif
        boff += 8192;
    };
    // The activation fragments (one float4 per tile and channel group).  FIRST CHAIN: double-buffered per group (A0 / A1 in
    // turn; the group's reads are dealt out evenly between its MFMAs) -- `part` is not alive yet, the registers are there.
    // SECOND CHAIN (3-board workgroups; the other forms have the registers and keep the double buffer: F::DB2): `part` holds
    // the first chain's sums, so the fragments live in A0 alone: a tile's fragment is dead after
    // its MFMAs of the group's last k-step, and the next group's is read into the same registers right there, one tile
    // behind (a read into registers the MFMA just issued still takes its operands from would need wait states).  The
    // single-buffered body is ~2 % slower than the double-buffered one (its reads crowd into the last k-step;
    // profiles/r04_two_chains.md), which is why the first chain, 5 of 9 taps, keeps the other.
    f32x4 A0[RT], A1[RT];
    auto read_a = [&](f32x4& dst, int rt, int imm) {
#if BK_EXP & 1   // timing experiment (make exp EXP=1|3): no LDS reads in the loops -- results are wrong
        (void)dst; (void)rt; (void)imm;
#else
        dst = *reinterpret_cast<lds_cf32x4*>(ap[rt] + imm);
#endif
    };
#if BK_EXP & 1
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) { A0[rt] = f32x4{1.f, 2.f, 3.f, 4.f} * (float)lane; A1[rt] = f32x4{.5f, .25f, .125f, 1.f} * (float)lane; }
#endif

    // One group of 16 input slots (Wc: this group's weights, Wn: receives the group two ahead).  ONE code body per chain for
    // all its taps: the interior tiles always run, every edge tile sits behind a wave-uniform branch (specialised copies of
    // the tap body -- tens of KB of straight-line MFMAs -- were measured slower).  sx0/sx1/sy0/sy1: skip that edge class.
    // DB: the double-buffered body (first chain).
    auto do_group = [&](auto DBc, auto GIc, f32x4 (&Wc)[CTW], f32x4 (&Wn)[CTW], int delta, bool sx0, bool sx1, bool sy0, bool sy1) {
        constexpr bool DB = decltype(DBc)::value;
        constexpr int g = decltype(GIc)::value;
        constexpr int JN = (FIRST && g == 1) ? 3 : 4;   // layer 0: planes 16..26 take 3 k-steps
        f32x4 (&Ac)[RT] = (DB && (g & 1)) ? A1 : A0;
        f32x4 (&An)[RT] = (DB && !(g & 1)) ? A1 : A0;
        // k-steps [j0, j1) of tiles [T0, T1): k-step by k-step over the tiles (an accumulator comes round again after
        // (T1 - T0) x CTW MFMAs: with two waves on the SIMD a short distance costs -- every 2 MFMAs 496.8 k leaf-evals/s,
        // every 4 500.9 k, every 10 501.7 k)
        auto mfmas = [&](int j0, int j1, auto T0, auto T1, auto&& after) {
#pragma unroll
            for (int j = j0; j < j1; ++j)
#pragma unroll
                for (int rt = decltype(T0)::value; rt < decltype(T1)::value; ++rt) {
#pragma unroll
                    for (int ct = 0; ct < CTW; ++ct)
                        acc[ct][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wc[ct][j], Ac[rt][j], acc[ct][rt], 0, 0, 0);
                    after(j, rt);
                }
        };
        // the fragment of tile rt for the next group (the next tap's group 0 behind the last group)
        auto next_a = [&](int rt) {
            if (g == G - 1) {
                ap[rt] += delta;
                read_a(An[rt], rt, 0);
            } else {
                read_a(An[rt], rt, (g + 1) * 64);
            }
        };
        using TA0 = std::integral_constant<int, F::A0>;
        using TA1 = std::integral_constant<int, F::A1>;
        load_w(Wn);
        if constexpr (DB) {
            auto nothing = [](int, int) {};
            mfmas(0, 1, TA0{}, TA1{}, nothing);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) next_a(rt);
            mfmas(1, JN, TA0{}, TA1{}, nothing);
#ifndef BK_PHASE_FENCES
            // spread the RT activation reads and the CTW weight loads evenly between the interior tiles' MFMAs (F::FRONT: up front)
            if constexpr (F::FRONT && (F::A1 - F::A0) * CTW * JN >= RT) sched_front<(F::A1 - F::A0) * CTW * JN, RT, CTW>(std::make_integer_sequence<int, RT>{});
            else sched_pattern<(F::A1 - F::A0) * CTW * JN, RT, CTW>(std::make_integer_sequence<int, RT>{});
#endif
            PHASE_FENCE;
            if constexpr (F::X0 >= 0) { if (!sx0) mfmas(0, JN, std::integral_constant<int, F::X0>{}, std::integral_constant<int, F::X0 + 1>{}, nothing); }
            if constexpr (F::X1 >= 0) { if (!sx1) mfmas(0, JN, std::integral_constant<int, F::X1>{}, std::integral_constant<int, F::X1 + 1>{}, nothing); }
            if constexpr (F::Y0a >= 0) { if (!sy0) mfmas(0, JN, std::integral_constant<int, F::Y0a>{}, std::integral_constant<int, F::Y0b>{}, nothing); }
            if constexpr (F::Y1 >= 0) { if (!sy1) mfmas(0, JN, std::integral_constant<int, F::Y1>{}, std::integral_constant<int, F::Y1 + 1>{}, nothing); }
        } else {
            int prev = -1;
            // a range of tiles: its MFMAs (if it runs), the reads in the last k-step, each one tile behind
            auto range = [&](bool run, auto T0, auto T1) {
                auto behind = [&](int j, int rt) {
                    if (j == JN - 1) {
                        if (prev >= 0) next_a(prev);
                        prev = rt;
                    }
                };
                if (run) {
                    mfmas(0, JN, T0, T1, behind);
                } else {
#pragma unroll
                    for (int rt = decltype(T0)::value; rt < decltype(T1)::value; ++rt) behind(JN - 1, rt);
                }
            };
            range(true, TA0{}, TA1{});
#ifndef BK_PHASE_FENCES
            sched_pattern<(F::A1 - F::A0) * CTW * JN, F::A1 - F::A0 - 1, CTW>(std::make_integer_sequence<int, F::A1 - F::A0 - 1>{});
#endif
            PHASE_FENCE;
            if constexpr (F::X0 >= 0) range(!sx0, std::integral_constant<int, F::X0>{}, std::integral_constant<int, F::X0 + 1>{});
            if constexpr (F::X1 >= 0) range(!sx1, std::integral_constant<int, F::X1>{}, std::integral_constant<int, F::X1 + 1>{});
            if constexpr (F::Y0a >= 0) range(!sy0, std::integral_constant<int, F::Y0a>{}, std::integral_constant<int, F::Y0b>{});
            if constexpr (F::Y1 >= 0) range(!sy1, std::integral_constant<int, F::Y1>{}, std::integral_constant<int, F::Y1 + 1>{});
            next_a(prev);                               // the last tile's own (the one read that follows its MFMAs directly)
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>;
    using I5 = std::integral_constant<int, 5>;
    using I6 = std::integral_constant<int, 6>;
    using I7 = std::integral_constant<int, 7>;
    // one tap; DB: first chain; PHc: ring slot of the tap's first group
    auto tap = [&](auto DBc, auto PHc, int t) {
        const int ky = t / KW, kx = t - ky * KW;
        const int d = t == TAPS - 1 ? 0 : (kx == KW - 1 ? RP - (KW - 1) * REC : REC);
        const bool lo_x = kx < KW / 2, hi_x = kx > KW / 2, lo_y = ky < KW / 2, hi_y = ky > KW / 2;
        // 3 boards: a wave's x-edge tile holds x = 0 (wm 0) or x = 8 (wm 1) points, its y-edge tiles y = 0 or y = 8
        const bool sx0 = F::WM_EDGES ? (wm == 0 ? lo_x : hi_x) : lo_x, sx1 = hi_x;
        const bool sy0 = F::WM_EDGES ? (wm == 0 ? lo_y : hi_y) : lo_y, sy1 = hi_y;
        // group n of the layer (ring slot (PH + n) % RING) fetches group n + RING - 1 into the slot group n - 1 has left
        constexpr int P0 = decltype(PHc)::value;        // slot of this tap's group 0
        auto grp = [&](auto GIc) {
            constexpr int g = decltype(GIc)::value;
            do_group(DBc, GIc, W[(P0 + g) % RING], W[(P0 + g + RING - 1) % RING], d, sx0, sx1, sy0, sy1);
        };
        grp(I0{});
        grp(I1{});
        if constexpr (!FIRST) {
            grp(I2{});
            grp(I3{});
            grp(I4{});
            grp(I5{});
            grp(I6{});
            grp(I7{});
        }
    };

#pragma unroll
    for (int rt = 0; rt < RT; ++rt) read_a(A0[rt], rt, 0);
    // taps [t0, t1) of one chain.  3x3: every tap starts at slot PH (8 groups per tap = whole turns of the ring).  Layer 0 (2
    // groups per tap): RING / 2 taps make a turn, so the loop is unrolled that far; t0 and t1 are multiples of it here
    auto taps = [&](auto DBc, int t0, int t1) {
        if constexpr (FIRST) {
            constexpr int U = RING / 2;
#pragma unroll 1
            for (int t = t0; t < t1; t += U) {
                tap(DBc, std::integral_constant<int, 0>{}, t);
                tap(DBc, std::integral_constant<int, 2>{}, t + 1);
                if constexpr (U == 4) {
                    tap(DBc, std::integral_constant<int, 4>{}, t + 2);
                    tap(DBc, std::integral_constant<int, 6>{}, t + 3);
                }
            }
        } else {
#pragma unroll 1
            for (int t = t0; t < t1; ++t) tap(DBc, std::integral_constant<int, PH>{}, t);
        }
    };
    static_assert(!FIRST || (HALF % (RING / 2) == 0 && (TAPS - 1) % (RING / 2) == 0), "layer 0: 12 + 12 taps in whole turns, then tap 24 at slot 0");
    // first chain: taps [0, HALF), an even number of groups: the second chain finds its first fragments in A0
    taps(std::true_type{}, 0, HALF);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct) {
            part[ct][rt] = acc[ct][rt];
            acc[ct][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    // second chain: taps [HALF, TAPS)
    using DB2 = std::integral_constant<bool, F::DB2>;
    taps(DB2{}, HALF, FIRST ? TAPS - 1 : TAPS);
    if constexpr (FIRST) tap(DB2{}, std::integral_constant<int, 0>{}, TAPS - 1);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        ap[rt] -= (KW - 1) * (RP + REC);                // back to tap (0,0): the taps advanced it by KW - 1 rows and KW - 1 columns
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct) acc[ct][rt] = part[ct][rt] + acc[ct][rt];
    }
}

// bias + ReLU + in-place store: accumulator (rt, ct) of a lane = output slots 16*(CTW*wn + ct) + 4*kq .. +3 of its position
template <class F>
__device__ __forceinline__ void store_layer(char* actb, const f32x4 (&acc)[F::CTW][F::RT],
                                                 const f32x4 (&bv)[F::CTW], lds_cchar* const (&ap)[F::RT], int store_c,
                                                 unsigned valid, int dummy_byte) {
#pragma unroll
    for (int rt = 0; rt < F::RT; ++rt) {
        // the lane's record of tile rt = its tap-(0,0) read pointer + a per-lane constant; padding rows go to the dummy record
        typedef __attribute__((address_space(3))) char lds_char;
        lds_char* wp = (valid >> rt) & 1 ? (lds_char*)(ap[rt]) + store_c : (lds_char*)actb + dummy_byte;
#pragma unroll
        for (int ct = 0; ct < F::CTW; ++ct) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[ct][rt][e] + bv[ct][e], 0.f);
            *reinterpret_cast<__attribute__((address_space(3))) f32x4*>(wp + ct * 64) = v;
        }
    }
}
template <class F>
__device__ __forceinline__ void load_bias(f32x4 (&bv)[F::CTW], const float* __restrict__ bias, int wn, int kq) {
#pragma unroll
    for (int ct = 0; ct < F::CTW; ++ct) bv[ct] = *reinterpret_cast<const f32x4*>(bias + 16 * (F::CTW * wn + ct) + 4 * kq);
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- stage NB boards' feature planes: NCHW global -> position-major LDS (layer-0 layout) ----
// every thread fetches its <= ceil(NB*2187/THREADS) elements first (all loads in flight: the loop used to pay one
// global-memory latency per element), then scatters them; non-temporal: the planes are read once and must not
// evict weight lines from L2.  The caller's __syncthreads() follows.
// ---- the same from POSITION RECORDS (a.feats_dtype == BK_FEATS_POS_: small requests of the ticket path): the feature encoder's body
// (bk_encode_dev.h: bitboards by ballots, chains by flood fill in registers, per-point mask algebra) runs on the workgroup's first 256
// threads -- one thread per board point of up to three boards -- and every thread puts its point's 27 plane values straight into the
// layer-0 layout.  The planes never exist in memory, and the request is one kernel instead of two: the encoder launch and the dependent
// dispatch behind it (~8 us of a ~130-us round trip of the one-tree search) are gone.  In the cooperative forms every slice encodes
// its board(s) for itself.  The encoder's 4 KB of LDS records sit in the activation region, behind the layer-0 input.
template <int NB, int THREADS>
__device__ __forceinline__ void stage_positions(const bk_eval_args& a, char* actb, int b0, int nb, int tid) {
    using G = Geo<NB>;
    static_assert(THREADS >= 256, "the encoder's mapping: one thread per board point of up to three boards");
    constexpr int SCRATCH = (G::L0_BYTES + 15) & ~15;
    static_assert(SCRATCH + (int)sizeof(bk_enc::EncLds) <= G::L3_BYTES, "the encoder's records live behind the layer-0 input");
    bk_enc::EncLds& S = *reinterpret_cast<bk_enc::EncLds*>(actb + SCRATCH);
    // zero the layer-0 region (halo!); the barriers inside encode_points order it before the stores below
    for (int i = tid; i < G::L0_BYTES / 16; i += THREADS) reinterpret_cast<f32x4*>(actb)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned char v[27];
    bool live;
    int p, q;
    bk_enc::encode_points(static_cast<const unsigned char*>(a.feats) + (size_t)b0 * BK_POS_BYTES, nb, tid, S, v, live, p, q);
    if (live) {
        const int y = q / 9, x = q - 9 * y;
        char* rec = actb + G::addr0(p, y, x);
#pragma unroll
        for (int c = 0; c < 27; ++c) *reinterpret_cast<float*>(rec + in_slot(c) * 4) = (float)v[c];
    }
}

template <int NB, int THREADS>
__device__ __forceinline__ void stage_input(const bk_eval_args& a, char* actb, int b0, int nb, int tid) {
    using G = Geo<NB>;
    if (a.feats_dtype == BK_FEATS_POS_) {               // (uniform over the launch)
        stage_positions<NB, THREADS>(a, actb, b0, nb, tid);
        return;
    }
    constexpr int PER = (NB * 2187 + THREADS - 1) / THREADS;
    const int n = nb * 2187;
    // every load is issued unconditionally (from a clamped, always valid index) and selected afterwards: a load behind a per-lane
    // condition waits for its data where the branches join, i.e. the loads went out ONE AT A TIME -- ~800 cycles each, 26 per thread
    // in a 256-thread three-board workgroup: 21 k cycles = 9 us of a 160-us request (in-kernel stamps, round 5)
    float v[PER];
    if (a.feats_dtype == BK_FEATS_F32_) {
        const float* X = static_cast<const float*>(a.feats) + (size_t)b0 * 2187;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int e = tid + THREADS * k;
            v[k] = __builtin_nontemporal_load(X + (e < n ? e : 0));
        }
    } else {
        const uint8_t* X = static_cast<const uint8_t*>(a.feats) + (size_t)b0 * 2187;
        uint8_t raw[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int e = tid + THREADS * k;
            raw[k] = __builtin_nontemporal_load(X + (e < n ? e : 0));
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) v[k] = (float)raw[k];
    }
    STAMP_T(t_issued);                                  // (diagnostic builds: loads issued / LDS zeroed / barrier passed / scattered)
    // zero the layer-0 region (halo!) while the loads fly
    for (int i = tid; i < G::L0_BYTES / 16; i += THREADS) reinterpret_cast<f32x4*>(actb)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    STAMP_T(t_zeroed);
    __syncthreads();
    STAMP_T(t_barrier);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int e = tid + THREADS * k;
        if (e < n) {
            const int b = e / 2187, ee = e - b * 2187, c = ee / 81, q = ee - c * 81, y = q / 9, x = q - 9 * y;
            *reinterpret_cast<float*>(actb + G::addr0(b, y, x) + in_slot(c) * 4) = v[k];
        }
    }
    STAMP_T(t_scattered);
#ifdef BK_STAMPS
    if (a.stamps && (tid & 63) == 0 && (tid >> 6) < 4)   // slot 31: the three phases' cycles, 16 bits each
        a.stamps[((size_t)blockIdx.x * 4 + (tid >> 6)) * 32 + 31] = ((t_zeroed - t_issued) & 0xffff) | (((t_barrier - t_zeroed) & 0xffff) << 16) |
                                                                   (((t_scattered - t_barrier) & 0xffff) << 32);
#endif
}

// ---- heads of one board (one wave): untied-bias 1x1 conv, then softmax (PolicyNet) or the value MLP + tanh ----
// rec_base: LDS byte address of the board's point (0,0) record minus addr3(0,0,0) terms, i.e. G::addr3(b, y, x) is
// formed by the caller's geometry; hv: 96 floats of LDS scratch; bg: the board's index in the batch.
template <class G>
__device__ __forceinline__ void run_heads(const bk_eval_args& a, const bk_net_params& P, const char* actb, float* hv,
                                              int net, int lane, int board, int bg) {
    float s[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int q = lane + 64 * k;
        float d = 0.f;
        if (q < 81) {
            // four chains -- element e of every float4 of the record -- combined pairwise, then the untied bias (one chain
            // of 128 terms until round 3: at |logit| ~ 65 its partial sums cost up to 1e-5 of the 1e-4 budget)
            const int y = q / 9, x = q - 9 * y;
            const f32x4* rec = reinterpret_cast<const f32x4*>(actb + G::addr3(board, y, x));
            const f32x4* hw = reinterpret_cast<const f32x4*>(P.head_w);
            f32x4 d4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
            for (int cc = 0; cc < 32; ++cc) {
                const f32x4 v = rec[cc];
                const f32x4 w = hw[cc];
#pragma unroll
                for (int e = 0; e < 4; ++e) d4[e] = __builtin_fmaf(v[e], w[e], d4[e]);
            }
            d = ((d4[0] + d4[1]) + (d4[2] + d4[3])) + P.head_b[q];
        }
        s[k] = d;
    }
    if (net == 0) {
        // PolicyNet: logits = head; probs = softmax
        const float m = wave_max(fmaxf(s[0], (lane + 64 < 81) ? s[1] : -INFINITY));
        const float e0 = expf(s[0] - m);
        const float e1 = (lane + 64 < 81) ? expf(s[1] - m) : 0.f;
        const float inv = 1.f / wave_sum(e0 + e1);
        if (a.logits) {
            a.logits[(size_t)bg * 81 + lane] = s[0];
            if (lane + 64 < 81) a.logits[(size_t)bg * 81 + lane + 64] = s[1];
        }
        if (a.probs) {
            a.probs[(size_t)bg * 81 + lane] = e0 * inv;
            if (lane + 64 < 81) a.probs[(size_t)bg * 81 + lane + 64] = e1 * inv;
        }
    } else {
        // ValueNet: (BN2d folded) ReLU -> lin1 (BN1d folded) -> ReLU -> lin2 -> tanh
        hv[lane] = fmaxf(s[0], 0.f);
        if (lane + 64 < 81) hv[lane + 64] = fmaxf(s[1], 0.f);
        __builtin_amdgcn_wave_barrier();
        // all 81 rows of lin1 in flight at once (one L2 latency instead of nine), then the sum in its order
        float w1[81];
#pragma unroll
        for (int q = 0; q < 81; ++q) w1[q] = P.lin1_wt[q * 64 + lane];
        // four chains (q mod 4; the bias opens chain 0), combined pairwise
        float z4[4] = {P.lin1_b[lane], 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 81; ++q) z4[q & 3] = __builtin_fmaf(w1[q], hv[q], z4[q & 3]);
        float z = (z4[0] + z4[1]) + (z4[2] + z4[3]);
        z = fmaxf(z, 0.f);
        const float v = wave_sum(z * P.lin2_w[lane]) + P.lin2_b;
        if (lane == 0 && a.values) a.values[bg] = tanhf(v);
    }
}

// GATED: the redo of an f16x2 call on the device-pointer path (bk_eval_device*), enqueued right behind the f16x2
// kernel on the same stream: a no-op unless that kernel raised the call's overflow tag.  A separate instantiation so
// that profiles keep the real fp32 launches and these (normally empty) ones apart.
template <int NB, bool GATED>
__global__ void __launch_bounds__(Geo<NB>::THREADS) bk_leaf_eval_kernel(const bk_eval_args a) {
    using G = Geo<NB>;
    if constexpr (GATED) {
        if (__builtin_nontemporal_load(a.gate) != a.gate_tag) return;   // uniform over the grid
        // "this call was redone": the call's own word in pinned host memory (the host counts the words that changed)
        if (a.gate_count && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(a.gate_counter, a.gate_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* actb = smem;
    const int dummy_byte = G::L3_BYTES;
    float* hs = reinterpret_cast<float*>(smem + G::L3_BYTES) + G::DUMMY_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // block -> (net, task).  Blocks are dealt round-robin over the 8 XCDs (b and b+8 share one), so
    // while both nets have tasks left bit 2 of the block id picks the net: each XCD's 4 MB L2 then
    // streams ONE net's 3.9 MB of weights.  The surplus tasks of the larger net follow linearly.
    // (Speed only; any placement is correct.)
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
    // wave grid and the weight ring (see conv_layer); layer 0's first two channel groups are requested before the staging
    using F = Tiles<NB>;
    const int wm = wave / F::WN, wn = wave - wm * F::WN;
    f32x4 Wr[F::RING][F::CTW];
    {
        const __amdgpu_buffer_rsrc_t wr0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.wfrag + wn * F::CTW * 256), 0, 0x7ffffff0, 0x00020000);
#pragma unroll
        for (int n = 0; n < F::RING - 1; ++n)
#pragma unroll
            for (int ct = 0; ct < F::CTW; ++ct)
                Wr[n][ct] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr0, lane * 16 + ct * 1024, n * 8192, 0));
    }
    stage_input<NB, G::THREADS>(a, actb, b0, nb, tid);
    __syncthreads();

    STAMP(1);
    constexpr int RT = F::RT;
    const int kq = lane >> 4;
    f32x4 acc[F::CTW][RT];
    // this lane's positions, decoded ONCE: LDS byte offsets for the activation-fragment reads of layer 0 /
    // layers 1..6 (tap (0,0), chunk kq) and for the epilogue stores (record + this wave's couts)
    lds_cchar* ap3[RT];
    unsigned valid = 0;
    // epilogue stores: record of (tile, lane) = ap3 + RP3 + REC3 - 16 kq (the tap-(1,1) position) + this wave's couts
    const int store_c = RP3 + REC3 - kq * 16 + (16 * F::CTW * wn + 4 * kq) * 4;
    f32x4 bv[F::CTW];
    // ---- layer 0: 5x5, 27 -> 128 ----
    const int row_i = wm * RT * 16 + (lane & 15);       // this lane's column of the row table
    {
        lds_cchar* ap0[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) ap0[rt] = (lds_cchar*)actb + (g_rows<NB>.a0[row_i + rt * 16] + kq * 16);
        conv_layer<F, true>(actb, P.wfrag, acc, lane, wm, wn * F::CTW, ap0, Wr);
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {                   // (fetched now rather than kept alive through layer 0)
        const int e = g_rows<NB>.a3v[row_i + rt * 16];
        ap3[rt] = (lds_cchar*)actb + ((e & ~1) + kq * 16);
        valid |= (unsigned)(e & 1) << rt;
    }
    load_bias<F>(bv, P.bias, wn, kq);                   // behind the conv (its second chain needs the registers): lands during the barrier
    STAMP(2);
    __syncthreads();  // everyone done reading the input planes
    // the 128-ch layout overlaps the input region: the halo must read as zero
    if constexpr (NB == 3) {
        // every data record is about to be overwritten in full (27 rows x 9 points x 128 channels): clear only the 38
        // halo records -- row 0, column 0 of rows 1..27, and the record behind the last row (20 KB instead of 147 KB)
        for (int i = tid; i < 38 * (REC3 / 16); i += G::THREADS) {
            const int rec = i / (REC3 / 16), ch = i - rec * (REC3 / 16);
            const int base = rec < 10 ? rec * REC3 : rec < 37 ? (rec - 9) * RP3 : G::NROWS3 * RP3;
            *reinterpret_cast<f32x4*>(actb + base + ch * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    } else {
        for (int i = tid; i < G::L3_BYTES / 16; i += G::THREADS) reinterpret_cast<f32x4*>(actb)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    STAMP(3);
    store_layer<F>(actb, acc, bv, ap3, store_c, valid, dummy_byte);
    STAMP(4);
    __syncthreads();
    STAMP(5);

    // ---- layers 1..6: 3x3, 128 -> 128, in place ----
#pragma unroll 1
    for (int L = 1; L < 7; ++L) {
        conv_layer<F, false>(actb, P.wfrag + BK_L0_FLOATS + (L - 1) * BK_L3_FLOATS, acc, lane, wm, wn * F::CTW, ap3, Wr);
        load_bias<F>(bv, P.bias + L * 128, wn, kq);
        STAMP(2 + 4 * L);
        __syncthreads();
        STAMP(3 + 4 * L);
        store_layer<F>(actb, acc, bv, ap3, store_c, valid, dummy_byte);
        STAMP(4 + 4 * L);
        __syncthreads();
        STAMP(5 + 4 * L);
    }

    // ---- heads: wave w handles board w ----
    if (wave < nb) run_heads<G>(a, P, actb, hs + wave * 96, net, lane, wave, b0 + wave);
    STAMP(30);
}

// ---- the cooperative form for small batches ----------------------------------------------------------------------------
// With fewer one-board tasks than CUs the form above leaves most of the chip idle and a batch takes the 0.33 ms one CU
// needs for a whole network (a 1600-rollout genmove is ~10 such batches of 40..80 boards).  Here S = SC x SR workgroups on
// S CUs share ONE board of one net: each computes one of SC ranges of output channels for one of SR ranges of the board's
// points in every layer, publishes that slice in global memory (L2), meets its S-1 peers at a counter and fetches what
// they computed into its own LDS image of the activations.  Every dot product runs in the same k order on the same
// instruction as in the forms above, so the results are bit-identical to theirs.
//   * splits: 8x1, 4x1, 2x1 (output channels only); 1x3 for 65..80 tasks, where only 3 CUs per board are to be had and 8
//     cout tiles do not divide by 3 (three ranges of two 16-point tiles, all 128 channels); 2x3 and 4x3 for the smallest
//     batches.  bk_coop_slices() picks by task count;
//   * waves: (8/SC cout tiles) x (RH groups of 6/SR/RH position tiles); weights, biases, LDS layout: as for 1 board;
//   * exchange buffer: [task][layer parity][point][128 slots] fp32 -- a workgroup that is one layer ahead writes the other
//     parity, and cannot get two ahead before every peer has arrived at the counter in between;
//   * coherence (peers may sit on another XCD with another L2) without cache-wide fences: slices are written and read with
//     device-scope (sc1) buffer accesses, every wave waits for its stores to be acknowledged before the workgroup
//     barrier, then ONE thread adds 1 to the task's counter (device-scope atomic) and polls until all S arrivals of this
//     layer are in.  The counter runs 0 .. 7S and the slice that computes the heads puts it back to 0;
//   * placement: blocks x + 8(S j + s) are the S slices of task 8j + x: with the round-robin dealing of blocks to XCDs they
//     share an L2 and follow each other in that XCD's dispatch order (a partly resident group waits only for blocks that
//     are dispatched before any later group's).  Speed only; nothing above depends on it;
//   * the poll is BOUNDED (COOP_SPIN_LIMIT polls, ~50 ms): if the peers do not show up -- the card shared with something
//     that holds CUs for that long -- the workgroup raises the flag coop_err, which travels to the host with the outputs,
//     and runs on to the end (the grid always drains); bk_wait then clears the counters and recomputes the request with
//     the one-CU form.  The failure is STICKY on the device until then: the workgroup also raises the engine's poison word
//     (bk_internal.h), and cooperative launches already queued behind the failed one check it at entry, raise their own
//     request's flag and do not wait for anybody -- their counters are not trustworthy and their outputs are redone too.
template <int SC, int SR, int RH>
struct CoopTiles {
    static_assert((SC == 1 || SC == 2 || SC == 4 || SC == 8) && (SR == 1 || SR == 3) && (6 / SR) % RH == 0 && SC * SR > 1,
                  "SC slices of 128/SC output channels x SR slices of 6/SR position tiles");
    static constexpr int S = SC * SR;                   // workgroups per board
    static constexpr int ROWT = 6 / SR;                 // position tiles per workgroup
    static constexpr int RT = ROWT / RH, CTW = 1;
    // The one-board tile set (tile_row<1>: tiles 0..2 interior, 3 x-edges, 4 y = 0, 5 y = 8) dealt out to the waves:
    //   SR == 1, RH == 2 (the 4x1 and 2x1 forms, 33..64 and 81..128 tasks): row group 0 = tiles {0, 1, 4}, row group 1 =
    //       {2, 3, 5} -- two tiles that run every tap and one y-edge tile that skips its outward taps (the row group plays
    //       the part `wm` plays for 3-board workgroups): 24 instead of 27 tile-taps per wave and 3x3 layer;
    //   every other form: no wave could get shorter by skipping (RH == 6: one tile per wave, the slowest decides; SR == 3:
    //       three workgroups of two tiles, one of them holds two interior tiles): tiles in the order of TILE below, every tap.
    static constexpr bool SKIP = SR == 1 && RH == 2;
    static constexpr int A0 = 0, A1 = SKIP ? 2 : RT, X0 = -1, X1 = -1, Y0a = SKIP ? 2 : -1, Y0b = SKIP ? 3 : -1, Y1 = -1;
    static constexpr bool WM_EDGES = SKIP, DB2 = true;
    static constexpr int RING = 4;
    // canonical tile of the workgroup's k-th tile (k = rh * RT + rt) in point range sr
    static constexpr int tile(int sr, int k) {
        constexpr int PAIRS[6] = {0, 1, 2, 4, 3, 5};    // SR == 3: {0,1} {2,4} {3,5}
        constexpr int HALVES[6] = {0, 1, 4, 2, 3, 5};   // SR == 1, RH == 2
        return SKIP ? HALVES[k] : SR == 3 ? PAIRS[2 * sr + k] : k;
    }
    // which point range (SR == 3) computes board point q
    static constexpr int owner(int q) {
        const int y = q / 9, x = q - 9 * y;
        const int i = (y - 1) * 7 + (x - 1);
        const int t = y == 0 ? 4 : y == 8 ? 5 : (x == 0 || x == 8 || i >= 48) ? 3 : i / 16;
        constexpr int SR_OF[6] = {0, 0, 1, 2, 1, 2};
        return SR == 3 ? SR_OF[t] : 0;
    }
    static constexpr int CT = 8 / SC;                   // cout tiles per workgroup
    static constexpr int NW = CT * RH, THREADS = 64 * NW;
    static constexpr bool FRONT = RT == 1;              // a wave that holds ONE tile: the channel group's only read right behind its first MFMA (sched_front)
    static constexpr int XCHG_FLOATS = 2 * 81 * 128;    // per task: two layer parities
};
constexpr int COOP_SPIN_LIMIT = 1 << 15;

template <int SC, int SR, int RH>
__global__ void __launch_bounds__((CoopTiles<SC, SR, RH>::THREADS)) bk_leaf_eval_coop_kernel(const bk_eval_args a) {
    using G = Geo<1>;
    using F = CoopTiles<SC, SR, RH>;
    constexpr int THREADS = F::THREADS, RT = F::RT, S = F::S;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* actb = smem;
    const int dummy_byte = G::L3_BYTES;
    float* hs = reinterpret_cast<float*>(smem + G::L3_BYTES) + G::DUMMY_FLOATS;
    __shared__ int dead;                                // a poll of this workgroup ran out: stop waiting

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x;
    const int slice = (bid >> 3) % S, task = ((bid >> 3) / S) * 8 + (bid & 7);
    if (task >= a.tasks_p + a.tasks_v) return;          // the grid is padded to whole groups of 8 tasks
    const int net = task >= a.tasks_p;
    const bk_net_params& P = a.net[net];
    const int bg = net ? a.off_v + task - a.tasks_p : a.off_p + task;
    float* xb = a.coop_xchg + (size_t)task * F::XCHG_FLOATS;
    unsigned int* cnt = a.coop_sync + task * BK_COOP_SYNC_STRIDE;   // one counter per 256 B: polls spread over the L2 channels

    const int sc = slice % SC, sr = slice / SC;         // this workgroup's cout range and point range
    const int wc = wave / RH, rh = wave - wc * RH;
    const int wn = sc * F::CT + wc;                     // this wave's cout tile (of 8)
    f32x4 Wr[F::RING][1];
    {
        const __amdgpu_buffer_rsrc_t wr0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.wfrag + wn * 256), 0, 0x7ffffff0, 0x00020000);
#pragma unroll
        for (int n = 0; n < F::RING - 1; ++n)
            Wr[n][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr0, lane * 16, n * 8192, 0));
    }
    if (tid == 0) {
        // an earlier launch of this engine timed out and the host has not cleared the counters yet: do not trust them
        const bool poisoned = __hip_atomic_load(a.coop_sync + BK_COOP_POISON_WORD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        if (poisoned) __hip_atomic_store(a.coop_err, a.coop_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        dead = poisoned;
    }
    STAMP(0);
    stage_input<1, THREADS>(a, actb, bg, 1, tid);
    __syncthreads();
    STAMP(1);

    const int kq = lane >> 4;
    f32x4 acc[1][RT];
    lds_cchar *ap0[RT], *ap3[RT];
    int storea[RT], xoff[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const TileRow fr = tile_row<1>(0, F::tile(sr, rh * RT + rt), lane & 15);
        ap0[rt] = (lds_cchar*)actb + (G::addr0(0, fr.y, fr.x) - 2 * RP0 - 2 * REC0 + kq * 16);
        ap3[rt] = (lds_cchar*)actb + (G::addr3(0, fr.y, fr.x) - RP3 - REC3 + kq * 16);
        storea[rt] = fr.valid ? G::addr3(0, fr.y, fr.x) + (16 * wn + 4 * kq) * 4 : dummy_byte;
        xoff[rt] = fr.valid ? (9 * fr.y + fr.x) * 128 + 16 * wn + 4 * kq : -1;
    }
    f32x4 bv[1];

    // which of this thread's chunks of the exchange (chunk i = tid + THREADS k: point i / 32, channels 4 (i % 32) ..) another
    // workgroup computes: the same in every layer
    unsigned foreign_mask = 0;
    {
        constexpr int OWN = 32 / SC, PER = (81 * 32 + THREADS - 1) / THREADS;
        static_assert(PER <= 32, "one bit per chunk");
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = tid + THREADS * k;
            if (i < 81 * 32 && !(((i & 31) / OWN) == sc && F::owner(i >> 5) == sr)) foreign_mask |= 1u << k;
        }
    }
    // own slice: LDS + exchange buffer; then meet the peers and fetch theirs.  After the last layer only slice 0 goes on.
    // Coherence without cache-wide fences (a buffer_wbl2 / buffer_inv per wave and layer cost more than the convolutions
    // once 30 groups shared an L2): the slices are written and read with device-scope (sc1) accesses, a wave waits for
    // its stores to be acknowledged before the workgroup barrier, and the counter is a device-scope atomic.
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(xb, 0, F::XCHG_FLOATS * 4, 0x00020000);
    constexpr int SC1 = 16;                             // cache-policy bit of the buffer builtins on gfx94x/95x
    auto exchange = [&](int L) -> bool {
        const int par = (L & 1) * (81 * 128 * 4);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[0][rt][e] + bv[0][e], 0.f);
            *reinterpret_cast<f32x4*>(actb + storea[rt]) = v;
            if (xoff[rt] >= 0) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), xr, xoff[rt] * 4, par, SC1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(3 + 4 * L);
        __syncthreads();
        const bool last = L == 6;
#ifdef BK_TEST_HOOKS
        if (a.coop_fault && L == 3 && task == 0 && slice == 1) return false;   // test hook: a peer that never arrives
#endif
        if (tid == 0) {
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!(last && slice != 0) && !dead) {
                const unsigned int target = (unsigned int)S * (L + 1);
                int spins = 0;
#pragma unroll 1
                while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > COOP_SPIN_LIMIT) {
                        // a plain system-scope store (every writer stores the same tag): the flag word may live in pinned
                        // host memory, next to the outputs the kernel writes there directly for small requests
                        __hip_atomic_store(a.coop_err, a.coop_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        __hip_atomic_store(a.coop_sync + BK_COOP_POISON_WORD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        dead = 1;
                        break;
                    }
                }
            }
        }
        if (last && slice != 0) return false;           // uniform over the workgroup
        __syncthreads();
        STAMP(4 + 4 * L);
        // the peers' slices: every 16-byte chunk (point q, channels 4c..4c+3) this workgroup did not compute itself
        constexpr int PER = (81 * 32 + THREADS - 1) / THREADS;
        auto foreign = [&](int i) { return (foreign_mask >> ((i - tid) / THREADS)) & 1u; };
        f32x4 v[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = tid + THREADS * k;
            if (foreign(i)) v[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, i * 16, par, SC1));
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = tid + THREADS * k, q = i >> 5, c = i & 31;
            if (foreign(i)) {
                const int y = q / 9, x = q - 9 * y;
                *reinterpret_cast<f32x4*>(actb + G::addr3(0, y, x) + c * 16) = v[k];
            }
        }
        __syncthreads();
        STAMP(5 + 4 * L);
        return true;
    };

    // ---- layer 0: 5x5, 27 -> 128 ----
    load_bias<F>(bv, P.bias, wn, kq);
    conv_layer<F, true>(actb, P.wfrag, acc, lane, rh, wn, ap0, Wr);
    STAMP(2);
    __syncthreads();
    for (int i = tid; i < G::L3_BYTES / 16; i += THREADS) reinterpret_cast<f32x4*>(actb)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    exchange(0);
    // ---- layers 1..6: 3x3, 128 -> 128 ----
#pragma unroll 1
    for (int L = 1; L < 7; ++L) {
        load_bias<F>(bv, P.bias + L * 128, wn, kq);
        conv_layer<F, false>(actb, P.wfrag + BK_L0_FLOATS + (L - 1) * BK_L3_FLOATS, acc, lane, rh, wn, ap3, Wr);
        STAMP(2 + 4 * L);
        __syncthreads();
        if (!exchange(L)) return;
    }
    if (tid == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    if (wave == 0) run_heads<G>(a, P, actb, hs, net, lane, 0, bg);
    STAMP(30);
}

template <int SC, int SR, int RH>
hipError_t launch_coop(const bk_eval_args& a, hipStream_t stream) {
    constexpr int S = SC * SR;
    static bool attr_set_dev[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    auto kern = bk_leaf_eval_coop_kernel<SC, SR, RH>;
    if (!attr_set_dev[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Geo<1>::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set_dev[dev] = true;
    }
    bk_eval_args args = a;
    args.tasks_p = a.B_policy - a.off_p;
    args.tasks_v = a.B_value - a.off_v;
    const int tasks = args.tasks_p + args.tasks_v;
    if (tasks == 0) return hipSuccess;
    const int grid = (tasks + 7) / 8 * 8 * S;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(CoopTiles<SC, SR, RH>::THREADS), Geo<1>::LDS_BYTES, stream, args);
    return hipGetLastError();
}

// ---- three boards on SC CUs ------------------------------------------------------------------------------------------------
// Between the whole-board forms there are gaps: 129..256 tasks take a round of one-board workgroups (298 us) and 257..512 one
// of two-board workgroups (530 us) however far they are from filling it -- and the requests of lock-step self-play sit in
// those gaps (a 256-game pool asks for ~370 tasks per step, a 128-game one for ~185).  Here SC = 2 or 4 workgroups on SC CUs
// share THREE boards of one net on the 3-board tile set (Tiles<3>: 63 of 72 tile-taps): each computes 128 / SC output channels
// of all 243 points in every layer, publishes them (L2), meets its peers at the group's counter and fetches theirs into its
// own LDS image -- the exchange of the one-board cooperative form above, with its coherence, placement and bounded poll.  1.5
// boards per CU (up to 384 tasks) and 0.75 (up to 192).  Every dot product runs in the same order on the same instruction as in
// every other form: bit-identical results.
template <int SC>
struct Coop3Tiles {
    static_assert(SC == 2 || SC == 4, "three boards on two or four CUs");
    static constexpr int S = SC, WM = 2, RT = 8, CTW = 1, TB = 0;
    static constexpr int CT = 8 / SC;                   // cout tiles per workgroup
    static constexpr int NW = WM * CT, THREADS = 64 * NW;
    static constexpr int A0 = 1, A1 = 6, X0 = 0, X1 = -1, Y0a = 6, Y0b = 8, Y1 = -1;   // as Tiles<3>
    static constexpr bool WM_EDGES = true, DB2 = true, FRONT = false;
    static constexpr int RING = 4;
    static constexpr int XCHG_FLOATS = 2 * 243 * 128;   // per group: two layer parities
};
static_assert(Coop3Tiles<2>::XCHG_FLOATS * 4 * BK_COOP3_MAX_GROUPS <= BK_COOP_XCHG_BYTES, "exchange buffer");
// Three boards on EIGHT CUs (round 5; 81..96 tasks = up to 32 groups x 8 CUs: every CU of the chip busy, where the 2-CUs-per-
// board form of the one-board tile set leaves a quarter of them idle): a workgroup computes ONE 16-cout tile of all 243 points.
// With one cout tile the two position groups would be two waves of eight tiles on two of the four SIMDs, so each group's tiles
// [x-edge | 5 interior | y-edge a | y-edge b] are split between two waves: half 0 = [x-edge | 3 interior] (6 + 27 = 33 tile-taps
// per 3x3 layer), half 1 = [2 interior | y-edge a | y-edge b] (18 + 12 = 30): four waves, one per SIMD.  The halves are two
// tile-class layouts, i.e. two instantiations of conv_layer; a wave runs the one that belongs to it (wave-uniform).
template <int H>
struct Coop3Tiles8 {
    static constexpr int S = 8, WM = 2, RT = 4, CTW = 1, TB = 4 * H, CT = 1;
    static constexpr int NW = 4, THREADS = 64 * NW;
    static constexpr int A0 = H == 0 ? 1 : 0, A1 = H == 0 ? 4 : 2, X0 = H == 0 ? 0 : -1, X1 = -1;
    static constexpr int Y0a = H == 0 ? -1 : 2, Y0b = H == 0 ? -1 : 4, Y1 = -1;
    static constexpr bool WM_EDGES = true, DB2 = true, FRONT = false;
    static constexpr int RING = 4;
    static constexpr int XCHG_FLOATS = 2 * 243 * 128;
};

// the body of the three-boards kernels for one wave's tile set F: the wave is (position group wm, cout tile wn of 8) of slice
// sc of group grp; `dead` is the workgroup's "a poll ran out" word.  Every wave of a workgroup runs the same sequence of
// barriers, whichever F it instantiates.
template <class F>
__device__ __forceinline__ void coop3_body(const bk_eval_args& a, char* smem, int* dead, int tid, int lane, int wave, int wm, int wn,
                                           int sc, int grp) {
    using G = Geo<3>;
    constexpr int THREADS = F::THREADS, RT = F::RT, S = F::S;
    char* actb = smem;
    const int dummy_byte = G::L3_BYTES;
    float* hs = reinterpret_cast<float*>(smem + G::L3_BYTES) + G::DUMMY_FLOATS;
    const int net = grp >= a.tasks_p;
    const bk_net_params& P = a.net[net];
    const int b0 = (net ? a.off_v + (grp - a.tasks_p) * 3 : a.off_p + grp * 3);
    const int nb = min(3, (net ? a.B_value : a.B_policy) - b0);
    float* xb = a.coop_xchg + (size_t)grp * F::XCHG_FLOATS;
    unsigned int* cnt = a.coop_sync + grp * BK_COOP_SYNC_STRIDE;

    f32x4 Wr[F::RING][1];
    {
        const __amdgpu_buffer_rsrc_t wr0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.wfrag + wn * 256), 0, 0x7ffffff0, 0x00020000);
#pragma unroll
        for (int n = 0; n < F::RING - 1; ++n)
            Wr[n][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr0, lane * 16, n * 8192, 0));
    }
    if (tid == 0) {
        const bool poisoned = __hip_atomic_load(a.coop_sync + BK_COOP_POISON_WORD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        if (poisoned) __hip_atomic_store(a.coop_err, a.coop_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        *dead = poisoned;
    }
    STAMP(0);
    stage_input<3, THREADS>(a, actb, b0, nb, tid);
    __syncthreads();
    STAMP(1);

    const int kq = lane >> 4;
    f32x4 acc[1][RT];
    lds_cchar* ap3[RT];
    unsigned valid = 0;
    int xoff[RT];
    const int store_c = RP3 + REC3 - kq * 16 + (16 * wn + 4 * kq) * 4;
    f32x4 bv[1];
    const int row_i = (wm * 8 + F::TB) * 16 + (lane & 15);      // this wave's first tile in the 3-board row table
    {
        lds_cchar* ap0[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) ap0[rt] = (lds_cchar*)actb + (g_rows<3>.a0[row_i + rt * 16] + kq * 16);
        conv_layer<F, true>(actb, P.wfrag, acc, lane, wm, wn, ap0, Wr);
        STAMP(2);
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int e = g_rows<3>.a3v[row_i + rt * 16];
        ap3[rt] = (lds_cchar*)actb + ((e & ~1) + kq * 16);
        valid |= (unsigned)(e & 1) << rt;
        const TileRow fr = tile_row<3>(wm, F::TB + rt, lane & 15);
        xoff[rt] = fr.valid ? (81 * fr.b + 9 * fr.y + fr.x) * 128 + 16 * wn + 4 * kq : -1;
    }
    load_bias<F>(bv, P.bias, wn, kq);
    __syncthreads();                                    // everyone done reading the input planes
    for (int i = tid; i < 38 * (REC3 / 16); i += THREADS) {   // the 38 halo records (see the 3-board form)
        const int rec = i / (REC3 / 16), ch = i - rec * (REC3 / 16);
        const int base = rec < 10 ? rec * REC3 : rec < 37 ? (rec - 9) * RP3 : G::NROWS3 * RP3;
        *reinterpret_cast<f32x4*>(actb + base + ch * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(xb, 0, F::XCHG_FLOATS * 4, 0x00020000);
    constexpr int SC1 = 16;
    // own channels: LDS + exchange buffer; meet; the peers' channels into LDS.  After the last layer only slice 0 goes on.
    auto exchange = [&](int L) -> bool {
        const int par = (L & 1) * (243 * 128 * 4);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            typedef __attribute__((address_space(3))) char lds_char;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[0][rt][e] + bv[0][e], 0.f);
            lds_char* wp = (valid >> rt) & 1 ? (lds_char*)(ap3[rt]) + store_c : (lds_char*)actb + dummy_byte;
            *reinterpret_cast<__attribute__((address_space(3))) f32x4*>(wp) = v;
            if (xoff[rt] >= 0) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), xr, xoff[rt] * 4, par, SC1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(3 + 4 * L);
        __syncthreads();
        const bool last = L == 6;
#ifdef BK_TEST_HOOKS
        if (a.coop_fault && L == 3 && grp == 0 && sc == 1) return false;   // test hook: a peer that never arrives
#endif
        if (tid == 0) {
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!(last && sc != 0) && !*dead) {
                const unsigned int target = (unsigned int)S * (L + 1);
                int spins = 0;
#pragma unroll 1
                while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > COOP_SPIN_LIMIT) {
                        __hip_atomic_store(a.coop_err, a.coop_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        __hip_atomic_store(a.coop_sync + BK_COOP_POISON_WORD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        *dead = 1;
                        break;
                    }
                }
            }
        }
        if (last && sc != 0) return false;              // uniform over the workgroup
        __syncthreads();
        STAMP(4 + 4 * L);
        // the peers' channels: chunk j = (point q, foreign 16-byte channel chunk cf), BATCH in flight per thread
        constexpr int OWN = 32 / S, FC = 32 - OWN, N = 243 * FC, PER = (N + THREADS - 1) / THREADS;
        // loads in flight per thread: all of them where the registers are there -- the 256-thread form (S == 8: one wave per SIMD, 27
        // chunks per thread) fetched in three rounds of 12 and spent 10.8 k cycles per layer on 106 KB, the 512-thread forms 5.2 k
        constexpr int BATCH = S == 2 ? 8 : S == 4 ? 12 : PER;
#pragma unroll 1
        for (int k0 = 0; k0 < PER; k0 += BATCH) {
            f32x4 v[BATCH];
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                const int j = tid + THREADS * (k0 + k);
                if (j < N) {
                    const int q = j / FC, cf = j - q * FC, c = cf < sc * OWN ? cf : cf + OWN;
                    v[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, q * 512 + c * 16, par, SC1));
                }
            }
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                const int j = tid + THREADS * (k0 + k);
                if (j < N) {
                    const int q = j / FC, cf = j - q * FC, c = cf < sc * OWN ? cf : cf + OWN;
                    const int b = q / 81, r = q - 81 * b, y = r / 9, x = r - 9 * y;
                    *reinterpret_cast<f32x4*>(actb + G::addr3(b, y, x) + c * 16) = v[k];
                }
            }
        }
        __syncthreads();
        STAMP(5 + 4 * L);
        return true;
    };

    exchange(0);
#pragma unroll 1
    for (int L = 1; L < 7; ++L) {
        conv_layer<F, false>(actb, P.wfrag + BK_L0_FLOATS + (L - 1) * BK_L3_FLOATS, acc, lane, wm, wn, ap3, Wr);
        STAMP(2 + 4 * L);
        load_bias<F>(bv, P.bias + L * 128, wn, kq);
        __syncthreads();
        if (!exchange(L)) return;
    }
    if (tid == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    if (wave < nb) run_heads<G>(a, P, actb, hs + wave * 96, net, lane, wave, b0 + wave);
    STAMP(30);
}

template <int SC>
__global__ void __launch_bounds__((SC == 8 ? Coop3Tiles8<0>::THREADS : Coop3Tiles<SC == 8 ? 4 : SC>::THREADS)) bk_leaf_eval_coop3_kernel(const bk_eval_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int dead;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x;
    // blocks x + 8 (S j + s) are the S slices of group 8 j + x: one XCD, one L2 (see the one-board form)
    const int sc = (bid >> 3) % SC, grp = ((bid >> 3) / SC) * 8 + (bid & 7);
    if (grp >= a.tasks_p + a.tasks_v) return;           // the grid is padded to whole sets of 8 groups
    if constexpr (SC == 8) {
        // four waves = (position group, half of its tiles); the workgroup's one cout tile is its slice number
        const int wm = wave >> 1;
        if (wave & 1) coop3_body<Coop3Tiles8<1>>(a, smem, &dead, tid, lane, wave, wm, sc, sc, grp);
        else coop3_body<Coop3Tiles8<0>>(a, smem, &dead, tid, lane, wave, wm, sc, sc, grp);
    } else {
        using F = Coop3Tiles<SC == 8 ? 4 : SC>;
        const int wm = wave / F::CT, wn = sc * F::CT + (wave - wm * F::CT);   // position group; this wave's cout tile (of 8)
        coop3_body<F>(a, smem, &dead, tid, lane, wave, wm, wn, sc, grp);
    }
}

template <int SC>
hipError_t launch_coop3(const bk_eval_args& a, hipStream_t stream) {
    static bool attr_set_dev[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    auto kern = bk_leaf_eval_coop3_kernel<SC>;
    constexpr int THREADS = SC == 8 ? Coop3Tiles8<0>::THREADS : Coop3Tiles<SC == 8 ? 4 : SC>::THREADS;
    if (!attr_set_dev[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Geo<3>::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set_dev[dev] = true;
    }
    bk_eval_args args = a;
    args.tasks_p = (a.B_policy - a.off_p + 2) / 3;
    args.tasks_v = (a.B_value - a.off_v + 2) / 3;
    const int groups = args.tasks_p + args.tasks_v;
    if (groups == 0) return hipSuccess;
    if (groups > BK_COOP3_MAX_GROUPS || (SC == 8 && groups > BK_COOP3_FORM_8_MAX_GROUPS)) return hipErrorInvalidValue;
    const int grid = (groups + 7) / 8 * 8 * SC;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), Geo<3>::LDS_BYTES, stream, args);
    return hipGetLastError();
}

template <int NB, bool GATED>
hipError_t launch_nb(const bk_eval_args& a, hipStream_t stream) {
    static bool attr_set_dev[64] = {false};  // the attribute is per device: one flag per device ordinal
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    bool& attr_set = attr_set_dev[dev];
    auto kern = bk_leaf_eval_kernel<NB, GATED>;
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
    hipLaunchKernelGGL(kern, dim3(grid), dim3(Geo<NB>::THREADS), Geo<NB>::LDS_BYTES, stream, args);
    return hipGetLastError();
}

// ---- executed work, counted from the tile tables the kernels run on (bk_plan_flops, bench.py) ----------------------------
// (tile, tap) pairs one position group of a workgroup executes in a kw x kw layer: interior tiles run every tap, an edge
// tile skips the kw * (kw/2) taps that point off the board on its side -- exactly the branches of conv_layer::do_group.
template <class F>
constexpr long tile_taps(int kw) {
    const int taps = kw * kw, skip = kw * (kw / 2);
    const int nx = (F::X0 >= 0) + (F::X1 >= 0);
    const int ny = (F::Y0a >= 0 ? F::Y0b - F::Y0a : 0) + (F::Y1 >= 0);
    return (long)(F::A1 - F::A0) * taps + (long)(nx + ny) * (taps - skip);
}
static_assert(tile_taps<Tiles<3>>(3) == 63 && tile_taps<Tiles<3>>(5) == 170, "3 boards: 63 of 72 / 170 of 200 tile-taps");
static_assert(tile_taps<Coop3Tiles8<0>>(3) == 33 && tile_taps<Coop3Tiles8<1>>(3) == 30 && tile_taps<Coop3Tiles8<0>>(3) + tile_taps<Coop3Tiles8<1>>(3) == tile_taps<Tiles<3>>(3),
              "three boards on eight CUs: the two halves of a position group's tiles, 33 + 30 = 63 tile-taps");
static_assert(tile_taps<Tiles<2>>(3) == 87 && tile_taps<Tiles<1>>(3) == 48 && tile_taps<Tiles<1>>(5) == 130, "2 boards: 87 of 99; 1 board: 48 of 54");
// FLOP the matrix unit executes for ONE net on one NB-board workgroup: per (tile, tap) 7 k-steps in layer 0 (28 input
// slots) resp. 32 in a 3x3 layer, x 8 cout tiles, of v_mfma_f32_16x16x4_f32 (2 * 16 * 16 * 4 = 2,048 FLOP each)
template <int NB>
constexpr double mfma_flop_workgroup() {
    using F = Tiles<NB>;
    return 2048.0 * 8 * F::WM * (double)(tile_taps<F>(5) * 7 + 6 * tile_taps<F>(3) * 32);
}
// the cooperative forms deal the one-board tile set (6 position tiles x 8 cout tiles) out to their slices: with the y-edge
// tiles' outward taps skipped (SKIP: as the one-CU form) or every tap of every tile
static_assert(2 * tile_taps<CoopTiles<4, 1, 2>>(3) == tile_taps<Tiles<1>>(3) && CoopTiles<1, 3, 1>::ROWT * 3 == Tiles<1>::RT, "coop = 1-board tiles");
constexpr double coop_flop_task(bool skip) {
    return skip ? mfma_flop_workgroup<1>() : 2048.0 * 8 * (double)(6 * 25 * 7 + 6 * 6 * 9 * 32);
}

}  // namespace

double bk_coop_mfma_flop_per_task(int slices) {          // executed MFMA FLOP of one (net, board) task in the cooperative form
    return coop_flop_task(slices == 4 || slices == 2);
}

double bk_mfma_flop_per_workgroup(int nb) {
    return nb == 3 ? mfma_flop_workgroup<3>() : nb == 2 ? mfma_flop_workgroup<2>() : mfma_flop_workgroup<1>();
}


// Picks boards-per-workgroup.  A CU's matrix pipes are the bound, so the cost of a choice is (workgroup rounds over
// the CUs) x (time of one round of nb-board workgroups, one per CU).  NB=1 workgroups are small enough to sit two per
// CU but then share the pipes, so that buys nothing here.
//   f16x2: measured round times 86 : 150 : 195 us
//   fp32:  measured round times 0.33 : 0.55 : 0.754 ms
long bk_launch_cost(int B_policy, int B_value, int nb, int n_cu, int precision) {
    static const int t16[4] = {0, 44, 77, 100}, t32[4] = {0, 44, 73, 100};
    const long wgs = (B_policy + nb - 1) / nb + (B_value + nb - 1) / nb;
    return (wgs + n_cu - 1) / n_cu * (precision == BK_PRECISION_F16X2 ? t16 : t32)[nb];
}

int bk_pick_nb(int B_policy, int B_value, int n_cu, int precision, const bk_plan_opts& o) {
    if (o.force_nb >= 1 && o.force_nb <= 3) return o.force_nb;
    int best = 3;
    long best_cost = -1;
    for (int nb = 3; nb >= 1; --nb) {
        const long cost = bk_launch_cost(B_policy, B_value, nb, n_cu, precision);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = nb; }
    }
    return best;
}

// slices per board of the cooperative form for `tasks` one-board tasks on n_cu CUs; 0: use the ordinary forms
int bk_coop_slices(int tasks, int n_cu, const bk_plan_opts& o) {
    if (tasks <= 0 || tasks > BK_COOP_MAX_TASKS || o.force_nb) return 0;
    if (o.coop >= 0) {
        const int v = o.coop;
        if (v == 0) return 0;
        if ((v == 2 || v == 3 || v == 4 || v == 6 || v == 8 || v == 12) && (tasks + 7) / 8 * 8 * v <= 2 * n_cu) return v;
    }
    // measured (tools/coop_probe.py, us per call; one CU per board: 330), CUs per board:
    //   12 (4 cout x 3 point ranges): 68 at 2 tasks (8: 85), 106 at 9;   8: 104 .. 106 (9 .. 32 tasks);   6 (2 x 3): 104 .. 107
    //   (.. 33);   4: 111 .. 112 (.. 64);   3 (point ranges): 146 (.. 80);   2: 190 (.. 128)
    // blocks are dealt to the 8 XCDs in turn and task t sits on XCD t % 8: what has to fit is ceil(tasks / 8) groups on the
    // n_cu / 8 CUs of one XCD (85 tasks x 3 slices = 255 workgroups, but 33 of them on each of five XCDs: 239 us)
    // round 6 (tools/coop_forced_probe.py): with the one-tile waves' reads up front the 12-CU form also wins where TWO boards share an
    // XCD -- 9..16 tasks 67 us against the 8-CU form's 80 (at 17 tasks and beyond, three boards per XCD, it would put two workgroups
    // on a CU: 84 us)
    const int per_xcd = (tasks + 7) / 8, cus = n_cu / 8;
    for (int sl : {12, 8, 6, 4, 3, 2})
        if (per_xcd * sl <= cus) return sl;
    return 0;
}

// three boards on 2 / 4 CUs for the requests between the whole-board forms' ranges; 0: use the ordinary forms.
// Groups (three boards of one net) are dealt to the XCDs like the one-board form's tasks: ceil(groups / 8) x SC CUs of one XCD.
int bk_coop3_form(int B_policy, int B_value, int n_cu, const bk_plan_opts& o) {
    const int tasks = B_policy + B_value, groups = (B_policy + 2) / 3 + (B_value + 2) / 3;
    if (tasks <= 0 || groups > BK_COOP3_MAX_GROUPS) return 0;
    const int per_xcd = (groups + 7) / 8, cus = n_cu / 8;
    if (o.coop == 0 || o.force_nb) return 0;            // coop = 0: no cooperative launch of either kind
    if (o.coop3 >= 0) {
        const int v = o.coop3;
        if (v == 0) return 0;
        if ((v == 2 || v == 4) && per_xcd * v <= cus) return v == 2 ? BK_COOP3_FORM_2 : BK_COOP3_FORM_4;
        if (v == 8 && per_xcd * 8 <= cus && groups <= BK_COOP3_FORM_8_MAX_GROUPS) return BK_COOP3_FORM_8;
    }
    // measured (tools/coop_probe.py): see DESIGN 3
    if (tasks >= BK_COOP3_FORM_8_MIN && tasks <= BK_COOP3_FORM_8_MAX && groups <= BK_COOP3_FORM_8_MAX_GROUPS && per_xcd * 8 <= cus && BK_COOP3_FORM_8_DEFAULT)
        return BK_COOP3_FORM_8;
    if (tasks > BK_COOP_MAX_TASKS && tasks <= BK_COOP3_FORM_4_MAX && per_xcd * 4 <= cus) return BK_COOP3_FORM_4;
    if (tasks > n_cu && tasks <= BK_COOP3_FORM_2_MAX && per_xcd * 2 <= cus) return BK_COOP3_FORM_2;
    return 0;
}

// the cooperative form of a request, if any: the three-boards-on-eight-CUs form where it applies (in its range it replaces the
// 2-CUs-per-board form), else 2..12 CUs per board by task count, else three boards on 4 / 2 CUs; 0: whole-board workgroups
int bk_coop_form(int B_policy, int B_value, int n_cu, const bk_plan_opts& o) {
    const int f3 = bk_coop3_form(B_policy, B_value, n_cu, o);
    if (f3 == BK_COOP3_FORM_8 && o.coop < 0) return f3;     // (a forced number of CUs per board goes first)
    const int slices = bk_coop_slices(B_policy + B_value, n_cu, o);
    return slices ? slices : f3;
}

hipError_t bk_launch_leaf_eval_coop(const bk_eval_args& a, int slices, hipStream_t stream) {
    // wave grids measured (us per call at 2 / 63 tasks): 4 slices: 2 row groups 111 / 116, 3: 138 / 145, 6: 118 / 128;
    // 8 slices: 6 row groups 86, 3: 84;  2 slices: 2 row groups 189 / 194, 3: 200 / 204.  Every wave arriving and polling
    // for itself (no workgroup barriers around the meeting point): 177 at 63 tasks -- four times the pollers on the counters
    switch (slices) {
        case BK_COOP3_FORM_8: return launch_coop3<8>(a, stream);   // three boards on eight CUs
        case BK_COOP3_FORM_2: return launch_coop3<2>(a, stream);   // three boards on two CUs
        case BK_COOP3_FORM_4: return launch_coop3<4>(a, stream);   // three boards on four CUs
        case 12: return launch_coop<4, 3, 2>(a, stream);  // 4 cout ranges x 3 point ranges: 4 waves x 1 tile
        case 6: return launch_coop<2, 3, 2>(a, stream);   // 2 x 3: 8 waves x 1 tile
        case 8: return launch_coop<8, 1, 6>(a, stream);
        case 4: return launch_coop<4, 1, 2>(a, stream);
        case 3: return launch_coop<1, 3, 1>(a, stream);   // 3 point ranges x all 128 channels: 8 waves x 2 tiles
        default: return launch_coop<2, 1, 2>(a, stream);
    }
}

hipError_t bk_launch_leaf_eval(const bk_eval_args& a, int nb, hipStream_t stream) {
    if (a.gate) {
        switch (nb) {
            case 1: return launch_nb<1, true>(a, stream);
            case 2: return launch_nb<2, true>(a, stream);
            default: return launch_nb<3, true>(a, stream);
        }
    }
    switch (nb) {
        case 1: return launch_nb<1, false>(a, stream);
        case 2: return launch_nb<2, false>(a, stream);
        default: return launch_nb<3, false>(a, stream);
    }
}
