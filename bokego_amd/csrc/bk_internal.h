// bk_internal.h -- shared between the engine (host) and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// fragment-ordered weight buffer geometry (floats)
// per layer: [tap][group of 16 input slots][cout tile (8 x 16)][lane (64)][4]
// layer 0: 25 taps x 2 groups x 2048   layers 1..6: 9 x 8 x 2048
#define BK_L0_FLOATS (25 * 2 * 2048)
#define BK_L3_FLOATS (9 * 8 * 2048)
#define BK_WFRAG_FLOATS (BK_L0_FLOATS + 6 * BK_L3_FLOATS)
#define BK_WFRAG_PAD_FLOATS 16384  // the weight prefetch runs up to 64 KiB past the last layer (7 groups of 8 KiB, + a wave's offset)

// f16x2 path: per layer [k16 step][cout tile (4)][piece hi/lo][lane (64)][8 halfs] = 4096 halfs per step
#define BK16_L0_STEPS 52                      // 25 taps x 2 steps, zero-padded to a multiple of 4
#define BK16_L0_HALFS (BK16_L0_STEPS * 4096)
#define BK16_L3_HALFS (72 * 4096)             // 9 taps x 8 steps
#define BK16_WFRAG_HALFS (BK16_L0_HALFS + 6 * BK16_L3_HALFS)
#define BK16_WFRAG_PAD_HALFS (4 * 4096)       // the W prefetch runs 3 steps past the last layer

#define BK_PRECISION_F32 0
#define BK_PRECISION_F16X2 1

#define BK_FEATS_F32_ 0
#define BK_FEATS_U8_ 1
#define BK_FEATS_POS_ 2   // `feats` = 192-byte position records: the leaf kernel computes the planes itself (small requests; bk_encode_dev.h)

struct bk_net_params {
    const float* wfrag;    // BK_WFRAG_FLOATS (+pad), BatchNorm folded, fragment order of conv_layer (bk_kernels.hip);
                           // channels of layers 0..5 in the slot order bk_slot_perm
    const float* bias;     // [7][128] folded conv bias in slot order
    const float* head_w;   // [128]  (value net: BatchNorm2d(1) folded in)
    const float* head_b;   // [81]
    const float* lin1_wt;  // value: [81][64] = lin1.weight^T with BatchNorm1d folded
    const float* lin1_b;   // value: [64]
    const float* lin2_w;   // value: [64]
    float lin2_b;
    // f16x2 path
    const _Float16* wfrag16;  // BK16_WFRAG_HALFS (+pad): folded weights * 2^e_l as fp16 hi/lo fragments
    const float* bias16;      // [7][128] sa * folded bias
    float cscale16[7];        // sa_out / (sa_in * 2^e_l)
    float inv_sa16;           // 1 / sa
};

struct bk_eval_args {
    bk_net_params net[2];  // [0] policy, [1] value
    const void* feats;     // [B][27][9][9] f32 or u8 (device), or [B] position records of 192 bytes (BK_FEATS_POS_)
    int feats_dtype;
    int B_policy;          // PolicyNet runs on positions [0, B_policy)   (0: not at all)
    int B_value;           // ValueNet  runs on positions [0, B_value)
    int off_p, off_v;      // this launch covers positions [off_p, B_policy) / [off_v, B_value) only
    int tasks_p, tasks_v;  // filled by the launcher: ceil(B_x / NB)
    float* logits;         // [B][81] or null
    float* probs;          // [B][81] or null
    float* values;         // [B] or null
    unsigned int* overflow;      // f16x2: raised to overflow_tag (atomicMax) if an activation left the fp16 range
    unsigned int overflow_tag;   //   (result unreliable); host-buffer path: 1, device-pointer path: the call's sequence number
    // exact-fp32 kernel launched as the REDO of an f16x2 call on the device-pointer path: every workgroup returns at
    // once unless gate[0] == gate_tag (the f16x2 kernel of the same call raised the call's flag word)
    unsigned int* gate;
    unsigned int* gate_counter;  // the call's "redone" word (pinned host memory): receives gate_tag
    unsigned int gate_tag;
    int gate_count;              // this launch is the call's last one: it does the counting (a call may be two launches)
    // cooperative (cout-split) launches for small batches, bk_kernels.hip: exchange buffer [BK_COOP_MAX_TASKS][2][81][128]
    // fp32, one arrival counter per task (zero between launches), and the flag a workgroup raises to coop_tag when its
    // peers did not show up in time (word 1 of the ticket's flag block: bk_wait then redoes the request)
    float* coop_xchg;
    unsigned int* coop_sync;
    unsigned int* coop_err;
    unsigned int coop_tag;
    int coop_fault;              // -DBK_TEST_HOOKS builds only (option "coop_fault"): slice 1 of task 0 leaves before layer 3's meeting point
    unsigned long long* stamps;  // diagnostic builds (-DBK_STAMPS) only: [block][wave][32] s_memtime
};

// fp32 kernel: real channel held by slot s of a layer-0..5 output record (a bit permutation inside each block of 16):
// slot = 16g + 4kq + j is what MFMA k-step j of channel group g consumes from lane quad kq; chosen so that the k order
// of every dot product equals the round-1 kernel's (bit-identical results)
static inline int bk_slot_perm(int s) { return (s & ~15) | (((s >> 1) & 1) << 3) | (((s >> 2) & 1) << 2) | ((s & 1) << 1) | ((s >> 3) & 1); }

#define BK_DEV_FLAGS 256   // device-pointer path: overflow flag words, one per call in flight (bk_engine.cpp, d_dev_flag)

#define BK_POS_BYTES 192  // sizeof(bk_pos), include/bokego_go.h

// feature planes [B][27][9][9] u8 from B position records (bk_encode.hip)
hipError_t bk_launch_encode(const void* d_pos, int B, uint8_t* d_planes, hipStream_t stream);

// What used to be read from the environment on every request (BK_FORCE_NB, BK_NO_SPLIT, BK_COOP, BK_COOP3): the launch planner's
// switches, held per engine (read from the environment ONCE, in bk_engine_create; bk_engine_set_option changes them on a live
// engine).  The pure planner entry points of the C ABI (bk_plan_query / bk_plan_flops) use the defaults.
struct bk_plan_opts {
    int force_nb = 0;    // 1..3: every whole-board launch uses workgroups of that many boards
    int no_split = 0;    // 1: never k whole rounds + a tail launch
    int coop = -1;       // -1: by task count; 0: no cooperative launch of either kind; 2/3/4/6/8/12: that many CUs per board where it fits
    int coop3 = -1;      // -1: by task count; 0: no three-boards form; 2 / 4: that form where it fits
};
int bk_pick_nb(int B_policy, int B_value, int n_cu, int precision, const bk_plan_opts& o = bk_plan_opts());
long bk_launch_cost(int B_policy, int B_value, int nb, int n_cu, int precision);  // modelled time of one launch (arbitrary units)
hipError_t bk_launch_leaf_eval(const bk_eval_args& a, int nb, hipStream_t stream);
double bk_coop_mfma_flop_per_task(int slices);   // ... of one task in the cooperative form with `slices` CUs per board
double bk_mfma_flop_per_workgroup(int nb);   // fp32 kernel: executed MFMA FLOP of one net on one nb-board workgroup (tile tables)
#define BK_COOP_MAX_TASKS 128
#define BK_COOP_SYNC_STRIDE 64   // unsigned ints between two tasks' arrival counters
// behind the counters: the engine's POISON word.  A cooperative workgroup that gives up waiting raises it (beside its own
// request's coop_err); until the host has cleared it together with the counters (bk_wait of the failed ticket, stream-ordered)
// every cooperative launch that was already queued behind the failed one sees it at entry, raises ITS request's coop_err and
// runs through without waiting for anybody: the counters it would meet at are not trustworthy, and its outputs are redone too.
#define BK_COOP_POISON_WORD (BK_COOP_MAX_TASKS * BK_COOP_SYNC_STRIDE)
#define BK_COOP_SYNC_WORDS (BK_COOP_POISON_WORD + BK_COOP_SYNC_STRIDE)
int bk_coop_slices(int tasks, int n_cu, const bk_plan_opts& o = bk_plan_opts());   // 0: not a cooperative case
// three boards of one net on 2 / 4 CUs (bk_kernels.hip, bk_leaf_eval_coop3_kernel): form codes beside the one-board form's 2..12
#define BK_COOP3_FORM_2 102
#define BK_COOP3_FORM_4 104
#define BK_COOP3_FORM_8 108                           // three boards on EIGHT CUs: 81..96 tasks, at most 32 groups (32 x 8 CUs = the chip)
#define BK_COOP3_FORM_8_MIN 81
#define BK_COOP3_FORM_8_MAX 96
#define BK_COOP3_FORM_8_MAX_GROUPS 32
#define BK_COOP3_FORM_8_DEFAULT 1                     // 1: the planner picks it by itself in its range (else only option coop3 = 8)
#define BK_COOP3_MAX_GROUPS 128                       // 128 groups x 2 CUs = the chip
#define BK_COOP3_FORM_4_MAX 192                       // tasks: up to here four CUs per three boards ...
#define BK_COOP3_FORM_2_MAX 384                       // ... and two up to here (beyond: whole-board workgroups)
#define BK_COOP_XCHG_BYTES ((size_t)BK_COOP3_MAX_GROUPS * 2 * 243 * 128 * 4)   // the exchange buffer serves both forms
int bk_coop3_form(int B_policy, int B_value, int n_cu, const bk_plan_opts& o = bk_plan_opts());   // 0 / BK_COOP3_FORM_2 / BK_COOP3_FORM_4
int bk_coop_form(int B_policy, int B_value, int n_cu, const bk_plan_opts& o = bk_plan_opts());   // any of the above, or 0
hipError_t bk_launch_leaf_eval_coop(const bk_eval_args& a, int slices, hipStream_t stream);
hipError_t bk_launch_leaf_eval_f16(const bk_eval_args& a, int nb, hipStream_t stream);
