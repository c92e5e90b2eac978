// bk_internal.h -- shared between the engine (host) and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// fragment-ordered weight buffer geometry (floats)
// per layer: [tap][group of 8 cin][cout tile (4 x 32)][lane (64)][4]
// layer 0: 25 taps x 4 groups x 1024   layers 1..6: 9 x 16 x 1024
#define BK_L0_FLOATS (25 * 4 * 1024)
#define BK_L3_FLOATS (9 * 16 * 1024)
#define BK_WFRAG_FLOATS (BK_L0_FLOATS + 6 * BK_L3_FLOATS)
#define BK_WFRAG_PAD_FLOATS 4096  // the B prefetch reads one block (4 groups, 16 KiB) past the last layer

#define BK_FEATS_F32_ 0
#define BK_FEATS_U8_ 1

struct bk_net_params {
    const float* wfrag;    // BK_WFRAG_FLOATS (+pad), BatchNorm folded
    const float* bias;     // [7][128] folded conv bias
    const float* head_w;   // [128]  (value net: BatchNorm2d(1) folded in)
    const float* head_b;   // [81]
    const float* lin1_wt;  // value: [81][64] = lin1.weight^T with BatchNorm1d folded
    const float* lin1_b;   // value: [64]
    const float* lin2_w;   // value: [64]
    float lin2_b;
};

struct bk_eval_args {
    bk_net_params net[2];  // [0] policy, [1] value
    const void* feats;     // [B][27][9][9] f32 or u8 (device)
    int feats_dtype;
    int B_policy;          // PolicyNet runs on positions [0, B_policy)   (0: not at all)
    int B_value;           // ValueNet  runs on positions [0, B_value)
    int tasks_p, tasks_v;  // filled by the launcher: ceil(B_x / NB)
    float* logits;         // [B][81] or null
    float* probs;          // [B][81] or null
    float* values;         // [B] or null
    unsigned long long* stamps;  // diagnostic builds (-DBK_STAMPS) only: [block][wave][32] s_memtime
};

int bk_pick_nb(int B_policy, int B_value, int n_cu);
hipError_t bk_launch_leaf_eval(const bk_eval_args& a, int nb, hipStream_t stream);
