// bk_internal.h -- shared between the engine (host) and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// fragment-ordered weight buffer geometry (floats)
// layer 0: 4 waves x 25 taps x 4 groups x (64 lanes x 4)   layers 1..6: 4 x 9 x 16 x 256
#define BK_L0_WAVE_FLOATS (25 * 4 * 256)
#define BK_L0_FLOATS (4 * BK_L0_WAVE_FLOATS)
#define BK_L3_WAVE_FLOATS (9 * 16 * 256)
#define BK_L3_FLOATS (4 * BK_L3_WAVE_FLOATS)
#define BK_WFRAG_FLOATS (BK_L0_FLOATS + 6 * BK_L3_FLOATS)
#define BK_WFRAG_PAD_FLOATS 1024  // the B prefetch reads one 4 KiB block past the last slice

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
};

int bk_pick_nb(int B_policy, int B_value, int n_cu);
hipError_t bk_launch_leaf_eval(const bk_eval_args& a, int nb, hipStream_t stream);
