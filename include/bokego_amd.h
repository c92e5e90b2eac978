/*
 * bokego_amd.h -- C ABI of the MI355X leaf-evaluation engine (libbokego_amd.so).
 *
 * The reference (meiji163/bokego) has no FFI: its device boundary is the pair of Python
 * callables policy_net(x[B,27,9,9]) -> [B,81] and value_net(x) -> [B,1] invoked from
 *   bokego/nnet.py:265-275  policy_dist()  -> SOFT(policy(fts))      (one position per call)
 *   bokego/nnet.py:277-284  value()        -> v(fts).item()          (blocking)
 *   bokego/nnet.py:286-297  policy_sample()
 *   bokego/mcts.py:371-403  Go_MCTS.dist / .value  (the cache-miss path = "evaluate a leaf")
 * Each entry point below cites the reference interface it replaces.  All pointers are plain
 * host (or, where stated, device) pointers; no torch types cross this boundary.
 *
 * Conventions: every function returns BK_OK (0) or a negative bk_status; the message of the
 * last failure on an engine is available from bk_last_error().  One engine = one consumer thread
 * (the reference calls the nets from a single thread, gtp.py:98-108) and one private set of HIP
 * streams (copy-in, compute, copy-out, chained by events so that tickets overlap); several engines
 * per process/device are allowed.
 */
#ifndef BOKEGO_AMD_H
#define BOKEGO_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BK_ABI_VERSION 7

typedef enum bk_status {
    BK_OK = 0,
    BK_ERR_ARG = -1,     /* null/invalid argument                                      */
    BK_ERR_HIP = -2,     /* a HIP runtime call failed (message has the hipError string) */
    BK_ERR_OOM = -3,     /* host or device allocation failed                           */
    BK_ERR_BATCH = -4,   /* B > max_batch given at create                              */
    BK_ERR_NO_NET = -5,  /* output requested from a net the engine was created without */
    BK_ERR_NO_GPU = -6   /* no usable HIP device                                       */
} bk_status;

/* bits of `want` */
#define BK_WANT_LOGITS 1 /* PolicyNet.forward output, nnet.py:54-57            */
#define BK_WANT_PROBS 2  /* SOFT(logits), nnet.py:16,273                        */
#define BK_WANT_VALUE 4  /* ValueNet.forward output (after tanh), nnet.py:109-113 */

/* feature dtypes accepted by the *_device / submit entry points */
#define BK_FEATS_F32 0 /* float32 [B,27,9,9] NCHW: exactly nnet.features() output, nnet.py:182-262 */
#define BK_FEATS_U8 1  /* uint8   [B,27,9,9] NCHW: same planes (values 0..7) as bytes             */

/*
 * Unfolded weights under their reference state_dict names (nnet.py:31-53 / 73-101).
 * Block l (0..6) = conv.{3l} (Conv2d) + conv.{3l+1} (BatchNorm2d); head = conv.21
 * (Conv2dUntiedBias, nnet.py:138-180).  BatchNorm is folded inside bk_engine_create, in fp64.
 */
typedef struct bk_trunk_weights {
    const float *conv_w[7];  /* [0]: (128,27,5,5)  [1..6]: (128,128,3,3), OIHW */
    const float *conv_b[7];  /* (128)                                          */
    const float *bn_w[7];    /* BatchNorm2d weight (gamma) (128)               */
    const float *bn_b[7];    /* BatchNorm2d bias (beta)                        */
    const float *bn_mean[7]; /* running_mean                                   */
    const float *bn_var[7];  /* running_var, eps = 1e-5                        */
    const float *head_w;     /* conv.21.weight (1,128,1,1)                     */
    const float *head_b;     /* conv.21.bias   (1,9,9) untied                  */
} bk_trunk_weights;

typedef struct bk_value_head_weights { /* nnet.py:96-101 */
    const float *bn_w, *bn_b, *bn_mean, *bn_var;             /* BatchNorm2d(1): 1 element each */
    const float *lin1_w, *lin1_b;                            /* (64,81), (64)                  */
    const float *lin_bn_w, *lin_bn_b, *lin_bn_mean, *lin_bn_var; /* BatchNorm1d(64)             */
    const float *lin2_w, *lin2_b;                            /* (1,64), (1)                    */
} bk_value_head_weights;

typedef struct bk_policy_weights {
    bk_trunk_weights trunk;
} bk_policy_weights;

typedef struct bk_value_weights {
    bk_trunk_weights trunk;
    bk_value_head_weights head;
} bk_value_weights;

typedef struct bk_engine bk_engine;

typedef struct bk_stats_t {
    uint64_t evals;          /* positions evaluated                          */
    uint64_t batches;        /* kernel launches                              */
    uint64_t max_batch_seen;
    double kernel_ms_sum;    /* sum of HIP-event kernel durations (profiling on) */
    uint64_t kernel_ms_count;
    double last_kernel_ms;
    uint64_t f16_overflow_fallbacks; /* host-buffer requests redone in fp32 because an activation left the fp16 range */
    uint64_t f16_device_overflow;    /* bk_eval_device* calls redone in fp32 (gated launch on the caller's stream) for the same reason */
    uint64_t positions_encoded;      /* position records turned into feature planes on the GPU */
    uint64_t split_launches;         /* evaluations run as whole rounds of 3-board workgroups + a shorter tail launch */
    uint64_t coop_launches;          /* small fp32 batches run with 2..12 CUs per board (cooperative launch) */
    uint64_t coop_fallbacks;         /* ... of which redone by the one-CU form because a workgroup gave up waiting for its peers */
    /* ABI 6 (metrics row: "mean batch, queue wait") */
    double mean_batch;               /* evals / batches */
    double queue_wait_ms_sum;        /* profiling on, ticket path: per request, the time between the host handing it over and the GPU starting its first kernel */
    uint64_t queue_wait_count;       /* ... number of requests in that sum */
    double host_wait_ms_sum;         /* time the caller spent blocked inside bk_wait */
    uint64_t failed_submissions;     /* bk_submit* calls that failed after queueing work: drained, slot left clean (BK_ERR_HIP / BK_ERR_OOM returned) */
} bk_stats_t;

int bk_abi_version(void);
int bk_device_count(void);

/*
 * Replaces: PolicyNet()/ValueNet() construction + load_state_dict + .eval() + .to(device)
 * (boke.py:30-38, mcts.py:74-76).  Either net may be NULL (not both).  Weights are copied;
 * caller memory is not referenced after return.  max_batch bounds B of every later call.
 */
int bk_engine_create(const bk_policy_weights *policy, const bk_value_weights *value, int device_id,
                     int max_batch, bk_engine **out);
int bk_engine_destroy(bk_engine *e);

/*
 * New weights into a live engine (ABI 5).  Replaces: an optimizer step on a module that keeps being called -- the
 * reference's REINFORCE loop plays the next batch of games with the policy it has just updated
 * (bin/selfplay.py:80-84 self_play(pi, ...), 117-119 optimizer.step()) -- and load_state_dict on a constructed net
 * (boke.py:31-37).  A NULL net is left as it is; a net the engine was created without cannot be added.  Folding,
 * packing and the copies happen before the call returns (~10 ms; creating an engine is 40-60 ms: streams, pinned
 * slots); every later request sees the new weights.  No ticket may be outstanding (BK_ERR_ARG), and device-pointer
 * calls the caller enqueued on its own streams must have completed.
 */
int bk_engine_set_weights(bk_engine *e, const bk_policy_weights *policy, const bk_value_weights *value);

/*
 * Replaces: policy(fts) / v(fts) on host tensors (nnet.py:272-273, 283-284), batched.
 * feats: host float32 [B,27,9,9]; logits/probs: host [B,81]; values: host [B].
 * Output pointers for bits not in `want` may be NULL.  Synchronous.
 */
int bk_eval(bk_engine *e, const float *feats, int B, int want, float *logits, float *probs, float *values);
/* same with uint8 feature planes (4x less PCIe traffic) */
int bk_eval_u8(bk_engine *e, const uint8_t *feats, int B, int want, float *logits, float *probs, float *values);

/*
 * Device-resident variant (the reference's `--gpu` path keeps tensors on the device,
 * nnet.py:272 `.to(device)`): all pointers are device pointers on the engine's device.
 * `stream` is the hipStream_t to launch on (NULL = HIP's null stream, i.e. torch's default
 * current stream).  Asynchronous on that stream; the caller orders its own reads after it.
 */
int bk_eval_device(bk_engine *e, const void *d_feats, int feats_dtype, int B, int want, float *d_logits,
                   float *d_probs, float *d_values, void *stream);

/*
 * Asynchronous host-buffer variant for the batched leaf queue: returns a ticket (>0) or a
 * negative bk_status.  Host buffers must stay alive until bk_wait(ticket) returns.  At most
 * BK_MAX_INFLIGHT tickets may be outstanding; with two or three in flight the H2D copy of one, the
 * kernel of another and the D2H copy of a third run concurrently.
 * How a request travels (results never depend on it; every variant is bit-identical):
 *   - at most 256 positions, fp32 engine: no copies at all -- the feature encoder (bk_submit_positions) reads the records from
 *     the pinned slot, the leaf kernel writes its flag word and outputs into the pinned output block (option no_direct: copies);
 *     up to 128 network tasks run as the cooperative launch (2..12 CUs per board);
 *   - larger requests: H2D, kernels and D2H on three event-chained streams; host planes of >= 4 MiB are staged by a small
 *     pool of copy threads whose slices are sent as they land, and from 16 MiB the first 768 positions are launched as soon as
 *     THEIR planes have arrived, running while the rest is copied (options no_head_part = 1, copy_threads = 0: off).
 */
#define BK_MAX_INFLIGHT 4
int64_t bk_submit(bk_engine *e, const void *feats, int feats_dtype, int B, int want, float *logits, float *probs,
                  float *values);
int bk_wait(bk_engine *e, int64_t ticket);

/*
 * MCTS expansion batches (mcts.py:185-192 expands one node: ONE policy evaluation, mcts.py:371-383,
 * but a value for every new child, mcts.py:393-403): PolicyNet outputs are produced only for the
 * first n_policy positions (logits/probs buffers are [n_policy,81]), the value for all B.
 */
int64_t bk_submit_prefix(bk_engine *e, const void *feats, int feats_dtype, int B, int n_policy, int want,
                         float *logits, float *probs, float *values);
int bk_eval_device_prefix(bk_engine *e, const void *d_feats, int feats_dtype, int B, int n_policy, int want,
                          float *d_logits, float *d_probs, float *d_values, void *stream);

/*
 * Position records in, instead of feature planes ("optional HIP encoder writing uint8 planes
 * straight into the engine's input buffer").  `positions` is an array of B 192-byte bk_pos records
 * (include/bokego_go.h) whose liberty cache has been refreshed by the host (bk_pos_liberties /
 * bk_pool_collect_pos -- the history-dependent part of nnet.features(), nnet.py:213-215, go.py:220-243,
 * stays on the host); the GPU computes the 27 planes of nnet.py:216-262 from each record and evaluates
 * them.  Results are bit-identical to bk_submit_prefix(bk_pos_features_u8(...)); 192 B instead of 2,187 B
 * cross PCIe per leaf and the host spends no time encoding.
 */
#define BK_POS_RECORD_BYTES 192
int64_t bk_submit_positions(bk_engine *e, const void *positions, int B, int n_policy, int want, float *logits,
                            float *probs, float *values);
/* the encoder alone: planes[B][27][9][9] uint8 on the host (tests, diagnostics); synchronous */
int bk_encode_positions(bk_engine *e, const void *positions, int B, uint8_t *planes);

/*
 * Arithmetic of the conv stacks (new; the reference computes in torch fp32, nnet.py:31-57,73-113):
 *   BK_PRECISION_FP32   (default) v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulation (bit-identical to
 *                       an fmaf chain) -- the reference's arithmetic width
 *   BK_PRECISION_F16X2  opt-in: every operand split into an fp16 hi/lo pair (22 significant bits), three
 *                       v_mfma_f32_32x32x16_f16 per K step, fp32 accumulation; measured as close to a float64
 *                       evaluation of the reference as the fp32 kernel is, ~4x the throughput.
 *                       If an activation leaves the fp16 range (|x| >= 4094) the kernel raises a flag and the
 *                       request is redone on the fp32 kernel: by bk_wait() for host-buffer requests
 *                       (bk_stats().f16_overflow_fallbacks), and for bk_eval_device* by an fp32 launch enqueued
 *                       behind the f16x2 one on the caller's stream that is a no-op unless the flag was raised
 *                       (bk_stats().f16_device_overflow counts the calls redone).  Either way the caller never
 *                       sees a clamped result.
 * The environment variable BK_PRECISION=f32|f16x2 sets the default of new engines.
 */
#define BK_PRECISION_FP32 0
#define BK_PRECISION_F16X2 1
int bk_engine_set_precision(bk_engine *e, int precision);
int bk_engine_get_precision(bk_engine *e);

int bk_engine_set_profiling(bk_engine *e, int on); /* HIP-event timing of every kernel launch */

/*
 * Diagnostic switches of one engine (ABI 7).  Their defaults come from the environment, which is read ONCE, inside
 * bk_engine_create (BK_PRECISION, BK_FORCE_NB, BK_NO_SPLIT, BK_COOP, BK_COOP3, BK_NO_DIRECT, BK_NO_HEAD_PART, BK_COPY_THREADS,
 * BK_ENCODE_OVERLAP, BK_NO_FUSE_ENCODE, BK_ROCTX); no request ever looks at the environment, and a live engine is changed through this call
 * only.  Results never depend on any of them (every launch form is bit-identical); they exist so that tests and probes can
 * force each form.  Names: "force_nb" (0 | 1..3 boards per workgroup), "no_split", "coop" (-1 by task count | 0 off | 2, 3,
 * 4, 6, 8, 12 CUs per board), "coop3" (-1 | 0 | 2 | 4 | 8), "no_direct", "no_head_part", "copy_threads", "encode_overlap",
 * "no_fuse_encode" (1: small requests of position records run the encoder kernel in front of the leaf kernel again instead of
 * letting the leaf kernel compute the planes from the records while it stages them; BK_NO_FUSE_ENCODE at create), "direct_rows"
 * (requests of position records up to this many rows -- default 1024 -- take the copy-free one-kernel path; BK_DIRECT_ROWS).
 * Unknown name or a value the switch does not know (coop = 5, copy_threads < 0): BK_ERR_ARG, nothing changed.  Like every call
 * on an engine handle it belongs to the one thread that submits (the handle is single-consumer); the planner's switches
 * ("force_nb", "no_split", "coop", "coop3") are refused while tickets are in flight -- a request's redo after a failed cooperative
 * launch must see the plan its submission saw.  Builds with -DBK_TEST_HOOKS (bk_has_test_hooks() == 1; never shipped as libbokego_amd.so) add
 * "coop_fault" and "fault_submit" (fault injection for tests/test_gpu_hooks.py).
 */
int bk_engine_set_option(bk_engine *e, const char *name, int value);
int bk_engine_get_option(bk_engine *e, const char *name, int *value);
int bk_has_test_hooks(void);

/*
 * The engine as the evaluator of the native step loop (include/bokego_tree.h: bk_pools_run drives lock-step game pools
 * through two callbacks): fills *out with { ctx = e, submit = bk_submit_positions(PROBS for the policy rows | VALUE), wait =
 * bk_wait }.  Replaces: the per-position calls of Go_MCTS.dist / .value (mcts.py:371-403) for every tree of a self-play
 * worker (bin/selfplay.py:177-199), with no interpreter between two batches.  `out` must not outlive the engine.
 */
struct bk_evaluator;
int bk_engine_evaluator(bk_engine *e, struct bk_evaluator *out);
int bk_stats(bk_engine *e, bk_stats_t *out);
int bk_engine_max_batch(bk_engine *e);

/*
 * Launch planner, as a pure function (no engine, no GPU): how a request of n_policy PolicyNet rows + n_value ValueNet
 * rows would be launched on a device with n_cu compute units at `precision`.  Returns the CUs per board of the
 * cooperative small-batch form (12/8/6/4/3/2; ticket path, fp32 only), 108 / 104 / 102 for groups of three boards shared by 8 / 4 /
 * 2 CUs (requests of 81..96 tasks in at most 32 groups / 129..192 / 257..384 tasks), or 0 when the ordinary form runs;
 * *boards_per_workgroup (may be NULL) receives the ordinary form's workgroup size (1..3) for a single launch.
 */
int bk_plan_query(int n_policy, int n_value, int n_cu, int precision, int *boards_per_workgroup);
/*
 * What the fp32 kernel's matrix unit executes for such a request, counted from the tile tables the kernel is compiled
 * from (bk_kernels.hip, Tiles<NB>: 16-position tiles x taps that are not skipped x k-steps x cout tiles x 2,048 FLOP per
 * v_mfma_f32_16x16x4_f32), over the launch plan the engine would use (a single launch, k rounds of 3-board workgroups +
 * a tail, or -- cooperative != 0 and a small request -- the cooperative form), and the algorithmic work beside it
 * (valid taps only: 2 x 66,706,944 FLOP per PolicyNet row, 2 x 66,712,192 per ValueNet row).  The quotient
 * is what separates MFMA-pipe occupancy from the roofline fraction bench.py reports: padding rows (243 -> 256), edge taps
 * inside mixed tiles, the 27 -> 28 channel pad of layer 0.  Pure function, no GPU.  Any out pointer may be NULL.
 */
int bk_plan_flops(int n_policy, int n_value, int n_cu, int cooperative, double *executed_mfma_flop,
                  double *algorithmic_flop, int *n_launches);
int bk_engine_synchronize(bk_engine *e);
const char *bk_last_error(bk_engine *e); /* e == NULL: last error of a failed create */

#ifdef __cplusplus
}
#endif
#endif /* BOKEGO_AMD_H */
