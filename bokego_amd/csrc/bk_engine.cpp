// bk_engine.cpp -- host side of libbokego_amd.so: weight folding/packing, buffers, streams, C ABI.
//
// Reference call sites replaced (see include/bokego_amd.h for the per-function citations):
//   boke.py:30-38 (construct + load_state_dict + eval), mcts.py:74-76 (.to(device)),
//   nnet.py:265-297 (policy_dist / value / policy_sample forward calls).
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bokego_amd.h"
#include "../../include/bokego_tree.h"   // bk_evaluator (bk_engine_evaluator), bk_pos
#include "bk_internal.h"

namespace {

thread_local std::string g_create_error;

// The ONE place this file reads the environment, called from bk_engine_create only (and from the marker loader it triggers):
// every BK_* switch becomes a field of the engine there.  A request never looks at the environment -- getenv is not safe
// against a concurrent setenv (Python's os.environ[...] = in another thread), and a stray variable in an operator's shell
// must not change a running engine.  bk_engine_set_option changes a switch of a live engine.
const char* env_str(const char* name) { return getenv(name); }
int env_int(const char* name, int dflt) {
    const char* v = env_str(name);
    return v && *v ? atoi(v) : dflt;
}

// roctx ranges around the requests, so that a rocprofv3 --marker-trace timeline shows "bk_submit B=.. np=.." / "bk_wait"
// above the kernels and copies they enqueue ("roctx ranges around bk_eval").  The marker library is looked up at
// run time (no link dependency; an engine on a box without it simply has no ranges) when BK_ROCTX is set -- tools that
// profile set it; an unprofiled run does not pay for the calls.
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        if (!env_str("BK_ROCTX")) return;
        for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
                push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
                pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (push && pop) return;
                push = nullptr;
                pop = nullptr;
            }
        }
    }
};
const Roctx& roctx() {
    static const Roctx r;
    return r;
}
struct RoctxRange {   // RAII: every return path of the C-ABI functions closes its range
    bool on;
    explicit RoctxRange(const char* what, int B = -1, int np = -1) : on(roctx().push != nullptr) {
        if (!on) return;
        char buf[96];
        if (B >= 0) snprintf(buf, sizeof buf, "%s B=%d n_policy=%d", what, B, np);
        else snprintf(buf, sizeof buf, "%s", what);
        roctx().push(buf);
    }
    ~RoctxRange() {
        if (on) roctx().pop();
    }
};

#define BK_STAMP_BLOCKS 16384
constexpr double kBnEps = 1e-5;  // torch.nn.BatchNorm default, nnet.py:33,98-99

// Staging of a large host buffer (the reference-shaped call: f32 planes, 35.8 MB at B = 4,096): a single memcpy into the
// pinned slot runs at ~10 GB/s and the H2D copy only starts behind it -- 3.5 + 0.7 ms in front of an 8 ms kernel.  Here a
// few worker threads copy the buffer slice by slice while the submitting thread enqueues each slice's H2D copy as soon as
// that slice has landed, so staging costs about as long as the slower of the two (both ~0.7 ms).  Option copy_threads = 0 (BK_COPY_THREADS at create) turns it
// off.  Workers sleep on a condition variable between requests.
class CopyPool {
public:
    static constexpr int kMaxSlices = 64;
    explicit CopyPool(int n) {
        for (int i = 0; i < n; ++i) th_.emplace_back([this] { run(); });
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> g(m_);
            quit_ = true;
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    // copy src -> dst in `n` slices of `slice` bytes (the last one shorter); ready(i) is called on the calling thread, in
    // order, once slice i is in place
    template <typename Ready>
    bool copy(void* dst, const void* src, size_t bytes, size_t slice, Ready ready) {
        const int n = (int)((bytes + slice - 1) / slice);
        if (n > kMaxSlices) return false;
        for (int i = 0; i < n; ++i) done_[i].store(0, std::memory_order_relaxed);
        {
            // a worker may still be leaving the previous job's loop (every slice of that job is in place, but the worker has
            // not yet seen "no slice left"): the job fields change only while nobody reads them
            std::unique_lock<std::mutex> g(m_);
            idle_cv_.wait(g, [&] { return active_ == 0; });
            dst_ = static_cast<char*>(dst);
            src_ = static_cast<const char*>(src);
            bytes_ = bytes;
            slice_ = slice;
            n_ = n;
            next_.store(0, std::memory_order_relaxed);
            ++gen_;
        }
        cv_.notify_all();
        bool ok = true;
        for (int i = 0; i < n; ++i) {
            while (!done_[i].load(std::memory_order_acquire)) std::this_thread::yield();
            ok = ready(i) && ok;
        }
        return ok;
    }

private:
    void run() {
        unsigned long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return quit_ || gen_ != seen; });
                if (quit_) return;
                seen = gen_;
                ++active_;
            }
            for (;;) {
                const int i = next_.fetch_add(1, std::memory_order_relaxed);
                if (i >= n_) break;
                const size_t off = (size_t)i * slice_;
                std::memcpy(dst_ + off, src_ + off, std::min(slice_, bytes_ - off));
                done_[i].store(1, std::memory_order_release);
            }
            {
                std::lock_guard<std::mutex> g(m_);
                --active_;
            }
            idle_cv_.notify_one();
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, idle_cv_;
    int active_ = 0;      // workers inside a job's loop
    bool quit_ = false;
    unsigned long gen_ = 0;
    char* dst_ = nullptr;
    const char* src_ = nullptr;
    size_t bytes_ = 0, slice_ = 0;
    int n_ = 0;
    std::atomic<int> next_{0};
    std::atomic<int> done_[kMaxSlices];
};

struct Slot {  // one in-flight host-buffer request
    int64_t ticket = 0;
    int B = 0, n_policy = 0, want = 0;
    float *logits = nullptr, *probs = nullptr, *values = nullptr;  // caller's host buffers
    void* h_in = nullptr;                                          // pinned staging
    // outputs of a request live in ONE device block and ONE pinned block, laid out per request as
    // [flag (64 B) | values B | probs n_policy*81 | logits n_policy*81], so a single D2H brings everything back
    char* h_out = nullptr;
    char* d_out = nullptr;
    size_t off_values = 0, off_probs = 0, off_logits = 0, out_bytes = 0;
    bool flag_dirty = false;  // the device flag was raised: reset it before the slot's next f16x2 launch
    void* d_in = nullptr;
    void* d_pos = nullptr;  // position records (bk_submit_positions): encoded into d_in on the GPU
    // the pinned blocks as the GPU addresses them: small fp32 requests skip both copies -- the encoder reads the position
    // records straight from h_in and the leaf kernel writes flag + outputs straight into h_out (submit_common, "direct")
    void* h_in_dev = nullptr;
    char* h_out_dev = nullptr;
    bool direct = false;    // this request's outputs were written to h_out by the kernel (no D2H copy was enqueued)
    int dtype = 0;
    const void* d_feats = nullptr;   // what the request's leaf kernel read (d_in, or the records in h_in_dev: fused encoding): its redo reads the same
    hipEvent_t in_ready = nullptr;   // H2D of this request finished (copy-in stream)
    hipEvent_t head_ready = nullptr; // ... of its first part (large requests are launched in two parts)
    hipEvent_t computed = nullptr;   // kernel of this request finished (compute stream)
    hipEvent_t done = nullptr;       // D2H of this request finished (copy-out stream)
    bool busy = false;
};

}  // namespace

struct bk_engine {
    int device = 0;
    int max_batch = 0;
    int n_cu = 256;
    bool has_policy = false, has_value = false;
    int precision = BK_PRECISION_F32;
    // ticket path: H2D, kernels and D2H run on three streams chained by per-slot events, so the copies of
    // one request overlap the kernel of another (MI355X has separate SDMA engines per direction)
    hipStream_t stream = nullptr;       // compute
    hipStream_t s_in = nullptr, s_out = nullptr;
    CopyPool* copy_pool = nullptr;      // created by the first large host-buffer request
    std::vector<void*> dev_allocs;
    bk_net_params net[2]{};
    Slot slots[BK_MAX_INFLIGHT];
    int64_t next_ticket = 1;
    std::string err;
    // stats / profiling.  bk_stats may be called from another thread than the one that submits (a monitor polling it): the
    // counters are bumped / read with relaxed atomics, the timing-event ring is guarded by ev_m, and bk_stats makes no call
    // that waits for the device (events are polled with hipEventQuery; the one device-side counter is fetched by a 4-byte
    // copy on a stream of its own)
    bk_stats_t st{};
    bool profiling = false;
    std::mutex ev_m;
    struct TimedLaunch { hipEvent_t start, stop; double host_ms; bool ticket; };
    std::vector<TimedLaunch> ev_ring;
    size_t ev_head = 0, ev_pending = 0;
    hipEvent_t epoch = nullptr;          // recorded at create on an idle device: GPU-side times are measured from it ...
    std::chrono::steady_clock::time_point epoch_host;   // ... and host-side times from this instant (queue_wait_ms_sum)
    // device-pointer path, f16x2: which calls were redone in fp32.  The gated redo kernel of call number N stores N into word
    // N % BK_DEV_FLAGS of this PINNED block (a plain system-scope store by one thread; no atomics on host memory), the host
    // counts the words that changed -- when bk_stats is asked, and before a word is reused 256 calls later -- without any
    // call into the device.  (A copy from device memory, even on a stream of its own, queues on the copy engine behind the
    // tickets' copies, which wait for their kernels: 32 ms for a counter.)
    unsigned int* h_redo = nullptr;
    unsigned int* h_redo_dev = nullptr;  // the same block as the GPU addresses it
    unsigned int redo_seen[BK_DEV_FLAGS] = {};
    // switches (bk_engine_set_option; defaults from the environment, read once at create)
    bk_plan_opts plan;                   // force_nb, no_split, coop, coop3
    int no_direct = 0, no_head_part = 0, encode_overlap = 0, copy_threads = 6, no_fuse_encode = 0;
    // requests of position records up to this many rows take the copy-free one-kernel path (option "direct_rows"): 256 until round 6; a
    // 512-game generation's steps (~340 rows) without their H2D copy, encoder kernel and two event hops: 0.822 -> 0.789 s
    // (tools/direct_rows_probe.py; 512 / 1024 / 4096 measure the same); larger requests keep the three-stream chain, which overlaps
    // their copies with other tickets' kernels
    int direct_rows = 1024;
#ifdef BK_TEST_HOOKS
    // test builds only (make hooks): the n-th HIP call of every ticket submission (fault_submit), or of anything from now on
    // (bk_debug_fail_nth_hip_call), reports a failure instead of being made; coop_fault: a cooperative peer deserts
    int fault_at = 0, fault_seen = 0, fault_submit = 0, coop_fault = 0;
#endif
    // device-pointer path, f16x2: a ring of BK_DEV_FLAGS words, one per call (call number % BK_DEV_FLAGS), zeroed in stream
    // order in front of the call's f16x2 kernel, which raises it to the call number on overflow: that is what gates the
    // call's fp32 redo kernel.  One word per call, so calls running concurrently on different caller streams cannot hide
    // each other's overflow (with ONE shared word, atomicMax(flag, N) was a no-op once call N+1 had raised it: ADVICE r2).
    unsigned int* d_dev_flag = nullptr;
    unsigned int dev_seq = 0;
    // cooperative small-batch launches (ticket path on the compute stream only: one exchange buffer): the exchange buffer
    // and the per-task arrival counters, BK_COOP_SYNC_STRIDE words apart
    float* d_coop_xchg = nullptr;
    unsigned int* d_coop_sync = nullptr;
    unsigned long long* d_stamps = nullptr;  // diagnostic builds only
};

namespace {

int fail(bk_engine* e, int code, const std::string& msg) {
    if (e) e->err = msg; else g_create_error = msg;
    return code;
}
#ifdef BK_TEST_HOOKS
// test hook (option "fault_submit", armed by submit_common for the length of one submission): true = this HIP call "fails"
inline bool inject_fault(bk_engine* e) {
    if (!e || e->fault_at <= 0 || ++e->fault_seen != e->fault_at) return false;
    e->fault_at = 0;                                     // one shot
    return true;
}
#else
inline bool inject_fault(bk_engine*) { return false; }
#endif
inline void bump(uint64_t& c, uint64_t v = 1) { __atomic_fetch_add(&c, v, __ATOMIC_RELAXED); }
#define HIP_TRY(e, call)                                                                       \
    do {                                                                                       \
        hipError_t _s = inject_fault(e) ? hipErrorUnknown : (call);                            \
        if (_s != hipSuccess)                                                                  \
            return fail(e, _s == hipErrorOutOfMemory ? BK_ERR_OOM : BK_ERR_HIP,                \
                        std::string(#call) + ": " + hipGetErrorString(_s));                    \
    } while (0)

bool trunk_ok(const bk_trunk_weights& t) {
    for (int l = 0; l < 7; ++l)
        if (!t.conv_w[l] || !t.conv_b[l] || !t.bn_w[l] || !t.bn_b[l] || !t.bn_mean[l] || !t.bn_var[l]) return false;
    return t.head_w && t.head_b;
}

// Fold BatchNorm2d (eval) into conv l and emit the MFMA fragment order consumed by conv_layer<> in bk_kernels.hip
// (v_mfma_f32_16x16x4_f32, weights = A operand):
//   index = (((tap*G + g)*8 + ctile)*64 + lane)*4 + j,   G = 2 (layer 0) / 8 groups of 16 input slots
//   MFMA row r = lane & 15 of cout tile ctile holds output SLOT 16*ctile + r; lane quad kq = lane >> 4 and k-step j
//   consume input SLOT 16g + 4kq + j.  Slot -> real channel: bk_slot_perm for the outputs of layers 0..5 (so that the
//   k order of every dot product equals the round-1 kernel's: bit-identical results), identity for layer 6
//   (the heads read natural channel order) and, for layer 0's input, the first 24 planes by the same permutation and
//   planes 24..26 in k-step 2 of the second group (7 k-steps of 4 = 28 channels).
void pack_trunk(const bk_trunk_weights& t, std::vector<float>& wfrag, std::vector<float>& bias) {
    wfrag.assign(BK_WFRAG_FLOATS + BK_WFRAG_PAD_FLOATS, 0.f);
    bias.assign(7 * 128, 0.f);
    size_t base = 0;
    for (int l = 0; l < 7; ++l) {
        const int K = l == 0 ? 5 : 3, cin = l == 0 ? 27 : 128, G = l == 0 ? 2 : 8, TAPS = K * K;
        std::vector<double> scale(128);
        for (int co = 0; co < 128; ++co) scale[co] = (double)t.bn_w[l][co] / std::sqrt((double)t.bn_var[l][co] + kBnEps);
        auto out_ch = [&](int slot) { return l == 6 ? slot : bk_slot_perm(slot); };
        for (int sl = 0; sl < 128; ++sl) {
            const int co = out_ch(sl);
            bias[l * 128 + sl] = (float)(((double)t.conv_b[l][co] - (double)t.bn_mean[l][co]) * scale[co] + (double)t.bn_b[l][co]);
        }
        for (int tp = 0; tp < TAPS; ++tp)
            for (int g = 0; g < G; ++g)
                for (int ct = 0; ct < 8; ++ct)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 4; ++j) {
                            const int co = out_ch(16 * ct + (lane & 15)), kq = lane >> 4;
                            int ci = bk_slot_perm(16 * g + 4 * kq + j);
                            if (l == 0 && g == 1) ci = j < 2 ? ci : (j == 2 && kq < 3 ? 24 + kq : cin);
                            float v = 0.f;
                            if (ci < cin) v = (float)((double)t.conv_w[l][((size_t)co * cin + ci) * TAPS + tp] * scale[co]);
                            wfrag[base + ((((size_t)tp * G + g) * 8 + ct) * 64 + lane) * 4 + j] = v;
                        }
        base += (size_t)TAPS * G * 2048;
    }
}

// f16x2 path: folded weights scaled by 2^e_l and split into fp16 hi/lo, in the fragment order of
// conv_layer16<> (bk_kernels_f16.hip):
//   index = (((ks*4 + ntile)*2 + piece)*64 + lane)*8 + j,  ks = tap*S + s
//   cout = 32*ntile + (lane&31), cin = 16s + 8(lane>>5) + j
constexpr float kSa16 = 16.f;  // activation scale (power of two): |activation| < 65504/16 representable
void pack_trunk16(const bk_trunk_weights& t, std::vector<_Float16>& wfrag, std::vector<float>& bias16, float* cscale) {
    wfrag.assign(BK16_WFRAG_HALFS + BK16_WFRAG_PAD_HALFS, (_Float16)0.f);
    bias16.assign(7 * 128, 0.f);
    size_t base = 0;
    for (int l = 0; l < 7; ++l) {
        const int K = l == 0 ? 5 : 3, cin = l == 0 ? 27 : 128, S = l == 0 ? 2 : 8, TAPS = K * K;
        std::vector<double> scale(128);
        double wmax = 0;
        for (int co = 0; co < 128; ++co) {
            scale[co] = (double)t.bn_w[l][co] / std::sqrt((double)t.bn_var[l][co] + kBnEps);
            bias16[l * 128 + co] = (float)(kSa16 * (((double)t.conv_b[l][co] - (double)t.bn_mean[l][co]) * scale[co] + (double)t.bn_b[l][co]));
            for (int i = 0; i < cin * TAPS; ++i) wmax = std::max(wmax, std::fabs((double)t.conv_w[l][(size_t)co * cin * TAPS + i] * scale[co]));
        }
        int el = wmax > 0 ? (int)std::floor(std::log2(32768.0 / wmax)) : 0;  // max |w| * 2^el <= 2^15
        el = std::max(-14, std::min(el, 24));
        const double sw = std::ldexp(1.0, el);
        const double sa_in = l == 0 ? 1.0 : kSa16;
        cscale[l] = (float)(kSa16 / (sa_in * sw));
        for (int tp = 0; tp < TAPS; ++tp)
            for (int sidx = 0; sidx < S; ++sidx)
                for (int nt = 0; nt < 4; ++nt)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int co = 32 * nt + (lane & 31), ci = 16 * sidx + 8 * (lane >> 5) + j;
                            double w = 0;
                            if (ci < cin) w = (double)t.conv_w[l][((size_t)co * cin + ci) * TAPS + tp] * scale[co] * sw;
                            const _Float16 hi = (_Float16)w;
                            const _Float16 lo = (_Float16)(w - (double)hi);
                            const size_t ks = (size_t)tp * S + sidx;
                            const size_t at = base + (((ks * 4 + nt) * 2) * 64 + lane) * 8 + j;
                            wfrag[at] = hi;
                            wfrag[at + 512] = lo;
                        }
        base += l == 0 ? BK16_L0_HALFS : BK16_L3_HALFS;
    }
}

// a fresh device buffer holding h; recorded in `fresh` (the caller owns it until the whole set of weights is in place)
template <typename T>
int upload(bk_engine* e, const std::vector<T>& h, const T** out, std::vector<void*>& fresh) {
    void* d = nullptr;
    HIP_TRY(e, hipMalloc(&d, h.size() * sizeof(T)));
    fresh.push_back(d);
    HIP_TRY(e, hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = static_cast<const T*>(d);
    return BK_OK;
}

int setup_trunk(bk_engine* e, const bk_trunk_weights& t, bk_net_params& np, double head_scale, double head_shift, std::vector<void*>& fresh) {
    std::vector<float> wfrag, bias, hw(128), hb(81);
    pack_trunk(t, wfrag, bias);
    // Conv2dUntiedBias (nnet.py:175-180); for the value net BatchNorm2d(1) is folded in:
    //   bn(h) = (h - mean) * s + beta  with  s = gamma / sqrt(var + eps)
    for (int c = 0; c < 128; ++c) hw[c] = (float)((double)t.head_w[c] * head_scale);
    for (int q = 0; q < 81; ++q) hb[q] = (float)((double)t.head_b[q] * head_scale + head_shift);
    std::vector<_Float16> w16;
    std::vector<float> b16;
    pack_trunk16(t, w16, b16, np.cscale16);
    np.inv_sa16 = 1.f / kSa16;
    int rc;
    if ((rc = upload(e, wfrag, &np.wfrag, fresh))) return rc;
    if ((rc = upload(e, bias, &np.bias, fresh))) return rc;
    if ((rc = upload(e, hw, &np.head_w, fresh))) return rc;
    if ((rc = upload(e, hb, &np.head_b, fresh))) return rc;
    if ((rc = upload(e, w16, &np.wfrag16, fresh))) return rc;
    if ((rc = upload(e, b16, &np.bias16, fresh))) return rc;
    return BK_OK;
}

bool weights_ok(const bk_policy_weights* policy, const bk_value_weights* value) {
    if (policy && !trunk_ok(policy->trunk)) return false;
    if (value) {
        const bk_value_head_weights& h = value->head;
        if (!trunk_ok(value->trunk) || !h.bn_w || !h.bn_b || !h.bn_mean || !h.bn_var || !h.lin1_w || !h.lin1_b ||
            !h.lin_bn_w || !h.lin_bn_b || !h.lin_bn_mean || !h.lin_bn_var || !h.lin2_w || !h.lin2_b)
            return false;
    }
    return true;
}

// every device buffer a net's parameter block points at
std::vector<const void*> net_buffers(const bk_net_params& np) {
    return {np.wfrag, np.bias, np.head_w, np.head_b, np.lin1_wt, np.lin1_b, np.lin2_w, np.wfrag16, np.bias16};
}

// Fold BatchNorm, pack and upload the weights of the nets given (nullptr: that net is left alone).  All or nothing: the new
// weights go into FRESH device buffers, and only when every one of them is in place do the engine's parameter blocks switch
// over (the old buffers are freed); on any failure the fresh buffers are freed and the engine is exactly as before -- never
// a mixture of old and new weights.  The caller has made sure nothing on the device still reads the old buffers.
int load_weights(bk_engine* e, const bk_policy_weights* policy, const bk_value_weights* value) {
    bk_net_params np[2] = {e->net[0], e->net[1]};
    std::vector<void*> fresh;
    auto build = [&]() -> int {
        int rc;
        if (policy) {
            np[0] = bk_net_params{};
            if ((rc = setup_trunk(e, policy->trunk, np[0], 1.0, 0.0, fresh))) return rc;
        }
        if (value) {
            np[1] = bk_net_params{};
            const bk_value_head_weights& h = value->head;
            const double s = (double)h.bn_w[0] / std::sqrt((double)h.bn_var[0] + kBnEps);
            const double shift = (double)h.bn_b[0] - (double)h.bn_mean[0] * s;
            if ((rc = setup_trunk(e, value->trunk, np[1], s, shift, fresh))) return rc;
            // lin1 (64,81) + BatchNorm1d(64) folded, stored transposed [81][64] for coalesced reads
            std::vector<float> w1t(81 * 64), b1(64), w2(64);
            for (int j = 0; j < 64; ++j) {
                const double sj = (double)h.lin_bn_w[j] / std::sqrt((double)h.lin_bn_var[j] + kBnEps);
                for (int q = 0; q < 81; ++q) w1t[q * 64 + j] = (float)((double)h.lin1_w[j * 81 + q] * sj);
                b1[j] = (float)(((double)h.lin1_b[j] - (double)h.lin_bn_mean[j]) * sj + (double)h.lin_bn_b[j]);
                w2[j] = h.lin2_w[j];
            }
            if ((rc = upload(e, w1t, &np[1].lin1_wt, fresh))) return rc;
            if ((rc = upload(e, b1, &np[1].lin1_b, fresh))) return rc;
            if ((rc = upload(e, w2, &np[1].lin2_w, fresh))) return rc;
            np[1].lin2_b = h.lin2_b[0];
        }
        return BK_OK;
    };
    if (const int rc = build()) {
        for (void* d : fresh) (void)hipFree(d);
        return rc;
    }
    for (int i = 0; i < 2; ++i) {
        if (!(i == 0 ? (const void*)policy : (const void*)value)) continue;
        for (const void* old : net_buffers(e->net[i])) {
            if (!old) continue;
            auto it = std::find(e->dev_allocs.begin(), e->dev_allocs.end(), const_cast<void*>(old));
            if (it != e->dev_allocs.end()) e->dev_allocs.erase(it);
            (void)hipFree(const_cast<void*>(old));
        }
        e->net[i] = np[i];
    }
    e->dev_allocs.insert(e->dev_allocs.end(), fresh.begin(), fresh.end());
    return BK_OK;
}

int alloc_slot(bk_engine* e, Slot& s) {
    const size_t B = (size_t)e->max_batch;
    HIP_TRY(e, hipHostMalloc(&s.h_in, B * 2187 * sizeof(float), hipHostMallocDefault));
    HIP_TRY(e, hipMalloc(&s.d_in, B * 2187 * sizeof(float)));
    HIP_TRY(e, hipMalloc(&s.d_pos, B * BK_POS_BYTES));
    const size_t out_max = 64 + B * sizeof(float) + 2 * B * 81 * sizeof(float) + 3 * 64;  // + section padding
    HIP_TRY(e, hipHostMalloc((void**)&s.h_out, out_max, hipHostMallocDefault));
    HIP_TRY(e, hipMalloc((void**)&s.d_out, out_max));
    HIP_TRY(e, hipMemset(s.d_out, 0, 64));
    std::memset(s.h_out, 0, 64);
    // device-side addresses of the two pinned blocks (mapped into the device's address space by hipHostMalloc); a
    // failure here only disables the direct path
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, s.h_in, 0) == hipSuccess) s.h_in_dev = dp;
    if (hipHostGetDevicePointer(&dp, s.h_out, 0) == hipSuccess) s.h_out_dev = static_cast<char*>(dp);
    (void)hipGetLastError();
    HIP_TRY(e, hipEventCreateWithFlags(&s.in_ready, hipEventDisableTiming));
    HIP_TRY(e, hipEventCreateWithFlags(&s.head_ready, hipEventDisableTiming));
    HIP_TRY(e, hipEventCreateWithFlags(&s.computed, hipEventDisableTiming));
    HIP_TRY(e, hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    return BK_OK;
}

void free_slot(Slot& s) {
    if (s.h_in) (void)hipHostFree(s.h_in);
    if (s.h_out) (void)hipHostFree(s.h_out);
    if (s.d_in) (void)hipFree(s.d_in);
    if (s.d_pos) (void)hipFree(s.d_pos);
    if (s.d_out) (void)hipFree(s.d_out);
    if (s.in_ready) (void)hipEventDestroy(s.in_ready);
    if (s.head_ready) (void)hipEventDestroy(s.head_ready);
    if (s.computed) (void)hipEventDestroy(s.computed);
    if (s.done) (void)hipEventDestroy(s.done);
    s = Slot{};
}

int check_want(bk_engine* e, int B, int n_policy, int want) {
    if (!e) return BK_ERR_ARG;
    if (B < 0 || (want & ~7) || want == 0) return fail(e, BK_ERR_ARG, "bad B or want mask");
    if (n_policy < 0 || n_policy > B) return fail(e, BK_ERR_ARG, "n_policy must be in [0, B]");
    if (B > e->max_batch) return fail(e, BK_ERR_BATCH, "B exceeds max_batch given at bk_engine_create");
    if ((want & (BK_WANT_LOGITS | BK_WANT_PROBS)) && !e->has_policy)
        return fail(e, BK_ERR_NO_NET, "policy outputs requested but the engine has no PolicyNet");
    if ((want & BK_WANT_VALUE) && !e->has_value)
        return fail(e, BK_ERR_NO_NET, "value requested but the engine has no ValueNet");
    return BK_OK;
}

// a redo word that changed since it was last looked at = one more device-path call redone in fp32 (called by the submitting
// thread, or by bk_stats; a late store by a call still in flight when its word is reused can be missed: a statistic)
void fold_redo(bk_engine* e, unsigned int slot) {
    const unsigned int v = __atomic_load_n(e->h_redo + slot, __ATOMIC_RELAXED);
    if (v != e->redo_seen[slot]) {
        e->redo_seen[slot] = v;
        bump(e->st.f16_device_overflow);
    }
}

// fold the (start, stop) pairs of launches that have FINISHED into the stats, oldest first; stops at the first one still in
// flight (hipEventQuery: never waits).  Caller holds ev_m.  wait_oldest: the ring is full -- wait for its oldest entry.
void drain_events(bk_engine* e, bool wait_oldest = false) {
    const size_t n = e->ev_ring.size();
    while (e->ev_pending) {
        const size_t i = (e->ev_head + n - e->ev_pending) % n;
        bk_engine::TimedLaunch& t = e->ev_ring[i];
        if (wait_oldest) {
            (void)hipEventSynchronize(t.stop);
            wait_oldest = false;
        } else if (hipEventQuery(t.stop) != hipSuccess) {
            (void)hipGetLastError();     // hipErrorNotReady is not an error
            break;
        }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess) {
            e->st.kernel_ms_sum += ms;
            bump(e->st.kernel_ms_count);
            e->st.last_kernel_ms = ms;
        }
        // how long the request sat between the host handing it over and the GPU starting its first kernel: GPU-side start
        // (measured from the epoch event) minus host-side enqueue time (measured from the instant the epoch was recorded)
        float since = 0.f;
        if (t.ticket && e->epoch && hipEventElapsedTime(&since, e->epoch, t.start) == hipSuccess) {
            e->st.queue_wait_ms_sum += std::max(0.0, (double)since - t.host_ms);
            bump(e->st.queue_wait_count);
        }
        --e->ev_pending;
    }
}

// Launch plan of the ordinary (one workgroup = 1..3 whole boards) forms: either one launch with the best single workgroup
// size, or k whole rounds of 3-board workgroups (one per CU) followed by a tail launch whose workgroup size makes the
// partial last round shortest -- e.g. 1,201 boards = 256 3-board workgroups + 217 2-board ones (1 + 0.77 rounds) instead
// of 401 3-board ones (2 rounds, the second with 111 CUs idle).  Costs are the measured per-round times.  A pure function
// of its arguments: enqueue() launches by it (with the engine's switches), bk_plan_flops() prices it (with the defaults).
struct LaunchPlan {
    int nb1;                       // single launch: boards per workgroup
    int head_p, head_v, tail_nb;   // tail_nb != 0: head_p + head_v 3-board workgroups first, the rest as tail_nb-board ones
};
LaunchPlan plan_launch(int B_policy, int B_value, int n_cu, int precision, const bk_plan_opts& o = bk_plan_opts()) {
    LaunchPlan pl{bk_pick_nb(B_policy, B_value, n_cu, precision, o), 0, 0, 0};
    const long single = bk_launch_cost(B_policy, B_value, pl.nb1, n_cu, precision);
    long best = single;
    const int full_p = B_policy / 3, full_v = B_value / 3;   // complete 3-board workgroups per net
    if (!o.force_nb && !o.no_split) {
        for (long k = 1; k * n_cu <= full_p + full_v; ++k) {
            const long head = k * n_cu;
            int hp = (int)std::min<long>(full_p, (head * full_p / (full_p + full_v)) & ~3L);  // keep the 4-block XCD pairing
            int hv = (int)(head - hp);
            if (hv > full_v) { hv = full_v; hp = (int)(head - hv); }
            const int rp = B_policy - 3 * hp, rv = B_value - 3 * hv;
            if (rp + rv == 0) break;
            const int nbt = bk_pick_nb(rp, rv, n_cu, precision, o);
            const long cost = k * 100 + bk_launch_cost(rp, rv, nbt, n_cu, precision) + (precision == BK_PRECISION_F16X2 ? 4 : 1);  // + a second launch's overhead
            // worth it from 3 % (f16x2: power-limited, the idle CUs of a ragged last round let the busy ones clock higher)
            // resp. 1 % (fp32: issue-limited at full clock, a shorter last round is a shorter launch)
            if (cost < best && cost * 100 <= single * (precision == BK_PRECISION_F16X2 ? 97 : 99)) { best = cost; pl.head_p = hp; pl.head_v = hv; pl.tail_nb = nbt; }
        }
    }
    return pl;
}

// enqueue one kernel launch on `stream`; all pointers are device pointers
// d_flag/tag: where and with what value the f16x2 kernel reports an activation outside the fp16 range.
// gated_redo (device-pointer path, f16x2): the exact-fp32 kernel follows on the same stream with the same launch plan,
// gated on d_flag[0] == tag -- stream-ordered, no host round trip, and a few microseconds when nothing overflowed.
// lo / hi: evaluate only positions [lo, hi) of the request (hi < 0: all B) -- a large host-buffer request is launched in two
// parts so that the first runs while the rest of its planes is still on the way (submit_common); buffers and row indices are
// those of the whole request.
int enqueue(bk_engine* e, const void* d_feats, int dtype, int B, int n_policy, int want, float* d_logits,
            float* d_probs, float* d_values, hipStream_t stream, int precision, unsigned int* d_flag,
            unsigned int tag = 1, bool gated_redo = false, bool allow_coop = false, int lo = 0, int hi = -1) {
    if (B == 0) return BK_OK;
    if (hi < 0) hi = B;
    const bool whole = lo == 0 && hi == B;
    bk_eval_args a{};
    a.net[0] = e->net[0];
    a.net[1] = e->net[1];
    a.feats = d_feats;
    a.feats_dtype = dtype;
    const int np_all = (want & (BK_WANT_LOGITS | BK_WANT_PROBS)) ? n_policy : 0;
    a.off_p = std::min(lo, np_all);
    a.B_policy = std::min(hi, np_all);
    a.off_v = (want & BK_WANT_VALUE) ? lo : 0;
    a.B_value = (want & BK_WANT_VALUE) ? hi : 0;
    a.logits = (want & BK_WANT_LOGITS) ? d_logits : nullptr;
    a.probs = (want & BK_WANT_PROBS) ? d_probs : nullptr;
    a.values = (want & BK_WANT_VALUE) ? d_values : nullptr;
    const int cnt_p = a.B_policy - a.off_p, cnt_v = a.B_value - a.off_v;   // rows of each net in this launch
    if (cnt_p + cnt_v == 0) return BK_OK;
#ifdef BK_STAMPS
    a.stamps = e->d_stamps;
#endif
    bool timed = false;
    size_t slot = 0;
    if (e->profiling) {
        std::lock_guard<std::mutex> g(e->ev_m);
        drain_events(e);
        if (e->ev_pending == e->ev_ring.size()) drain_events(e, /*wait_oldest=*/true);   // ring full
        slot = e->ev_head;
        e->ev_ring[slot].host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - e->epoch_host).count();
        e->ev_ring[slot].ticket = stream == e->stream && lo == 0;
        HIP_TRY(e, hipEventRecord(e->ev_ring[slot].start, stream));
        // the slot is taken now (its stop event is recorded below); until then a concurrent bk_stats must not fold it
        timed = true;
    }
    auto close_timing = [&]() -> int {
        if (!timed) return BK_OK;
        std::lock_guard<std::mutex> g(e->ev_m);
        HIP_TRY(e, hipEventRecord(e->ev_ring[slot].stop, stream));
        e->ev_head = (e->ev_head + 1) % e->ev_ring.size();
        ++e->ev_pending;
        return BK_OK;
    };
    a.overflow = d_flag;
    a.overflow_tag = tag;
    a.gate_count = 1;
    gated_redo = gated_redo && precision == BK_PRECISION_F16X2 && d_flag;
    auto launch = [&](const bk_eval_args& args, int nb) {
        if (precision != BK_PRECISION_F16X2) return bk_launch_leaf_eval(args, nb, stream);
        hipError_t rc = bk_launch_leaf_eval_f16(args, nb, stream);
        if (rc != hipSuccess || !gated_redo) return rc;
        bk_eval_args r = args;
        r.overflow = nullptr;
        r.gate = d_flag;
        r.gate_tag = tag;
        r.gate_counter = e->h_redo_dev + tag % BK_DEV_FLAGS;
        return bk_launch_leaf_eval(r, nb, stream);
    };
    // Small batches of the ticket path (engine's own stream, one exchange buffer): several CUs per board (bk_kernels.hip,
    // "cooperative form").  A workgroup that gives up waiting for its peers raises word 1 of the slot's flag block, which
    // travels to the host with the outputs: bk_wait then redoes the request with the one-CU form.
    const bool coop_ok = allow_coop && whole && precision == BK_PRECISION_F32 && stream == e->stream && e->d_coop_xchg && d_flag && !e->plan.force_nb;
    const int coop_form = coop_ok ? bk_coop_form(a.B_policy, a.B_value, e->n_cu, e->plan) : 0;   // 2..12 CUs per board, or three boards on 2 / 4 / 8
    if (const int slices = coop_form) {
        a.coop_xchg = e->d_coop_xchg;
        a.coop_sync = e->d_coop_sync;
        a.coop_err = d_flag + 1;
        a.coop_tag = 1;
#ifdef BK_TEST_HOOKS
        a.coop_fault = e->coop_fault;
#endif
        HIP_TRY(e, bk_launch_leaf_eval_coop(a, slices, stream));
        bump(e->st.coop_launches);
        if (const int trc = close_timing()) return trc;
        bump(e->st.evals, (uint64_t)B);
        bump(e->st.batches);
        if ((uint64_t)B > e->st.max_batch_seen) e->st.max_batch_seen = (uint64_t)B;
        return BK_OK;
    }
    const LaunchPlan pl = plan_launch(cnt_p, cnt_v, e->n_cu, precision, e->plan);
    const int nb1 = pl.nb1, head_p = pl.head_p, head_v = pl.head_v, tail_nb = pl.tail_nb;
    if (tail_nb) {
        bk_eval_args h = a;
        h.B_policy = a.off_p + 3 * head_p;
        h.B_value = a.off_v + 3 * head_v;
        h.gate_count = 0;  // the tail launch counts for the call
        HIP_TRY(e, launch(h, 3));
        bk_eval_args t = a;
        t.off_p = a.off_p + 3 * head_p;
        t.off_v = a.off_v + 3 * head_v;
        HIP_TRY(e, launch(t, tail_nb));
        bump(e->st.split_launches);
    } else {
        HIP_TRY(e, launch(a, nb1));
    }
    if (const int trc = close_timing()) return trc;
    bump(e->st.evals, (uint64_t)(hi - lo));
    bump(e->st.batches, lo == 0 ? 1 : 0);
    if ((uint64_t)B > e->st.max_batch_seen) e->st.max_batch_seen = (uint64_t)B;
    return BK_OK;
}

}  // namespace

extern "C" {

int bk_abi_version(void) { return BK_ABI_VERSION; }

int bk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int bk_engine_create(const bk_policy_weights* policy, const bk_value_weights* value, int device_id, int max_batch,
                     bk_engine** out) {
    if (!out) return fail(nullptr, BK_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!policy && !value) return fail(nullptr, BK_ERR_ARG, "at least one of policy/value weights is required");
    if (max_batch <= 0) return fail(nullptr, BK_ERR_ARG, "max_batch must be positive");
    if (!weights_ok(policy, nullptr)) return fail(nullptr, BK_ERR_ARG, "policy weights: NULL tensor pointer");
    if (!weights_ok(nullptr, value)) return fail(nullptr, BK_ERR_ARG, "value weights: NULL tensor pointer");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, BK_ERR_NO_GPU, "no HIP device visible (the engine has no CPU fallback)");
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, BK_ERR_ARG, "device_id out of range");

    bk_engine* e = new (std::nothrow) bk_engine();
    if (!e) return fail(nullptr, BK_ERR_OOM, "host allocation failed");
    e->device = device_id;
    e->max_batch = max_batch;
    e->has_policy = policy != nullptr;
    e->has_value = value != nullptr;
    e->precision = BK_PRECISION_F32;  // default: the reference's arithmetic width (torch fp32); f16x2 is opt-in
    // the environment is read here and nowhere else (see env_str): BK_PRECISION and the diagnostic switches
    if (const char* pz = env_str("BK_PRECISION")) e->precision = std::string(pz) == "f16x2" ? BK_PRECISION_F16X2 : BK_PRECISION_F32;
    e->plan.force_nb = env_int("BK_FORCE_NB", 0);
    e->plan.no_split = env_str("BK_NO_SPLIT") != nullptr;
    e->plan.coop = env_int("BK_COOP", -1);
    e->plan.coop3 = env_int("BK_COOP3", -1);
    e->no_direct = env_str("BK_NO_DIRECT") != nullptr;
    e->no_head_part = env_str("BK_NO_HEAD_PART") != nullptr;
    e->encode_overlap = env_str("BK_ENCODE_OVERLAP") != nullptr;
    e->no_fuse_encode = env_str("BK_NO_FUSE_ENCODE") != nullptr;
    e->direct_rows = env_int("BK_DIRECT_ROWS", 1024);
    e->copy_threads = env_int("BK_COPY_THREADS", 6);
    (void)roctx();                                           // BK_ROCTX: the marker library is looked up now, not by the first request
    int rc = BK_OK;
    auto bail = [&](int code) {
        g_create_error = e->err;
        bk_engine_destroy(e);
        return code;
    };
#define TRY_CREATE(call)                                                        \
    do {                                                                        \
        hipError_t _s = (call);                                                 \
        if (_s != hipSuccess) {                                                 \
            e->err = std::string(#call) + ": " + hipGetErrorString(_s);         \
            return bail(_s == hipErrorOutOfMemory ? BK_ERR_OOM : BK_ERR_HIP);   \
        }                                                                       \
    } while (0)
    TRY_CREATE(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    TRY_CREATE(hipGetDeviceProperties(&prop, device_id));
    e->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) {
        e->err = std::string("device is ") + prop.gcnArchName + ", this library carries gfx950 code only";
        return bail(BK_ERR_NO_GPU);
    }
    TRY_CREATE(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    TRY_CREATE(hipStreamCreateWithFlags(&e->s_in, hipStreamNonBlocking));
    TRY_CREATE(hipStreamCreateWithFlags(&e->s_out, hipStreamNonBlocking));

    if ((rc = load_weights(e, policy, value))) return bail(rc);
#ifdef BK_STAMPS
    TRY_CREATE(hipMalloc((void**)&e->d_stamps, (size_t)BK_STAMP_BLOCKS * 4 * 32 * 8));
    e->dev_allocs.push_back(e->d_stamps);
#endif
    TRY_CREATE(hipMalloc((void**)&e->d_dev_flag, BK_DEV_FLAGS * sizeof(unsigned int)));
    e->dev_allocs.push_back(e->d_dev_flag);
    TRY_CREATE(hipMemset(e->d_dev_flag, 0, BK_DEV_FLAGS * sizeof(unsigned int)));
    TRY_CREATE(hipHostMalloc((void**)&e->h_redo, BK_DEV_FLAGS * sizeof(unsigned int), hipHostMallocDefault));
    std::memset(e->h_redo, 0, BK_DEV_FLAGS * sizeof(unsigned int));
    TRY_CREATE(hipHostGetDevicePointer((void**)&e->h_redo_dev, e->h_redo, 0));
    {
        static_assert(BK_COOP_XCHG_BYTES >= (size_t)BK_COOP_MAX_TASKS * 2 * 81 * 128 * sizeof(float), "one buffer for both cooperative forms");
        const size_t xb = BK_COOP_XCHG_BYTES, sb = (size_t)BK_COOP_SYNC_WORDS * sizeof(unsigned int);
        TRY_CREATE(hipMalloc((void**)&e->d_coop_xchg, xb));
        e->dev_allocs.push_back(e->d_coop_xchg);
        TRY_CREATE(hipMalloc((void**)&e->d_coop_sync, sb));
        e->dev_allocs.push_back(e->d_coop_sync);
        TRY_CREATE(hipMemset(e->d_coop_sync, 0, sb));
    }
    for (auto& s : e->slots)
        if ((rc = alloc_slot(e, s))) return bail(rc);
    e->ev_ring.resize(512);
    for (auto& p : e->ev_ring) {
        TRY_CREATE(hipEventCreate(&p.start));
        TRY_CREATE(hipEventCreate(&p.stop));
    }
    TRY_CREATE(hipEventCreate(&e->epoch));
    TRY_CREATE(hipStreamSynchronize(e->stream));          // idle: the epoch event completes at once
    TRY_CREATE(hipEventRecord(e->epoch, e->stream));
    e->epoch_host = std::chrono::steady_clock::now();
    TRY_CREATE(hipEventSynchronize(e->epoch));
#undef TRY_CREATE
    *out = e;
    return BK_OK;
}

int bk_engine_destroy(bk_engine* e) {
    if (!e) return BK_ERR_ARG;
    (void)hipSetDevice(e->device);
    for (hipStream_t st : {e->s_in, e->stream, e->s_out})
        if (st) (void)hipStreamSynchronize(st);
    for (auto& s : e->slots) free_slot(s);
    for (auto& p : e->ev_ring) {
        if (p.start) (void)hipEventDestroy(p.start);
        if (p.stop) (void)hipEventDestroy(p.stop);
    }
    if (e->epoch) (void)hipEventDestroy(e->epoch);
    if (e->h_redo) (void)hipHostFree(e->h_redo);
    for (void* d : e->dev_allocs) (void)hipFree(d);
    for (hipStream_t st : {e->s_in, e->stream, e->s_out})
        if (st) (void)hipStreamDestroy(st);
    delete e->copy_pool;
    delete e;
    return BK_OK;
}

int bk_engine_set_weights(bk_engine* e, const bk_policy_weights* policy, const bk_value_weights* value) {
    if (!e) return BK_ERR_ARG;
    if (!policy && !value) return fail(e, BK_ERR_ARG, "neither policy nor value weights given");
    if ((policy && !e->has_policy) || (value && !e->has_value))
        return fail(e, BK_ERR_ARG, "the engine was created without that net");
    if (!weights_ok(policy, value)) return fail(e, BK_ERR_ARG, "weights: NULL tensor pointer");
    for (auto& s : e->slots)
        if (s.busy) return fail(e, BK_ERR_ARG, "tickets outstanding: bk_wait for them before replacing the weights");
    HIP_TRY(e, hipSetDevice(e->device));
    // bk_eval_device* launches may still run on caller streams (a non-blocking torch stream is not covered by the implicit
    // synchronisation of a null-stream copy): nothing on the device may read the old buffers when they are freed
    HIP_TRY(e, hipDeviceSynchronize());
    return load_weights(e, policy, value);      // all or nothing: fresh buffers, parameter blocks switched at the end
}

namespace {

constexpr int kSrcPositions = 2;  // beside BK_FEATS_F32 (0) / BK_FEATS_U8 (1)

int64_t submit_body(bk_engine* e, Slot* s, const void* src, int src_kind, int B, int n_policy, int want, float* logits,
                    float* probs, float* values);

// A submission failed part-way (a HIP call after the first enqueue returned an error): copies and kernels already queued
// still use the slot's staging and output blocks, and the slot is about to be handed to the next request.  Wait for
// everything the engine has queued, put the slot's flag words (device and pinned) back to zero and leave it free with
// nothing in flight; the engine's error text stays that of the failed call.  Later requests then run as usual.
void abort_submission(bk_engine* e, Slot* s) {
    const std::string why = e->err;
    for (hipStream_t st : {e->s_in, e->stream, e->s_out})
        if (st) (void)hipStreamSynchronize(st);
    if (s->d_out) (void)hipMemset(s->d_out, 0, 2 * sizeof(unsigned int));
    if (s->h_out) std::memset(s->h_out, 0, 2 * sizeof(unsigned int));
    (void)hipGetLastError();
    s->flag_dirty = false;
    s->direct = false;
    s->busy = false;
    bump(e->st.failed_submissions);
    e->err = why;
}

// common body of the ticket entry points; src_kind: BK_FEATS_F32, BK_FEATS_U8 or kSrcPositions
int64_t submit_common(bk_engine* e, const void* src, int src_kind, int B, int n_policy, int want, float* logits,
                      float* probs, float* values) {
    RoctxRange range(src_kind == kSrcPositions ? "bk_submit_positions" : "bk_submit", B, n_policy);
    int rc = check_want(e, B, n_policy, want);
    if (rc) return rc;
    if (B > 0 && !src) return fail(e, BK_ERR_ARG, src_kind == kSrcPositions ? "positions is NULL" : "feats is NULL");
    if (((want & BK_WANT_LOGITS) && !logits) || ((want & BK_WANT_PROBS) && !probs) || ((want & BK_WANT_VALUE) && !values))
        return fail(e, BK_ERR_ARG, "an output requested in `want` has a NULL buffer");
    Slot* s = nullptr;
    for (auto& c : e->slots)
        if (!c.busy) { s = &c; break; }
    if (!s) return fail(e, BK_ERR_ARG, "more than BK_MAX_INFLIGHT tickets outstanding");
    HIP_TRY(e, hipSetDevice(e->device));
#ifdef BK_TEST_HOOKS
    if (e->fault_submit > 0) {                            // tests: the n-th HIP call of this submission fails
        e->fault_at = e->fault_submit;
        e->fault_seen = 0;
    }
#endif
    const int64_t t = submit_body(e, s, src, src_kind, B, n_policy, want, logits, probs, values);
#ifdef BK_TEST_HOOKS
    e->fault_at = 0;
#endif
    if (t < 0) abort_submission(e, s);
    return t;
}

int64_t submit_body(bk_engine* e, Slot* s, const void* src, int src_kind, int B, int n_policy, int want, float* logits,
                    float* probs, float* values) {
    int rc = BK_OK;
    const size_t bytes = (size_t)B * (src_kind == kSrcPositions ? (size_t)BK_POS_BYTES : src_kind == BK_FEATS_U8 ? 2187 : 2187 * 4);
    int dtype = src_kind == BK_FEATS_F32 ? BK_FEATS_F32 : BK_FEATS_U8;  // what the leaf kernel reads from d_in
    // small requests (the single-tree genmove regime; for position records: up to direct_rows) have nothing to overlap: everything
    // goes on the compute stream and the two cross-stream event hops are saved; large ones use the three-stream chain
    const bool chained = B > (src_kind == kSrcPositions ? e->direct_rows : 256);
    hipStream_t sin = chained ? e->s_in : e->stream, sout = chained ? e->s_out : e->stream;
    // Small fp32 requests (the one-tree genmove regime: a 62-board expansion batch is a 108 us kernel) go "direct": no H2D
    // copy of position records -- the encoder reads them from the pinned slot over PCIe (12 KB) -- and no D2H copy -- the
    // leaf kernel writes the flag word and the outputs (a few hundred bytes) into the pinned output block, which the host
    // reads after the request's event.  Two of the five enqueues and ~12 us of copy kernels per round trip disappear
    // (rocprofv3 timeline, profiles/r03_genmove_timeline.md).  fp32 only: the f16x2 kernel's overflow flag is raised with
    // an atomic max, which is not used on host memory here.  Option no_direct = 1 restores the copies.
    const bool direct = !chained && B > 0 && e->precision == BK_PRECISION_F32 && s->h_in_dev && s->h_out_dev && !e->no_direct;
    s->direct = direct;
    if (B > 0) {
        void* d_dst = src_kind == kSrcPositions ? s->d_pos : s->d_in;
        const int copy_threads = e->copy_threads;
        // bytes [b0, b1) of the request: host buffer -> pinned slot -> device, on the copy-in stream
        auto stage = [&](size_t b0, size_t b1) -> int {
            const size_t n = b1 - b0;
            char* hsrc = static_cast<char*>(s->h_in) + b0;
            char* ddst = static_cast<char*>(d_dst) + b0;
            if (n >= ((size_t)4 << 20) && copy_threads > 0) {   // slices of >= 1 MiB, copied by the pool, H2D per slice
                if (!e->copy_pool) e->copy_pool = new (std::nothrow) CopyPool(std::min(copy_threads, 16));
                if (e->copy_pool) {
                    const size_t slice = std::max<size_t>((size_t)1 << 20, (n / 24 + 4095) & ~(size_t)4095);
                    hipError_t herr = hipSuccess;
                    const bool ok = e->copy_pool->copy(hsrc, static_cast<const char*>(src) + b0, n, slice, [&](int i) {
                        const size_t off = (size_t)i * slice;
                        const hipError_t rc = hipMemcpyAsync(ddst + off, hsrc + off, std::min(slice, n - off), hipMemcpyHostToDevice, sin);
                        if (rc != hipSuccess) herr = rc;
                        return rc == hipSuccess;
                    });
                    HIP_TRY(e, herr);
                    if (ok) return BK_OK;
                }
            }
            std::memcpy(hsrc, static_cast<const char*>(src) + b0, n);
            HIP_TRY(e, hipMemcpyAsync(ddst, hsrc, n, hipMemcpyHostToDevice, sin));
            return BK_OK;
        };
        // A large request of feature planes is evaluated in two parts: the first kHeadRows positions (two full rounds of
        // 3-board workgroups when both nets run) are launched as soon as THEIR planes have arrived and run while the rest is
        // still being staged and copied -- the reference-shaped call (f32 planes from host memory, 35.8 MB at B = 4,096)
        // then costs the kernel plus ~0.2 ms instead of plus 1.3 ms.  Results do not depend on how a request is launched.
        constexpr int kHeadRows = 768;
        const size_t row_bytes = bytes / (size_t)B;
        const bool two_part = chained && src_kind != kSrcPositions && B >= 3 * kHeadRows && bytes >= ((size_t)16 << 20) && !e->no_head_part;
        if (direct && src_kind == kSrcPositions) {
            std::memcpy(s->h_in, src, bytes);          // the encoder reads the records from here
        } else if (two_part) {
            if ((rc = stage(0, (size_t)kHeadRows * row_bytes))) return rc;
            HIP_TRY(e, hipEventRecord(s->head_ready, e->s_in));
            HIP_TRY(e, hipStreamWaitEvent(e->stream, s->head_ready, 0));
        } else if ((rc = stage(0, bytes))) {
            return rc;
        }
        // The encoder (13 us per 4,096 records) runs on the COMPUTE stream, in front of the request's leaf kernel.  Round 1
        // launched it on the copy-in stream so that it overlapped the previous request's leaf kernel; under that
        // kernel its workgroups wait for CUs, so it "ran" 110-370 us per call (6-8 % of the summed kernel time of a
        // self-play generation, profiles/r02_selfplay_*), for a kernel that needs 13 us.  Option encode_overlap = 1 restores that.
        const bool enc_overlap = e->encode_overlap != 0;
        // Small direct requests of position records: NO encoder launch at all -- the leaf kernel computes the planes from the records
        // while it stages them (bk_kernels.hip, stage_positions): one kernel instead of two on the critical path of a one-tree
        // search's request, ~8 us of a ~130-us round trip.  Larger requests keep the encoder kernel (13 us per 4,096 records at full
        // occupancy beats every workgroup encoding for itself).  Option no_fuse_encode = 1 restores the two kernels; same planes, same bits.
        const void* d_feats = s->d_in;
        if (direct && src_kind == kSrcPositions && !e->no_fuse_encode) {
            d_feats = s->h_in_dev;
            dtype = BK_FEATS_POS_;
            bump(e->st.positions_encoded, (uint64_t)B);
        } else if (src_kind == kSrcPositions && (enc_overlap || !chained)) {
            HIP_TRY(e, bk_launch_encode(direct ? s->h_in_dev : s->d_pos, B, static_cast<uint8_t*>(s->d_in), sin));
            bump(e->st.positions_encoded, (uint64_t)B);
        }
        if (chained && !two_part) {
            HIP_TRY(e, hipEventRecord(s->in_ready, e->s_in));
            HIP_TRY(e, hipStreamWaitEvent(e->stream, s->in_ready, 0));
        }
        if (src_kind == kSrcPositions && !enc_overlap && chained) {
            HIP_TRY(e, bk_launch_encode(s->d_pos, B, static_cast<uint8_t*>(s->d_in), e->stream));
            bump(e->st.positions_encoded, (uint64_t)B);
        }
        // output block of this request
        s->off_values = 64;
        s->off_probs = s->off_values + (((want & BK_WANT_VALUE) ? (size_t)B * 4 : 0) + 63) / 64 * 64;
        s->off_logits = s->off_probs + (((want & BK_WANT_PROBS) ? (size_t)n_policy * 81 * 4 : 0) + 63) / 64 * 64;
        s->out_bytes = s->off_logits + ((want & BK_WANT_LOGITS) ? (size_t)n_policy * 81 * 4 : 0);
        char* out_base = direct ? s->h_out_dev : s->d_out;
        unsigned int* d_flag = reinterpret_cast<unsigned int*>(out_base);
        if (direct) {
            std::memset(s->h_out, 0, 2 * sizeof(unsigned int));   // host write before the launch: visible to the kernel
        } else if (s->flag_dirty) {  // only after a flag was seen raised: both words are zero otherwise
            HIP_TRY(e, hipMemsetAsync(d_flag, 0, 2 * sizeof(unsigned int), e->stream));
            s->flag_dirty = false;
        }
        float* o_logits = reinterpret_cast<float*>(out_base + s->off_logits);
        float* o_probs = reinterpret_cast<float*>(out_base + s->off_probs);
        float* o_values = reinterpret_cast<float*>(out_base + s->off_values);
        if (two_part) {
            rc = enqueue(e, s->d_in, dtype, B, n_policy, want, o_logits, o_probs, o_values, e->stream, e->precision, d_flag, 1, false,
                         false, 0, kHeadRows);
            if (rc) return rc;
            if ((rc = stage((size_t)kHeadRows * row_bytes, bytes))) return rc;       // ... while the first part runs
            HIP_TRY(e, hipEventRecord(s->in_ready, e->s_in));
            HIP_TRY(e, hipStreamWaitEvent(e->stream, s->in_ready, 0));
            rc = enqueue(e, s->d_in, dtype, B, n_policy, want, o_logits, o_probs, o_values, e->stream, e->precision, d_flag, 1, false,
                         false, kHeadRows, B);
        } else {
            rc = enqueue(e, d_feats, dtype, B, n_policy, want, o_logits, o_probs, o_values, e->stream, e->precision, d_flag, 1, false,
                         /*allow_coop=*/true);
        }
        if (rc) return rc;
        if (chained) {
            HIP_TRY(e, hipEventRecord(s->computed, e->stream));
            HIP_TRY(e, hipStreamWaitEvent(e->s_out, s->computed, 0));
        }
        if (!direct) HIP_TRY(e, hipMemcpyAsync(s->h_out, s->d_out, s->out_bytes, hipMemcpyDeviceToHost, sout));  // flag + all outputs
    }
    HIP_TRY(e, hipEventRecord(s->done, sout));
    s->busy = true;
    s->ticket = e->next_ticket++;
    s->B = B;
    s->n_policy = n_policy;
    s->dtype = dtype;
    s->d_feats = dtype == BK_FEATS_POS_ ? s->h_in_dev : s->d_in;
    s->want = want;
    s->logits = logits;
    s->probs = probs;
    s->values = values;
    return s->ticket;
}

}  // namespace

int64_t bk_submit_prefix(bk_engine* e, const void* feats, int feats_dtype, int B, int n_policy, int want,
                         float* logits, float* probs, float* values) {
    if (!e) return BK_ERR_ARG;
    if (feats_dtype != BK_FEATS_F32 && feats_dtype != BK_FEATS_U8) return fail(e, BK_ERR_ARG, "bad feats_dtype");
    return submit_common(e, feats, feats_dtype, B, n_policy, want, logits, probs, values);
}

int64_t bk_submit_positions(bk_engine* e, const void* positions, int B, int n_policy, int want, float* logits,
                            float* probs, float* values) {
    if (!e) return BK_ERR_ARG;
    return submit_common(e, positions, kSrcPositions, B, n_policy, want, logits, probs, values);
}

int bk_encode_positions(bk_engine* e, const void* positions, int B, uint8_t* planes) {
    if (!e) return BK_ERR_ARG;
    if (B < 0) return fail(e, BK_ERR_ARG, "bad B");
    if (B > e->max_batch) return fail(e, BK_ERR_BATCH, "B exceeds max_batch given at bk_engine_create");
    if (B == 0) return BK_OK;
    if (!positions || !planes) return fail(e, BK_ERR_ARG, "positions or planes is NULL");
    Slot* s = nullptr;
    for (auto& c : e->slots)
        if (!c.busy) { s = &c; break; }
    if (!s) return fail(e, BK_ERR_ARG, "more than BK_MAX_INFLIGHT tickets outstanding");
    HIP_TRY(e, hipSetDevice(e->device));
    // pinned staging: records at the front of h_in, the planes behind them (h_in holds B x 8,748 bytes)
    uint8_t* h_planes = static_cast<uint8_t*>(s->h_in) + (((size_t)B * BK_POS_BYTES + 255) & ~(size_t)255);
    std::memcpy(s->h_in, positions, (size_t)B * BK_POS_BYTES);
    HIP_TRY(e, hipMemcpyAsync(s->d_pos, s->h_in, (size_t)B * BK_POS_BYTES, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e, bk_launch_encode(s->d_pos, B, static_cast<uint8_t*>(s->d_in), e->stream));
    HIP_TRY(e, hipMemcpyAsync(h_planes, s->d_in, (size_t)B * 2187, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    std::memcpy(planes, h_planes, (size_t)B * 2187);
    bump(e->st.positions_encoded, (uint64_t)B);
    return BK_OK;
}

int bk_wait(bk_engine* e, int64_t ticket) {
    if (!e) return BK_ERR_ARG;
    RoctxRange range("bk_wait");
    for (auto& s : e->slots) {
        if (!s.busy || s.ticket != ticket) continue;
        {
            const auto w0 = std::chrono::steady_clock::now();
            HIP_TRY(e, hipEventSynchronize(s.done));
            e->st.host_wait_ms_sum += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
        }
        const unsigned int* hf = reinterpret_cast<const unsigned int*>(s.h_out);
        if (s.B > 0 && (hf[0] || hf[1])) {
            // word 0: the f16x2 kernel clamped an activation -- redo this request on the exact fp32 kernel;
            // word 1: a workgroup of the cooperative launch gave up waiting for its peers -- redo with one CU per board
            // (the arrival counters are left anywhere and the engine's poison word is up: clear both, in stream order.
            // Cooperative requests that were queued behind the failed one ran before this memset: they saw the poison
            // word, raised their own word 1 and are redone here as well when their turn to be waited for comes)
            if (hf[1]) {
                bump(e->st.coop_fallbacks);
                HIP_TRY(e, hipMemsetAsync(e->d_coop_sync, 0, (size_t)BK_COOP_SYNC_WORDS * sizeof(unsigned int), e->stream));  // counters + poison word
            } else {
                bump(e->st.f16_overflow_fallbacks);
            }
            s.flag_dirty = true;
            int rc = enqueue(e, s.d_feats, s.dtype, s.B, s.n_policy, s.want, reinterpret_cast<float*>(s.d_out + s.off_logits),
                             reinterpret_cast<float*>(s.d_out + s.off_probs), reinterpret_cast<float*>(s.d_out + s.off_values),
                             e->stream, BK_PRECISION_F32, nullptr);
            if (rc) return rc;
            HIP_TRY(e, hipMemcpyAsync(s.h_out, s.d_out, s.out_bytes, hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(e, hipStreamSynchronize(e->stream));
            std::memset(s.h_out, 0, 2 * sizeof(unsigned int));
        }
        if (s.want & BK_WANT_LOGITS) std::memcpy(s.logits, s.h_out + s.off_logits, (size_t)s.n_policy * 81 * 4);
        if (s.want & BK_WANT_PROBS) std::memcpy(s.probs, s.h_out + s.off_probs, (size_t)s.n_policy * 81 * 4);
        if (s.want & BK_WANT_VALUE) std::memcpy(s.values, s.h_out + s.off_values, (size_t)s.B * 4);
        s.busy = false;
        return BK_OK;
    }
    return fail(e, BK_ERR_ARG, "unknown or already-waited ticket");
}

int64_t bk_submit(bk_engine* e, const void* feats, int feats_dtype, int B, int want, float* logits, float* probs,
                  float* values) {
    return bk_submit_prefix(e, feats, feats_dtype, B, B, want, logits, probs, values);
}

int bk_eval(bk_engine* e, const float* feats, int B, int want, float* logits, float* probs, float* values) {
    const int64_t t = bk_submit(e, feats, BK_FEATS_F32, B, want, logits, probs, values);
    if (t < 0) return (int)t;
    return bk_wait(e, t);
}

int bk_eval_u8(bk_engine* e, const uint8_t* feats, int B, int want, float* logits, float* probs, float* values) {
    const int64_t t = bk_submit(e, feats, BK_FEATS_U8, B, want, logits, probs, values);
    if (t < 0) return (int)t;
    return bk_wait(e, t);
}

int bk_eval_device(bk_engine* e, const void* d_feats, int feats_dtype, int B, int want, float* d_logits,
                   float* d_probs, float* d_values, void* stream) {
    return bk_eval_device_prefix(e, d_feats, feats_dtype, B, B, want, d_logits, d_probs, d_values, stream);
}

int bk_eval_device_prefix(bk_engine* e, const void* d_feats, int feats_dtype, int B, int n_policy, int want,
                          float* d_logits, float* d_probs, float* d_values, void* stream) {
    RoctxRange range("bk_eval_device", B, n_policy);
    int rc = check_want(e, B, n_policy, want);
    if (rc) return rc;
    if (feats_dtype != BK_FEATS_F32 && feats_dtype != BK_FEATS_U8) return fail(e, BK_ERR_ARG, "bad feats_dtype");
    if (B > 0 && !d_feats) return fail(e, BK_ERR_ARG, "d_feats is NULL");
    if (((want & BK_WANT_LOGITS) && !d_logits) || ((want & BK_WANT_PROBS) && !d_probs) ||
        ((want & BK_WANT_VALUE) && !d_values))
        return fail(e, BK_ERR_ARG, "an output requested in `want` has a NULL buffer");
    HIP_TRY(e, hipSetDevice(e->device));
    // `stream` is used as given: NULL is HIP's null (legacy default) stream, which is also what
    // torch.cuda.current_stream() is unless the caller switched streams.
    if (++e->dev_seq == 0) e->dev_seq = 1;  // 0 is the flag's reset value
    unsigned int* flag = e->d_dev_flag + e->dev_seq % BK_DEV_FLAGS;   // this call's own word (see d_dev_flag)
    if (e->precision == BK_PRECISION_F16X2) {                         // ... and its redo word: count what the call 256 ago left there
        std::lock_guard<std::mutex> g(e->ev_m);
        fold_redo(e, e->dev_seq % BK_DEV_FLAGS);
    }
    if (e->precision == BK_PRECISION_F16X2 && B > 0) HIP_TRY(e, hipMemsetAsync(flag, 0, sizeof(unsigned int), static_cast<hipStream_t>(stream)));
    return enqueue(e, d_feats, feats_dtype, B, n_policy, want, d_logits, d_probs, d_values,
                   static_cast<hipStream_t>(stream), e->precision, flag, e->dev_seq, true);
}

int bk_engine_set_precision(bk_engine* e, int precision) {
    if (!e) return BK_ERR_ARG;
    if (precision != BK_PRECISION_F32 && precision != BK_PRECISION_F16X2) return fail(e, BK_ERR_ARG, "unknown precision");
    e->precision = precision;
    return BK_OK;
}

int bk_engine_get_precision(bk_engine* e) { return e ? e->precision : BK_ERR_ARG; }

int bk_engine_set_profiling(bk_engine* e, int on) {
    if (!e) return BK_ERR_ARG;
    e->profiling = on != 0;
    return BK_OK;
}

int bk_engine_synchronize(bk_engine* e) {
    if (!e) return BK_ERR_ARG;
    HIP_TRY(e, hipSetDevice(e->device));
    for (hipStream_t st : {e->s_in, e->stream, e->s_out}) HIP_TRY(e, hipStreamSynchronize(st));
    return BK_OK;
}

int bk_stats(bk_engine* e, bk_stats_t* out) {
    // No call in here waits for the device's other streams (it used to be a hipDeviceSynchronize: a monitor polling the
    // counters stalled every stream of every engine on the card).  Timing events that have completed are folded, the
    // others stay pending; the redone device-path calls are read out of pinned memory (h_redo) -- a snapshot: calls still in
    // flight on caller streams are counted once they have run.
    if (!e || !out) return BK_ERR_ARG;
    {
        std::lock_guard<std::mutex> g(e->ev_m);
        if (e->ev_pending) drain_events(e);
        if (e->h_redo && e->dev_seq)                  // bk_eval_device* calls whose f16x2 kernel overflowed and were redone in fp32
            for (unsigned int i = 0; i < BK_DEV_FLAGS; ++i) fold_redo(e, i);
        bk_stats_t snap;
        const uint64_t* src = reinterpret_cast<const uint64_t*>(&e->st);
        uint64_t* dst = reinterpret_cast<uint64_t*>(&snap);
        static_assert(sizeof(bk_stats_t) % 8 == 0, "bk_stats_t is a sequence of 8-byte fields");
        for (size_t i = 0; i < sizeof(bk_stats_t) / 8; ++i) dst[i] = __atomic_load_n(src + i, __ATOMIC_RELAXED);
        snap.mean_batch = snap.batches ? (double)snap.evals / (double)snap.batches : 0.0;
        *out = snap;
    }
    return BK_OK;
}

int bk_engine_max_batch(bk_engine* e) { return e ? e->max_batch : BK_ERR_ARG; }

int bk_plan_query(int n_policy, int n_value, int n_cu, int precision, int* boards_per_workgroup) {
    if (n_policy < 0 || n_value < 0 || n_cu <= 0 || (precision != BK_PRECISION_F32 && precision != BK_PRECISION_F16X2)) return BK_ERR_ARG;
    if (boards_per_workgroup) *boards_per_workgroup = bk_pick_nb(n_policy, n_value, n_cu, precision);
    if (precision != BK_PRECISION_F32) return 0;
    return bk_coop_form(n_policy, n_value, n_cu);       // 2..12, or BK_COOP3_FORM_2 / _4 / _8 (102 / 104 / 108): three boards on 2 / 4 / 8 CUs
}

int bk_plan_flops(int n_policy, int n_value, int n_cu, int cooperative, double* executed_mfma_flop, double* algorithmic_flop,
                  int* n_launches) {
    if (n_policy < 0 || n_value < 0 || n_cu <= 0) return BK_ERR_ARG;
    // valid-tap multiply-adds of one position, PolicyNet (trunk + 1x1 head) and ValueNet (+ the 81-64-1 MLP)
    const double alg = 2.0 * (66706944.0 * n_policy + 66712192.0 * n_value);
    double exe = 0.0;
    int launches = 0;
    if (n_policy + n_value > 0) {
        const int form = cooperative ? bk_coop_form(n_policy, n_value, n_cu) : 0;
        if (form > 0 && form < 100) {
            exe = bk_coop_mfma_flop_per_task(form) * (n_policy + n_value);   // the one-board tile set, dealt out to the slices
            launches = 1;
        } else if (form >= 100) {
            exe = bk_mfma_flop_per_workgroup(3) * ((n_policy + 2) / 3 + (n_value + 2) / 3);   // the 3-board tile set, shared by 2 / 4 CUs
            launches = 1;
        } else {
            const LaunchPlan pl = plan_launch(n_policy, n_value, n_cu, BK_PRECISION_F32);
            auto wgs = [](int b, int nb) { return (double)((b + nb - 1) / nb); };
            if (pl.tail_nb) {
                exe = bk_mfma_flop_per_workgroup(3) * (pl.head_p + pl.head_v) +
                      bk_mfma_flop_per_workgroup(pl.tail_nb) * (wgs(n_policy - 3 * pl.head_p, pl.tail_nb) + wgs(n_value - 3 * pl.head_v, pl.tail_nb));
                launches = 2;
            } else {
                exe = bk_mfma_flop_per_workgroup(pl.nb1) * (wgs(n_policy, pl.nb1) + wgs(n_value, pl.nb1));
                launches = 1;
            }
        }
    }
    if (executed_mfma_flop) *executed_mfma_flop = exe;
    if (algorithmic_flop) *algorithmic_flop = alg;
    if (n_launches) *n_launches = launches;
    return BK_OK;
}

int bk_has_test_hooks(void) {
#ifdef BK_TEST_HOOKS
    return 1;
#else
    return 0;
#endif
}

namespace {
int* option_field(bk_engine* e, const std::string& n) {
    if (n == "force_nb") return &e->plan.force_nb;
    if (n == "no_split") return &e->plan.no_split;
    if (n == "coop") return &e->plan.coop;
    if (n == "coop3") return &e->plan.coop3;
    if (n == "no_direct") return &e->no_direct;
    if (n == "no_head_part") return &e->no_head_part;
    if (n == "encode_overlap") return &e->encode_overlap;
    if (n == "no_fuse_encode") return &e->no_fuse_encode;
    if (n == "direct_rows") return &e->direct_rows;
    if (n == "copy_threads") return &e->copy_threads;
#ifdef BK_TEST_HOOKS
    if (n == "coop_fault") return &e->coop_fault;
    if (n == "fault_submit") return &e->fault_submit;
#endif
    return nullptr;
}
}  // namespace

int bk_engine_set_option(bk_engine* e, const char* name, int value) {
    if (!e || !name) return BK_ERR_ARG;
    int* f = option_field(e, name);
    if (!f) return fail(e, BK_ERR_ARG, std::string("unknown engine option '") + name + "'");
    // the switches are plain ints that submissions read: like every other call on an engine handle (single consumer)
    // this one belongs to the thread that submits, between two submissions; values a switch does not know are refused, not ignored
    const std::string n = name;
    auto one_of = [&](std::initializer_list<int> ok) { return std::find(ok.begin(), ok.end(), value) != ok.end(); };
    bool ok = true;
    if (n == "force_nb") ok = value >= 0 && value <= 3;
    else if (n == "coop") ok = one_of({-1, 0, 2, 3, 4, 6, 8, 12});
    else if (n == "coop3") ok = one_of({-1, 0, 2, 4, 8});
    else if (n == "copy_threads") ok = value >= 0 && value <= 64;
    else if (n == "direct_rows") ok = value >= 0 && value <= (1 << 20);
    else if (n == "coop_fault" || n == "fault_submit") ok = value >= 0;   // (test builds: fault_submit = the HIP call of a submission that fails)
    else ok = value == 0 || value == 1;              // no_split, no_direct, no_head_part, encode_overlap, no_fuse_encode
    if (!ok) return fail(e, BK_ERR_ARG, std::string("engine option '") + name + "': value " + std::to_string(value) + " out of range");
    for (const auto& sl : e->slots)
        if (sl.busy && (n == "coop" || n == "coop3" || n == "force_nb" || n == "no_split"))
            return fail(e, BK_ERR_ARG, std::string("engine option '") + name + "' changes the launch planner: wait for the tickets in flight first");
    *f = value;
    return BK_OK;
}

int bk_engine_get_option(bk_engine* e, const char* name, int* value) {
    if (!e || !name || !value) return BK_ERR_ARG;
    const int* f = option_field(e, name);
    if (!f) return fail(e, BK_ERR_ARG, std::string("unknown engine option '") + name + "'");
    *value = *f;
    return BK_OK;
}

/* the engine as the evaluator of the native step loop (include/bokego_tree.h, bk_pools_run) */
namespace {
int64_t evaluator_submit(void* ctx, const bk_pos* recs, int B, int n_policy, float* probs, float* values) {
    bk_engine* e = static_cast<bk_engine*>(ctx);
    const int want = (n_policy > 0 && e->has_policy ? BK_WANT_PROBS : 0) | (e->has_value ? BK_WANT_VALUE : 0);
    return bk_submit_positions(e, recs, B, n_policy, want, nullptr, probs, values);
}
int evaluator_wait(void* ctx, int64_t ticket) { return bk_wait(static_cast<bk_engine*>(ctx), ticket); }
}  // namespace

int bk_engine_evaluator(bk_engine* e, bk_evaluator* out) {
    if (!e || !out) return BK_ERR_ARG;
    out->ctx = e;
    out->submit = evaluator_submit;
    out->wait = evaluator_wait;
    return BK_OK;
}

#ifdef BK_TEST_HOOKS
// test builds only (not part of include/bokego_amd.h; tests/test_gpu_hooks.py): the n-th HIP call the engine makes from now
// on reports a failure instead of being made, once.  The option "fault_submit" arms the same counter for every ticket submission.
int bk_debug_fail_nth_hip_call(bk_engine* e, int n) {
    if (!e) return BK_ERR_ARG;
    e->fault_at = n;
    e->fault_seen = 0;
    return BK_OK;
}
#endif

#ifdef BK_STAMPS
// diagnostic builds only (not part of include/bokego_amd.h)
int bk_debug_read_stamps(bk_engine* e, unsigned long long* out, int n_blocks) {
    if (!e || n_blocks > BK_STAMP_BLOCKS) return BK_ERR_ARG;
    HIP_TRY(e, hipDeviceSynchronize());
    HIP_TRY(e, hipMemcpy(out, e->d_stamps, (size_t)n_blocks * 4 * 32 * 8, hipMemcpyDeviceToHost));
    return BK_OK;
}
#endif

const char* bk_last_error(bk_engine* e) { return e ? e->err.c_str() : g_create_error.c_str(); }

}  // extern "C"
