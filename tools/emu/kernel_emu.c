/* kernel_emu.c -- CPU emulation of the SUMMATION ORDER of the fp32 leaf kernel (bokego_amd/csrc/bk_kernels.hip), for
 * tools/error_budget.py: which layer spends how much of the 1e-4 parity budget, and what a different order would buy.
 * A measuring tool, not a product path and not the oracle: nothing in bokego_amd/ or bench.py loads it.
 *
 * What the kernel does per output (position, output channel), reproduced here term by term:
 *   acc = 0; for every tap (row-major), every group g of 16 input slots, every MFMA k-step j, every lane quad kq (the k index
 *   of v_mfma_f32_16x16x4_f32, accumulated in ascending order): acc = fma(w, x, acc) with input channel bk_slot_perm(16g+4kq+j)
 *   (layer 0: 27 planes, group 1 has 3 k-steps, planes 24..26 in its third); then max(acc + bias, 0).
 *   Taps that point off the board contribute fma(w, 0, acc) = acc, whether the kernel skips them or not.
 * Variants (per layer, bit masks): two accumulators by tap parity, added at the end; float64 accumulation ("this layer exact").
 * Heads: the 128-term 1x1 dot as ONE chain (shipped until round 3) or as 4 chains (element e of every float4) combined pairwise;
 * the 81-term lin1 sum likewise; float64 variants.  gcc -O3 -march=native -ffp-contract=off -fopenmp.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct {
    const float* w0;      /* [25][27][128]  layer 0: tap, chain position, cout (folded) */
    const int32_t* ch0;   /* [27]           input plane of each chain position */
    const float* w3;      /* [6][9][128][128] layers 1..6: tap, chain position, cout */
    const int32_t* ch3;   /* [128]          input channel of each chain position */
    const float* bias;    /* [7][128] */
    const float* head_w;  /* [128] */
    const float* head_b;  /* [81] */
    const float* lin1_wt; /* [81][64] or NULL (policy net) */
    const float* lin1_b;  /* [64] */
    const float* lin2_w;  /* [64] */
    float lin2_b;
} emu_net;

int emu_split3 = 5, emu_split5 = 12;   /* first tap of the second chain (3x3 / 5x5 layers) */
static void conv_point(const float* in /*[C][81]*/, int C, int kw, int y, int x, const float* w /*[taps][nk][128]*/, const int32_t* ch,
                       int nk, const float* bias, int two_acc, int exact, float* out /*[128]*/) {
    const int r = kw / 2;
    if (exact) {
        double acc[128];
        for (int co = 0; co < 128; ++co) acc[co] = 0.0;
        for (int t = 0; t < kw * kw; ++t) {
            const int yy = y + t / kw - r, xx = x + t % kw - r;
            if (yy < 0 || yy > 8 || xx < 0 || xx > 8) continue;
            const float* wt = w + (size_t)t * nk * 128;
            for (int k = 0; k < nk; ++k) {
                const double xv = in[ch[k] * 81 + yy * 9 + xx];
                const float* wk = wt + k * 128;
                for (int co = 0; co < 128; ++co) acc[co] += (double)wk[co] * xv;
            }
        }
        for (int co = 0; co < 128; ++co) {
            const double v = acc[co] + (double)bias[co];
            out[co] = (float)(v > 0 ? v : 0);
        }
        (void)C;
        return;
    }
    float acc[2][128];
    memset(acc, 0, sizeof acc);
    for (int t = 0; t < kw * kw; ++t) {
        const int yy = y + t / kw - r, xx = x + t % kw - r;
        if (yy < 0 || yy > 8 || xx < 0 || xx > 8) continue;
        /* two_acc: 1 = by tap parity, 2 = first / second half of the taps as the kernel splits them (3x3: 0..4 | 5..8, 5x5: 0..11 | 12..24) */
        float* a = acc[two_acc == 1 ? (t & 1) : two_acc >= 2 ? (t >= (kw == 3 ? emu_split3 : emu_split5)) : 0];
        const float* wt = w + (size_t)t * nk * 128;
        for (int k = 0; k < nk; ++k) {
            const float xv = in[ch[k] * 81 + yy * 9 + xx];
            const float* wk = wt + k * 128;
            for (int co = 0; co < 128; ++co) a[co] = __builtin_fmaf(wk[co], xv, a[co]);
        }
    }
    for (int co = 0; co < 128; ++co) {
        const float s = two_acc ? acc[0][co] + acc[1][co] : acc[0][co];
        const float v = s + bias[co];
        out[co] = v > 0.f ? v : 0.f;
    }
}

/* one position through one net.  planes [27][81]; acts (optional) [7][128][81]; logits_or_head [81] = the 1x1 head's output
 * (policy: the logits; value: the input of the MLP before its ReLU); value_out: {pre-tanh, tanh} (value net only) */
void emu_forward(const emu_net* N, const float* planes, int two_acc_mask, int exact_mask, int head_mode /*0 chain, 1 four chains, 2 f64*/,
                 int lin_mode, float* acts, float* head_out, float* value_out) {
    const int split_kind = (two_acc_mask >> 8) ? 2 : 1;   /* bit 8 of the mask: halves instead of parity */
    float a[128 * 81], b[128 * 81], pt[128];
    for (int q = 0; q < 81; ++q) {
        conv_point(planes, 27, 5, q / 9, q % 9, N->w0, N->ch0, 27, N->bias, (two_acc_mask & 1) ? split_kind : 0, exact_mask & 1, pt);
        for (int c = 0; c < 128; ++c) a[c * 81 + q] = pt[c];
    }
    if (acts) memcpy(acts, a, sizeof a);
    float *cur = a, *nxt = b;
    for (int L = 1; L < 7; ++L) {
        for (int q = 0; q < 81; ++q) {
            conv_point(cur, 128, 3, q / 9, q % 9, N->w3 + (size_t)(L - 1) * 9 * 128 * 128, N->ch3, 128, N->bias + L * 128,
                       ((two_acc_mask >> L) & 1) ? split_kind : 0, (exact_mask >> L) & 1, pt);
            for (int c = 0; c < 128; ++c) nxt[c * 81 + q] = pt[c];
        }
        if (acts) memcpy(acts + (size_t)L * 128 * 81, nxt, sizeof a);
        float* t = cur; cur = nxt; nxt = t;
    }
    float h[81];
    for (int q = 0; q < 81; ++q) {
        if (head_mode == 2) {
            double d = 0;
            for (int c = 0; c < 128; ++c) d += (double)cur[c * 81 + q] * (double)N->head_w[c];
            h[q] = (float)(d + (double)N->head_b[q]);
        } else if (head_mode == 1) {
            float d[4] = {0.f, 0.f, 0.f, 0.f};
            for (int cc = 0; cc < 32; ++cc)
                for (int e = 0; e < 4; ++e) d[e] = __builtin_fmaf(cur[(4 * cc + e) * 81 + q], N->head_w[4 * cc + e], d[e]);
            h[q] = ((d[0] + d[1]) + (d[2] + d[3])) + N->head_b[q];
        } else {
            float d = 0.f;
            for (int c = 0; c < 128; ++c) d = __builtin_fmaf(cur[c * 81 + q], N->head_w[c], d);
            h[q] = d + N->head_b[q];
        }
    }
    memcpy(head_out, h, sizeof h);
    if (!N->lin1_wt) return;
    float hv[81], z[64];
    for (int q = 0; q < 81; ++q) hv[q] = h[q] > 0.f ? h[q] : 0.f;
    for (int j = 0; j < 64; ++j) {
        if (lin_mode == 2) {
            double s = N->lin1_b[j];
            for (int q = 0; q < 81; ++q) s += (double)N->lin1_wt[q * 64 + j] * (double)hv[q];
            z[j] = (float)(s > 0 ? s : 0);
        } else if (lin_mode == 1) {
            /* four chains (q mod 4), the bias opening chain 0, combined pairwise */
            float s[4] = {N->lin1_b[j], 0.f, 0.f, 0.f};
            for (int q = 0; q < 81; ++q) s[q & 3] = __builtin_fmaf(N->lin1_wt[q * 64 + j], hv[q], s[q & 3]);
            const float v = (s[0] + s[1]) + (s[2] + s[3]);
            z[j] = v > 0.f ? v : 0.f;
        } else {
            float s = N->lin1_b[j];
            for (int q = 0; q < 81; ++q) s = __builtin_fmaf(N->lin1_wt[q * 64 + j], hv[q], s);
            z[j] = s > 0.f ? s : 0.f;
        }
    }
    float v[64], u[64];
    if (lin_mode == 2) {
        double s = 0;
        for (int j = 0; j < 64; ++j) s += (double)z[j] * (double)N->lin2_w[j];
        value_out[0] = (float)(s + (double)N->lin2_b);
    } else {
        for (int j = 0; j < 64; ++j) v[j] = z[j] * N->lin2_w[j];
        for (int o = 32; o > 0; o >>= 1) {                 /* wave_sum: xor butterfly */
            for (int j = 0; j < 64; ++j) u[j] = v[j] + v[j ^ o];
            memcpy(v, u, sizeof v);
        }
        value_out[0] = v[0] + N->lin2_b;
    }
    value_out[1] = (float)tanh((double)value_out[0]);
}

/* B positions, in parallel; planes [B][27][81] uint8 */
void emu_batch(const emu_net* N, const uint8_t* planes, int B, int two_acc_mask, int exact_mask, int head_mode, int lin_mode,
               float* head_out /*[B][81]*/, float* value_out /*[B][2] or NULL*/) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < B; ++i) {
        float x[27 * 81], vo[2] = {0.f, 0.f};
        for (int k = 0; k < 27 * 81; ++k) x[k] = (float)planes[(size_t)i * 2187 + k];
        emu_forward(N, x, two_acc_mask, exact_mask, head_mode, lin_mode, 0, head_out + (size_t)i * 81, vo);
        if (value_out) { value_out[2 * i] = vo[0]; value_out[2 * i + 1] = vo[1]; }
    }
}
