/*
 * bk_demo.c -- the C ABI without Python: load two BKW1 weight files, encode the empty board with
 * libbkgo, evaluate it on the GPU with libbokego_amd, print the policy's best move and the value.
 *
 *   hipcc (or gcc) examples/bk_demo.c -Iinclude -Lbokego_amd -lbokego_amd -lbkgo -Wl,-rpath,$PWD/bokego_amd -o bk_demo
 *   ./bk_demo tests/golden/policy_19.bkw tests/golden/value_synth.bkw
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bokego_amd.h"
#include "bokego_go.h"

/* BKW1 (bokego_amd/bkw.py): "BKW1" u32 version u32 n u32 reserved, then n packed 84-byte entries
 * { char name[48]; u32 ndim; u32 dims[4]; u64 offset; u64 nelem }, then 16-byte aligned fp32 data. */
enum { BKW_HEAD = 16, BKW_ENTRY = 84, BKW_OFF_OFFSET = 68 };

static unsigned char *slurp(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(1); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    unsigned char *b = malloc(n);
    if (fread(b, 1, n, f) != (size_t)n) { perror("read"); exit(1); }
    fclose(f);
    return b;
}

static const float *tensor(const unsigned char *file, const char *name) {
    unsigned n;
    if (memcmp(file, "BKW1", 4)) { fprintf(stderr, "not a BKW1 file\n"); exit(1); }
    memcpy(&n, file + 8, 4);
    for (unsigned i = 0; i < n; ++i) {
        const unsigned char *e = file + BKW_HEAD + (size_t)i * BKW_ENTRY;
        unsigned long long off;
        memcpy(&off, e + BKW_OFF_OFFSET, 8);
        if (!strncmp((const char *)e, name, 48)) return (const float *)(file + off);
    }
    fprintf(stderr, "tensor %s missing\n", name);
    exit(1);
}

static void fill_trunk(bk_trunk_weights *t, const unsigned char *f) {
    static const int conv[7] = {0, 3, 6, 9, 12, 15, 18};
    char k[64];
    for (int l = 0; l < 7; ++l) {
        snprintf(k, sizeof k, "conv.%d.weight", conv[l]); t->conv_w[l] = tensor(f, k);
        snprintf(k, sizeof k, "conv.%d.bias", conv[l]); t->conv_b[l] = tensor(f, k);
        snprintf(k, sizeof k, "conv.%d.weight", conv[l] + 1); t->bn_w[l] = tensor(f, k);
        snprintf(k, sizeof k, "conv.%d.bias", conv[l] + 1); t->bn_b[l] = tensor(f, k);
        snprintf(k, sizeof k, "conv.%d.running_mean", conv[l] + 1); t->bn_mean[l] = tensor(f, k);
        snprintf(k, sizeof k, "conv.%d.running_var", conv[l] + 1); t->bn_var[l] = tensor(f, k);
    }
    t->head_w = tensor(f, "conv.21.weight");
    t->head_b = tensor(f, "conv.21.bias");
}

int main(int argc, char **argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s policy.bkw value.bkw\n", argv[0]); return 2; }
    const unsigned char *pf = slurp(argv[1]), *vf = slurp(argv[2]);
    bk_policy_weights pw;
    bk_value_weights vw;
    fill_trunk(&pw.trunk, pf);
    fill_trunk(&vw.trunk, vf);
    vw.head.bn_w = tensor(vf, "bn.weight"); vw.head.bn_b = tensor(vf, "bn.bias");
    vw.head.bn_mean = tensor(vf, "bn.running_mean"); vw.head.bn_var = tensor(vf, "bn.running_var");
    vw.head.lin1_w = tensor(vf, "lin1.weight"); vw.head.lin1_b = tensor(vf, "lin1.bias");
    vw.head.lin_bn_w = tensor(vf, "lin_bn.weight"); vw.head.lin_bn_b = tensor(vf, "lin_bn.bias");
    vw.head.lin_bn_mean = tensor(vf, "lin_bn.running_mean"); vw.head.lin_bn_var = tensor(vf, "lin_bn.running_var");
    vw.head.lin2_w = tensor(vf, "lin2.weight"); vw.head.lin2_b = tensor(vf, "lin2.bias");

    bk_engine *e = NULL;
    int rc = bk_engine_create(&pw, &vw, 0, 8, &e);
    if (rc) { fprintf(stderr, "bk_engine_create: %d %s\n", rc, bk_last_error(NULL)); return 1; }

    bk_pos pos;                       /* the empty board, black to move */
    bk_pos_init(&pos);
    float feats[2187], probs[81], value;
    bk_pos_features_f32(&pos, feats, 0);
    rc = bk_eval(e, feats, 1, BK_WANT_PROBS | BK_WANT_VALUE, NULL, probs, &value);
    if (rc) { fprintf(stderr, "bk_eval: %d %s\n", rc, bk_last_error(e)); return 1; }
    int best = 0;
    for (int i = 1; i < 81; ++i) if (probs[i] > probs[best]) best = i;
    printf("best move %c%d (index %d) p=%.6f value=%.6f\n", "ABCDEFGHJ"[best % 9], best / 9 + 1, best, probs[best], value);
    bk_engine_destroy(e);
    return 0;
}
