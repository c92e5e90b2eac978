/*
 * bkw_load.h -- minimal BKW1 reader for the C examples: fills the C ABI's weight structs with pointers into
 * a file image (bokego_amd/bkw.py documents the format).
 */
#ifndef BKW_LOAD_H
#define BKW_LOAD_H

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bokego_amd.h"

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

static void fill_value_head(bk_value_head_weights *h, const unsigned char *vf) {
    h->bn_w = tensor(vf, "bn.weight"); h->bn_b = tensor(vf, "bn.bias");
    h->bn_mean = tensor(vf, "bn.running_mean"); h->bn_var = tensor(vf, "bn.running_var");
    h->lin1_w = tensor(vf, "lin1.weight"); h->lin1_b = tensor(vf, "lin1.bias");
    h->lin_bn_w = tensor(vf, "lin_bn.weight"); h->lin_bn_b = tensor(vf, "lin_bn.bias");
    h->lin_bn_mean = tensor(vf, "lin_bn.running_mean"); h->lin_bn_var = tensor(vf, "lin_bn.running_var");
    h->lin2_w = tensor(vf, "lin2.weight"); h->lin2_b = tensor(vf, "lin2.bias");
}

#endif
