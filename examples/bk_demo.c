/*
 * bk_demo.c -- the C ABI without Python: load two BKW1 weight files, encode the empty board with
 * libbkgo, evaluate it on the GPU with libbokego_amd, print the policy's best move and the value.
 *
 *   hipcc (or gcc) examples/bk_demo.c -Iinclude -Lbokego_amd -lbokego_amd -lbkgo -Wl,-rpath,$PWD/bokego_amd -o bk_demo
 *   ./bk_demo tests/golden/policy_19.bkw tests/golden/value_synth.bkw
 */
#include "bkw_load.h"
#include "bokego_go.h"

int main(int argc, char **argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s policy.bkw value.bkw\n", argv[0]); return 2; }
    const unsigned char *pf = slurp(argv[1]), *vf = slurp(argv[2]);
    bk_policy_weights pw;
    bk_value_weights vw;
    fill_trunk(&pw.trunk, pf);
    fill_trunk(&vw.trunk, vf);
    fill_value_head(&vw.head, vf);

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
