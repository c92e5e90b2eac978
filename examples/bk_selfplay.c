/*
 * bk_selfplay.c -- BASELINE config 4 without Python: a generation of self-play games driven from C through the
 * four C ABIs (include/bokego_amd.h engine, bokego_go.h records, bokego_tree.h game pools, bokego_comm.h
 * all-reduce).  The step loop is bk_pools_run (include/bokego_tree.h) with the engine's own evaluator
 * (bk_engine_evaluator): lock-step pools rotate through one engine, so the host advances one pool's trees while the
 * GPU evaluates the others' leaves -- what bokego_amd/selfplay.py does by default.
 *
 *   make -C bokego_amd/csrc examples
 *   ./examples/bk_selfplay tests/golden/policy_19.bkw tests/golden/value_synth.bkw [games=512] [rollouts=400]
 *   multi-GPU (one process per GPU): BK_RANK=r BK_WORLD=n BK_COMM_ID_FILE=/shared/path ./examples/bk_selfplay ...
 *
 * Games are assigned gid % world and seeded by 20260 + gid, as in the Python driver, and the priors are normalised
 * by the same bk_normalise_rows: the two drivers play the same games move for move (the checksum below is over every
 * move of every game; tests/test_gpu_comm.py compares the statistics with the Python driver's).
 */
#include <time.h>
#include <unistd.h>

#include "bkw_load.h"
#include "bokego_comm.h"
#include "bokego_go.h"
#include "bokego_tree.h"

#define CAP 8192
#define NPOOLS 3

typedef struct {
    bk_pool *pool;
    int n_games;
} slot_t;

static double now(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + 1e-9 * t.tv_nsec;
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s policy.bkw value.bkw [games] [rollouts]\n", argv[0]); return 2; }
    const int n_games = argc > 3 ? atoi(argv[3]) : 512, rollouts = argc > 4 ? atoi(argv[4]) : 400;
    const int rank = getenv("BK_RANK") ? atoi(getenv("BK_RANK")) : 0, world = getenv("BK_WORLD") ? atoi(getenv("BK_WORLD")) : 1;
    const unsigned char *pf = slurp(argv[1]), *vf = slurp(argv[2]);
    bk_policy_weights pw;
    bk_value_weights vw;
    fill_trunk(&pw.trunk, pf);
    fill_trunk(&vw.trunk, vf);
    fill_value_head(&vw.head, vf);
    const int device = rank % (bk_device_count() > 0 ? bk_device_count() : 1);
    bk_engine *e = NULL;
    if (bk_engine_create(&pw, &vw, device, CAP, &e)) { fprintf(stderr, "bk_engine_create: %s\n", bk_last_error(NULL)); return 1; }

    bk_search_params prm;
    bk_search_params_default(&prm);
    prm.rollouts = rollouts;
    prm.noise_weight = 0.25f;
    prm.sample_plies = 8;
    prm.prune = 1;
    /* children evaluated at an expansion: the best by prior (the search visits them in that order and rarely gets past the
     * first few); what bokego_amd/selfplay.py uses */
    {
        const char *pe = getenv("BK_PRECISION");
        prm.eager_top = (pe && strcmp(pe, "f16x2") == 0) ? 6 : 4;
    }
    int mine_n = 0;
    for (int g = rank; g < n_games; g += world) ++mine_n;
    const int f32 = !(getenv("BK_PRECISION") && strcmp(getenv("BK_PRECISION"), "f16x2") == 0);
    if (f32 && mine_n >= 192) prm.eager_top = 2;   /* fp32 from 192 games per rank: two children (selfplay.py, EAGER_TOP) */

    /* a pool of 22..42 games on the fp32 engine asks for 81..128 tasks per step -- the 2-CUs-per-board launch, whose time does
     * not depend on the size: a leaf that reaches 70 visits sends its policy row and then its best children along with requests
     * that go out anyway (evaluation ahead of expansion, the same search), and the batch is held to 128 tasks
     * (selfplay.small_shard_defaults; profiles/r05_spec_probe.txt: 64 games per rank 0.181 -> 0.171 s) */
    const int per_pool = (mine_n + 1) / 2;
    const int small_shard = f32 && prm.eager_top > 2 && 3 * per_pool > 64 && 3 * per_pool <= 128;
    if (small_shard) { prm.speculate = 70; prm.speculate_rows = 8; }

    /* this rank's games, dealt to the pools round-robin */
    slot_t s[NPOOLS];
    uint64_t *seeds = malloc(sizeof(uint64_t) * (n_games + 1));
    int mine = 0;
    for (int g = rank; g < n_games; g += world) seeds[mine++] = 20260 + (uint64_t)g;
    /* two pools: the host advances one while the GPU evaluates the other's batch */
    const int npools = mine >= 2 ? 2 : 1;
    const long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    for (int i = 0; i < npools; ++i) {
        uint64_t *ps = malloc(sizeof(uint64_t) * (mine / npools + 1));
        int k = 0;
        for (int j = i; j < mine; j += npools) ps[k++] = seeds[j];
        /* the pools' worker threads spin between steps: 12 of a 16-CPU share, and no more than one per 8 games */
        int threads = ncpu > 8 ? (ncpu - 4 < 12 ? (int)ncpu - 4 : 12) : (int)ncpu / 2;
        if (threads > k / 8) threads = k / 8;
        s[i].pool = bk_pool_create(k, &prm, ps, threads < 1 ? 1 : threads);
        /* fp32 engine: a launch costs whole rounds of 3-board workgroups (256 CUs x 3 tasks); hold the batches to the whole
         * number of rounds nearest to what the pool asks for (~3 tasks per game and step), minus the per-net rounding */
        {
            const char *pe2 = getenv("BK_PRECISION");
            if (!(pe2 && strcmp(pe2, "f16x2") == 0)) {
                if (small_shard) {
                    bk_pool_set_task_cap(s[i].pool, 128);
                } else if (prm.eager_top <= 2) {   /* ~2 tasks per game and step: whole rounds of 1-, 2- or 3-board workgroups */
                    int rounds = (2 * k + 128) / 256;
                    bk_pool_set_task_cap(s[i].pool, 256 * (rounds < 1 ? 1 : rounds) - 4);
                } else if (3 * k > 128 && 3 * k <= 192) {
                    bk_pool_set_task_cap(s[i].pool, 188);   /* within the range of the three-boards-on-four-CUs launch */
                } else {
                    int rounds = (3 * k + 384) / 768;
                    bk_pool_set_task_cap(s[i].pool, 768 * (rounds < 1 ? 1 : rounds) - 4);
                }
                bk_pool_set_dedup(s[i].pool, 1);   /* equal records of one batch travel once: +4 % where the GPU is the limit */
            }
        }
        s[i].n_games = k;
        free(ps);
    }

    const double t0 = now();
    bk_evaluator ev;
    bk_run_info info;
    bk_pool *pools[NPOOLS];
    for (int i = 0; i < npools; ++i) pools[i] = s[i].pool;
    if (bk_engine_evaluator(e, &ev)) { fprintf(stderr, "bk_engine_evaluator: %s\n", bk_last_error(e)); return 1; }
    if (bk_pools_run(pools, npools, &ev, CAP, &info)) { fprintf(stderr, "bk_pools_run: %s\n", bk_last_error(e)); return 1; }
    const long steps = (long)info.steps, positions = (long)info.rows;
    const double secs = now() - t0;

    /* statistics vector, as bokego_amd/selfplay.py:pool_stats: 11 scalars (the last three: sum / sum of magnitudes / count of
     * the root's mean backed-up value at every move) + first-move histogram + root-child visit histogram */
    enum { NSCAL = 11, NST = NSCAL + 2 * 81 };
    double st[NST];
    memset(st, 0, sizeof st);
    unsigned long long check = 1469598103934665603ull;    /* FNV-1a over every move of every game (determinism check) */
    for (int i = 0; i < npools; ++i)
        for (int g = 0; g < s[i].n_games; ++g) {
            bk_game_info gi;
            int16_t mv[128];
            bk_pool_game_info(s[i].pool, g, &gi);
            const int nm = bk_pool_game_moves(s[i].pool, g, mv, 128);
            st[0] += 1; st[1] += gi.score > 0; st[2] += gi.score <= 0; st[3] += gi.n_moves; st[4] += gi.score;
            st[5] += (double)gi.n_value_evals; st[6] += (double)gi.n_policy_evals; st[7] += (double)gi.n_requests;
            bk_game_stats gs;
            bk_pool_game_stats(s[i].pool, g, &gs);
            st[8] += gs.sum_root_value; st[9] += gs.sum_abs_root_value; st[10] += (double)gs.n_root_values;
            if (nm > 0 && mv[0] >= 0) st[NSCAL + mv[0]] += 1;
            for (int k = 0; k < 81; ++k) st[NSCAL + 81 + k] += (double)gs.root_visits[k];
            for (int k = 0; k < nm; ++k) check = (check ^ (unsigned short)mv[k]) * 1099511628211ull;
        }
    double reduce_ms = 0, reduce_wait_ms = 0;
    int rccl = 0;
    char pci[32] = "";
    if (world > 1) {
        const char *idf = getenv("BK_COMM_ID_FILE");
        uint8_t id[BK_COMM_ID_BYTES];
        if (!idf) { fprintf(stderr, "BK_WORLD > 1 needs BK_COMM_ID_FILE\n"); return 1; }
        /* the file is [16-byte job tag | id]: a file left by another launch (other BK_COMM_JOB) is not ours */
        char tag[16] = {0}, got[16];
        const char *job = getenv("BK_COMM_JOB");
        if (job) strncpy(tag, job, sizeof tag);
        if (rank == 0) {
            if (bk_comm_unique_id(id)) { fprintf(stderr, "%s\n", bk_comm_last_error()); return 1; }
            char tmp[4096];
            snprintf(tmp, sizeof tmp, "%s.tmp", idf);
            remove(idf);
            FILE *f = fopen(tmp, "wb"); fwrite(tag, 1, sizeof tag, f); fwrite(id, 1, sizeof id, f); fclose(f); rename(tmp, idf);
        } else {
            for (;;) {
                FILE *f = fopen(idf, "rb");
                if (f) {
                    const int ok = fread(got, 1, sizeof got, f) == sizeof got && fread(id, 1, sizeof id, f) == sizeof id &&
                                   !memcmp(got, tag, sizeof tag);
                    fclose(f);
                    if (ok) break;
                }
                usleep(10000);
            }
        }
        bk_comm *c = NULL;
        if (bk_comm_init(rank, world, id, device, &c)) { fprintf(stderr, "bk_comm_init: %s\n", bk_comm_last_error()); return 1; }
        if (rank == 0) remove(idf);   /* every rank has joined: the next run must not find this id */
        /* the ranks finish their games at different times: wait for the slowest one at a barrier first (wait_ms: the skew), then time
         * the all-reduce alone (allreduce_ms: the collective) */
        const double t1 = now();
        if (bk_comm_barrier(c)) { fprintf(stderr, "barrier: %s\n", bk_comm_last_error()); return 1; }
        const double t2 = now();
        if (bk_comm_allreduce_sum_f64(c, st, NST)) { fprintf(stderr, "allreduce: %s\n", bk_comm_last_error()); return 1; }
        reduce_ms = (now() - t2) * 1e3;
        reduce_wait_ms = (t2 - t1) * 1e3;
        rccl = bk_comm_rccl_version();
        if (bk_comm_device_pci(c, pci, (int)sizeof pci)) pci[0] = 0;
        bk_comm_destroy(c);
    }
    double host[3] = {0, 0, 0};       /* seconds the pools spent advancing games / writing request rows / taking deliveries */
    for (int i = 0; i < npools; ++i) {
        double t[3];
        bk_pool_phase_seconds(s[i].pool, t);
        for (int k = 0; k < 3; ++k) host[k] += t[k];
    }
    double visit_sum = 0;
    for (int k = 0; k < 81; ++k) visit_sum += st[NSCAL + 81 + k];
    if (rank == 0)
        printf("{\"games\": %.0f, \"local_games\": %d, \"seconds\": %.4f, \"local_games_per_min\": %.0f, \"steps\": %ld, \"mean_batch\": %.0f, "
               "\"plies\": %.0f, \"black_wins\": %.0f, \"value_evals\": %.0f, \"policy_evals\": %.0f, \"allreduce_ms\": %.3f, \"allreduce_wait_ms\": %.3f, \"rccl_version\": %d, \"rank0_pci\": \"%s\", "
               "\"sum_root_value\": %.9f, \"sum_abs_root_value\": %.9f, \"n_root_values\": %.0f, \"root_visit_hist_sum\": %.0f, \"wait_s\": %.3f, "
               "\"host_advance_s\": %.3f, \"host_emit_s\": %.3f, \"host_deliver_s\": %.3f, \"moves_checksum\": \"%016llx\"}\n",
               st[0], mine, secs, mine / secs * 60, steps, steps ? (double)positions / steps : 0.0, st[3], st[1], st[5], st[6], reduce_ms, reduce_wait_ms, rccl, pci,
               st[8], st[9], st[10], visit_sum, info.wait_seconds,
               host[0], host[1], host[2], check);
    for (int i = 0; i < npools; ++i) bk_pool_destroy(s[i].pool);
    bk_engine_destroy(e);
    return 0;
}
