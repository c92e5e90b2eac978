// ThreadSanitizer driver of the game pools (bk_tree.cpp: worker team, lanes, emit phase): two pools of 48 games advanced in turn
// with a fake evaluator, 6 threads each, then two more pools driven by TWO caller threads at once (the team serves both callers'
// jobs side by side) and the team's selftests from two threads -- no Python, no GPU.   make -C bokego_amd/csrc tsan
// Also the gprof driver of the host side: tsan_pool [games per pool] [rollouts] [expand_thresh] [max_turns] [threads]
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/bokego_go.h"
#include "../../include/bokego_tree.h"

#include <cstdlib>
constexpr int CAP = 8192;
// one collect / evaluate (fake, a pure function of the record) / deliver step of a pool; false when every game is finished
static bool step(bk_pool* pool, long& steps, long& rows) {
    thread_local std::vector<bk_pos> recs(CAP);
    thread_local std::vector<float> probs((size_t)CAP * 81), values(CAP);
    int npol = 0;
    const int n = bk_pool_collect_pos(pool, recs.data(), CAP, &npol);
    if (n == 0) return false;
    for (int r = 0; r < npol; ++r) {
        const unsigned char* b = reinterpret_cast<const unsigned char*>(&recs[r]);
        float sum = 0;
        for (int k = 0; k < 81; ++k) sum += probs[(size_t)r * 81 + k] = 1.f + (float)((b[k] * 7 + k * 13 + r) % 17);
        for (int k = 0; k < 81; ++k) probs[(size_t)r * 81 + k] /= sum;
    }
    for (int r = 0; r < n; ++r) {
        const unsigned char* b = reinterpret_cast<const unsigned char*>(&recs[r]);
        unsigned h = 0;
        for (int k = 0; k < 96; ++k) h = h * 31 + b[k];
        values[r] = (float)(h % 2001) / 1000.f - 1.f;
    }
    bk_pool_deliver(pool, probs.data(), values.data());
    ++steps;
    rows += n;
    return true;
}

// the same fake networks behind the bk_evaluator callbacks of the C step loop (bk_pools_run), the batch "evaluated" by ANOTHER
// thread between submit and wait -- as the GPU does for the engine's evaluator
struct AsyncEval {
    std::thread worker;
    int64_t next = 1;
};
static void fill(const bk_pos* recs, int n, int npol, float* probs, float* values) {
    for (int r = 0; r < npol; ++r) {
        const unsigned char* b = reinterpret_cast<const unsigned char*>(&recs[r]);
        for (int k = 0; k < 81; ++k) probs[(size_t)r * 81 + k] = 1.f + (float)((b[k] * 7 + k * 13 + r) % 17);   // not normalised: the loop does that
    }
    for (int r = 0; r < n; ++r) {
        const unsigned char* b = reinterpret_cast<const unsigned char*>(&recs[r]);
        unsigned h = 0;
        for (int k = 0; k < 96; ++k) h = h * 31 + b[k];
        values[r] = (float)(h % 2001) / 1000.f - 1.f;
    }
}
static int64_t ev_submit(void* ctx, const bk_pos* recs, int n, int npol, float* probs, float* values) {
    AsyncEval* e = static_cast<AsyncEval*>(ctx);
    if (e->worker.joinable()) e->worker.join();           // (one batch in flight at a time is enough here)
    e->worker = std::thread(fill, recs, n, npol, probs, values);
    return e->next++;
}
static int ev_wait(void* ctx, int64_t) {
    AsyncEval* e = static_cast<AsyncEval*>(ctx);
    if (e->worker.joinable()) e->worker.join();
    return 0;
}

int main(int argc, char** argv) {
    bk_search_params prm;
    bk_search_params_default(&prm);
    const int G = argc > 1 ? atoi(argv[1]) : 48;
    prm.rollouts = argc > 2 ? atoi(argv[2]) : 60;
    prm.expand_thresh = argc > 3 ? atoi(argv[3]) : 8;
    prm.noise_weight = 0.25f;
    prm.sample_plies = 4;
    prm.max_turns = argc > 4 ? atoi(argv[4]) : 24;
    prm.prune = 1;
    prm.eager_top = 4;
    const int threads = argc > 5 ? atoi(argv[5]) : 6;
    std::vector<uint64_t> seeds(G);
    bk_pool* pools[2];
    for (int p = 0; p < 2; ++p) {
        for (int g = 0; g < G; ++g) seeds[g] = 20260 + 2 * g + p;
        pools[p] = bk_pool_create(G, &prm, seeds.data(), threads);
        if (!pools[p]) return 2;
    }
    long steps = 0, rows = 0;
    for (bool busy = true; busy;) {
        busy = false;
        for (int p = 0; p < 2; ++p) busy = step(pools[p], steps, rows) || busy;
    }
    // two callers at once: each thread drives a pool of its own to the end
    bk_pool* par[2];
    long psteps[2] = {0, 0}, prows[2] = {0, 0};
    for (int p = 0; p < 2; ++p) {
        for (int g = 0; g < G; ++g) seeds[g] = 30260 + 2 * g + p;
        par[p] = bk_pool_create(G, &prm, seeds.data(), threads);
        if (!par[p]) return 2;
        bk_pool_set_dedup(par[p], 1);     // the de-duplicating collect: liberty refresh in the lanes, hashing on the caller
    }
    {
        std::thread a([&] { while (step(par[0], psteps[0], prows[0])) {} }), b([&] { while (step(par[1], psteps[1], prows[1])) {} });
        a.join();
        b.join();
    }
    for (int p = 0; p < 2; ++p) bk_pool_destroy(par[p]);
    std::printf("tsan_pool: two callers at once: %ld + %ld steps, %ld + %ld rows\n", psteps[0], psteps[1], prows[0], prows[1]);
    long plies = 0;
    for (int p = 0; p < 2; ++p) {
        for (int g = 0; g < G; ++g) {
            bk_game_info gi;
            bk_pool_game_info(pools[p], g, &gi);
            plies += gi.n_moves;
        }
        bk_pool_destroy(pools[p]);
    }
    std::printf("tsan_pool: %ld steps, %ld rows, %ld plies\n", steps, rows, plies);
    // the step loop in C over three pools, batches completed by another thread; then a game carried to another pool by a snapshot
    {
        bk_pool* run[3];
        for (int p = 0; p < 3; ++p) {
            for (int g = 0; g < G; ++g) seeds[g] = 40260 + 3 * g + p;
            run[p] = bk_pool_create(G, &prm, seeds.data(), threads);
            if (!run[p]) return 2;
            if (p == 1) bk_pool_set_dedup(run[p], 1);
        }
        AsyncEval ae;
        bk_evaluator ev{&ae, ev_submit, ev_wait};
        bk_run_info info;
        if (bk_pools_run(run, 3, &ev, CAP, &info)) return 3;
        std::vector<unsigned char> snap((size_t)bk_pool_snapshot(run[0], 0, nullptr, 0));
        if (bk_pool_snapshot(run[0], 0, snap.data(), (long)snap.size()) != (long)snap.size()) return 4;
        if (bk_pool_restore(run[2], 1, snap.data(), (long)snap.size())) return 5;
        bk_game_stats a, b;
        bk_pool_game_stats(run[0], 0, &a);
        bk_pool_game_stats(run[2], 1, &b);
        if (std::memcmp(&a, &b, sizeof a)) return 6;
        for (int p = 0; p < 3; ++p) bk_pool_destroy(run[p]);
        std::printf("tsan_pool: bk_pools_run: %llu steps, %llu rows; snapshot %zu bytes\n", (unsigned long long)info.steps,
                    (unsigned long long)info.rows, snap.size());
    }
    int rc = bk_team_selftest(4, 20000);
    int rc2[2] = {0, 0};
    {
        std::thread a([&] { rc2[0] = bk_team_selftest(3, 20000); }), b([&] { rc2[1] = bk_team_selftest(4, 20000); });
        a.join();
        b.join();
    }
    return rc || rc2[0] || rc2[1] || bk_team_selftest_concurrent(3, 5000);
}
