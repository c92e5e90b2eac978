// Two 64-game pools (400 rollouts per move, fake evaluator) driven by two caller threads at once against the same two pools one
// after the other: the team (bk_tree.cpp) serves both callers' jobs side by side.  VERDICT r3 item 6.
//   g++ -O2 -std=c++17 -pthread tools/micro/two_pools.cpp -Lbokego_amd -lbkgo -Wl,-rpath,$PWD/bokego_amd -o /tmp/two_pools && /tmp/two_pools 4
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../../include/bokego_go.h"
#include "../../include/bokego_tree.h"

static double generation(int G, int threads, int seed0) {
    bk_search_params prm;
    bk_search_params_default(&prm);
    prm.rollouts = 400; prm.expand_thresh = 100; prm.noise_weight = 0.25f; prm.sample_plies = 4; prm.max_turns = 60; prm.prune = 1; prm.eager_top = 4;
    std::vector<uint64_t> seeds(G);
    for (int g = 0; g < G; ++g) seeds[g] = seed0 + g;
    bk_pool* p = bk_pool_create(G, &prm, seeds.data(), threads);
    constexpr int CAP = 8192;
    std::vector<bk_pos> recs(CAP);
    std::vector<float> probs((size_t)CAP * 81), values(CAP);
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        int npol = 0;
        const int n = bk_pool_collect_pos(p, recs.data(), CAP, &npol);
        if (!n) break;
        for (int r = 0; r < npol; ++r) {
            const unsigned char* b = (const unsigned char*)&recs[r];
            float sum = 0;
            for (int k = 0; k < 81; ++k) sum += probs[(size_t)r * 81 + k] = 1.f + (float)((b[k] * 7 + k * 13 + r) % 17);
            for (int k = 0; k < 81; ++k) probs[(size_t)r * 81 + k] /= sum;
        }
        for (int r = 0; r < n; ++r) {
            const unsigned char* b = (const unsigned char*)&recs[r];
            unsigned h = 0;
            for (int k = 0; k < 96; ++k) h = h * 31 + b[k];
            values[r] = (float)(h % 2001) / 1000.f - 1.f;
        }
        bk_pool_deliver(p, probs.data(), values.data());
    }
    const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    bk_pool_destroy(p);
    return t;
}

int main(int argc, char** argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 4, reps = argc > 2 ? atoi(argv[2]) : 3;
    generation(8, threads, 1);
    for (int r = 0; r < reps; ++r) {
        const double a = generation(64, threads, 100), b = generation(64, threads, 900);
        const auto t0 = std::chrono::steady_clock::now();
        std::thread x([&] { generation(64, threads, 100); }), y([&] { generation(64, threads, 900); });
        x.join();
        y.join();
        const double par = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("threads per pool %d: one after the other %.3f + %.3f = %.3f s, two callers at once %.3f s  ratio %.2f\n", threads, a, b, a + b, par, par / (a + b));
    }
    return 0;
}
