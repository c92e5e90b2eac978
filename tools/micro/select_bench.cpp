// Microbenchmark behind the node layout of csrc/bk_tree.cpp: one PUCT select over 72 children (mcts.py:219-234) with
// 232-byte nodes (position inside) and two divisions per child, with 48-byte nodes, with the child's average cached at
// backprop, and with packed divisions.   g++ -O2 -std=c++17 -ffp-contract=off select_bench.cpp && ./a.out
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
#include <immintrin.h>
struct Pos { char b[192]; };
struct Fat { Pos pos; double V = 0; int N = 0; float value = 0; uint8_t f[4]; int ko = 0, nk = 0, po = -1, mv = 0; };
struct Slim { double V = 0; double avg = 0; int N = 0; int mv = 0; };
template <class T> int sel_base(const std::vector<T>& nodes, const int* kids, int n, const double* prior, double c) {
    long total = 0;
    for (int i = 0; i < n; ++i) total += nodes[kids[i]].N;
    if (!total) total = 1;
    const double sq = std::sqrt((double)total);
    int best = -1; double bs = 0;
    for (int i = 0; i < n; ++i) {
        const T& k = nodes[kids[i]];
        const double avg = k.N == 0 ? 0.0 : k.V / (double)k.N;
        const double s = -avg + (c * prior[k.mv] * sq / (double)(1 + k.N));
        if (best < 0 || s > bs) { best = kids[i]; bs = s; }
    }
    return best;
}
int sel_avg(const std::vector<Slim>& nodes, const int* kids, int n, const double* prior, double c) {
    long total = 0;
    for (int i = 0; i < n; ++i) total += nodes[kids[i]].N;
    if (!total) total = 1;
    const double sq = std::sqrt((double)total);
    int best = -1; double bs = 0;
    for (int i = 0; i < n; ++i) {
        const Slim& k = nodes[kids[i]];
        const double s = -k.avg + (c * prior[k.mv] * sq / (double)(1 + k.N));
        if (best < 0 || s > bs) { best = kids[i]; bs = s; }
    }
    return best;
}
int sel_sse(const std::vector<Slim>& nodes, const int* kids, int n, const double* prior, double c) {
    long total = 0;
    for (int i = 0; i < n; ++i) total += nodes[kids[i]].N;
    if (!total) total = 1;
    const double sq = std::sqrt((double)total);
    int best = -1; double bs = 0;
    int i = 0;
    for (; i + 1 < n; i += 2) {
        const Slim& k0 = nodes[kids[i]]; const Slim& k1 = nodes[kids[i + 1]];
        __m128d num = _mm_set_pd(c * prior[k1.mv] * sq, c * prior[k0.mv] * sq);
        __m128d den = _mm_set_pd((double)(1 + k1.N), (double)(1 + k0.N));
        __m128d q = _mm_div_pd(num, den);
        double u[2]; _mm_storeu_pd(u, q);
        const double s0 = -k0.avg + u[0], s1 = -k1.avg + u[1];
        if (best < 0 || s0 > bs) { best = kids[i]; bs = s0; }
        if (s1 > bs) { best = kids[i + 1]; bs = s1; }
    }
    for (; i < n; ++i) {
        const Slim& k = nodes[kids[i]];
        const double s = -k.avg + (c * prior[k.mv] * sq / (double)(1 + k.N));
        if (best < 0 || s > bs) { best = kids[i]; bs = s; }
    }
    return best;
}
// gather (N, avg, prior) of the children into arrays, scores four at a time (vdivpd ymm), first maximum in a scalar pass:
// the same IEEE operations per child in the same order as sel_avg, so the same bits and the same choice.  Round 4, on the GPU
// box's EPYC 9575F: cached avg (shipped) 91 ns per 72-child select = 1.26 ns per child, divpd xmm 88 ns, this 111 ns -- the
// scalar loop already runs at the divider's pace and the staging costs more than the packed division saves.  Not adopted.
__attribute__((target("avx2"))) int sel_avx2(const std::vector<Slim>& nodes, const int* kids, int n, const double* prior, double c) {
    alignas(32) int nn[96];
    alignas(32) double av[96], pr[96], sc[96];
    long total = 0;
    for (int i = 0; i < n; ++i) {
        const Slim& k = nodes[kids[i]];
        nn[i] = 1 + k.N; av[i] = k.avg; pr[i] = prior[k.mv];
        total += k.N;
    }
    for (int i = n; i < ((n + 3) & ~3); ++i) { nn[i] = 1; av[i] = 0; pr[i] = 0; }
    if (!total) total = 1;
    const __m256d sq = _mm256_set1_pd(std::sqrt((double)total)), cc = _mm256_set1_pd(c);
    for (int i = 0; i < n; i += 4) {
        const __m256d num = _mm256_mul_pd(_mm256_mul_pd(cc, _mm256_load_pd(pr + i)), sq);
        const __m256d den = _mm256_cvtepi32_pd(_mm_load_si128((const __m128i*)(nn + i)));
        _mm256_store_pd(sc + i, _mm256_sub_pd(_mm256_div_pd(num, den), _mm256_load_pd(av + i)));
    }
    int best = -1; double bs = 0;
    for (int i = 0; i < n; ++i)
        if (best < 0 || sc[i] > bs) { best = kids[i]; bs = sc[i]; }
    return best;
}
// an unvisited child (N = 0, avg = +0.0) scores -0.0 + c p sq / 1.0 = c p sq exactly: no division for it.  In a real tree ~70 of a
// node's ~75 children are unvisited (main() below draws N accordingly when run with an argument).  EPYC 9575F, that mix: 90.8 ->
// 83.4 ns per select; in the tree 0.385 -> 0.354 ms of search per 1600-rollout move.  Adopted (csrc/bk_tree.cpp, select).
int sel_zero(const std::vector<Slim>& nodes, const int* kids, int n, const double* prior, double c) {
    long total = 0;
    for (int i = 0; i < n; ++i) total += nodes[kids[i]].N;
    if (!total) total = 1;
    const double sq = std::sqrt((double)total);
    int best = -1; double bs = 0;
    for (int i = 0; i < n; ++i) {
        const Slim& k = nodes[kids[i]];
        const double e = c * prior[k.mv] * sq;
        const double s = k.N == 0 ? e : -k.avg + (e / (double)(1 + k.N));
        if (best < 0 || s > bs) { best = kids[i]; bs = s; }
    }
    return best;
}
int main(int argc, char**) {
    const int NK = 72, NPAR = 200;
    std::mt19937 rng(1);
    std::vector<Fat> fat(NPAR * NK + 1); std::vector<Slim> slim(NPAR * NK + 1);
    std::vector<int> kids(NPAR * NK); std::vector<double> prior(81);
    for (auto& p : prior) p = (rng() % 1000) / 40000.0;
    for (int i = 0; i < NPAR * NK; ++i) {
        kids[i] = i + 1;
        int N = rng() % 50; if (argc > 1 && rng() % 72 >= 5) N = 0;   // with an argument: 5 of 72 children visited, as in a search
        double V = ((int)(rng() % 2000) - 1000) / 1000.0 * N;
        fat[i + 1].N = N; fat[i + 1].V = V; fat[i + 1].mv = i % 81;
        slim[i + 1].N = N; slim[i + 1].V = V; slim[i + 1].mv = i % 81; slim[i + 1].avg = N ? V / N : 0;
    }
    auto run = [&](const char* name, auto f) {
        long acc = 0; const int REP = 20000;
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < REP; ++r) { int p = r % NPAR; acc += f(&kids[p * NK]); }
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%-28s %7.1f ns per select (%d kids)  %.2f ns per child  [%ld]\n", name, 1e9 * dt / REP, NK, 1e9 * dt / REP / NK, acc);
    };
    for (int rep = 0; rep < 2; ++rep) {
        run("fat nodes, 2 divisions", [&](const int* k) { return sel_base(fat, k, NK, prior.data(), 4.0); });
        run("slim nodes, 2 divisions", [&](const int* k) { return sel_base(slim, k, NK, prior.data(), 4.0); });
        run("slim nodes, cached avg", [&](const int* k) { return sel_avg(slim, k, NK, prior.data(), 4.0); });
        run("slim, cached avg, divpd", [&](const int* k) { return sel_sse(slim, k, NK, prior.data(), 4.0); });
        run("slim, cached avg, N=0 shortcut", [&](const int* k) { return sel_zero(slim, k, NK, prior.data(), 4.0); });
        run("slim, cached avg, avx2", [&](const int* k) { return sel_avx2(slim, k, NK, prior.data(), 4.0); });
    }
}
