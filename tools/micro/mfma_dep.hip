// Microbenchmark: v_mfma_f32_16x16x4_f32 issued back to back when the same accumulator comes round again after DIST instructions
// (a rt-major order of the conv loop would reuse an accumulator at distance 2; the shipped j-major order at distance >= 5).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_dep.hip -o /tmp/mfma_dep && /tmp/mfma_dep
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DIST, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k(float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = lane * 0.001f + i; b[i] = lane * 0.002f - i; }
    f32x4 acc[DIST];
    for (int i = 0; i < DIST; ++i) acc[i] = f32x4{0, 0, 0, 0};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 64; ++i) acc[i % DIST] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i & 3], b[(i >> 2) & 3], acc[i % DIST], 0, 0, 0);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0;
    for (int i = 0; i < DIST; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * WAVES + (threadIdx.x >> 6)] = t1 - t0;
}

template <int DIST, int WAVES>
void run() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 64 * WAVES * 4); hipMalloc(&cyc, 256 * WAVES * 8);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<DIST, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[256 * 8];
    hipMemcpy(h, cyc, 256 * WAVES * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 256 * WAVES; ++i) s += (double)h[i];
    // per SIMD: WAVES / 4 waves share the matrix pipe
    printf("same accumulator every %2d MFMAs, %d wave(s) per SIMD: %.2f cycles per MFMA of a wave, %.2f per MFMA of the SIMD\n", DIST, WAVES / 4,
           s / (256 * WAVES) / iters / 64, s / (256 * WAVES) / iters / 64 / (WAVES / 4));
    hipFree(out); hipFree(cyc);
}
int main() {
    run<1, 4>(); run<2, 4>(); run<4, 4>(); run<8, 4>(); run<16, 4>();
    run<1, 8>(); run<2, 8>(); run<4, 8>(); run<16, 8>();
    return 0;
}
