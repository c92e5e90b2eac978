// Microbenchmark: back-to-back issue cost of the two exact-fp32 MFMA shapes on one wave per SIMD, operands in registers.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int NREAD>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = lane * 0.001f + i; b[i] = lane * 0.002f - i; }
    const char* lp = lds + lane * 528;
    unsigned long long t0, t1;
    if constexpr (SHAPE == 16) {
        f32x4 acc[32];
        for (int i = 0; i < 32; ++i) acc[i] = f32x4{0, 0, 0, 0};
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[(i + j) & 3], acc[i], 0, 0, 0);
                if (j == 0) {
#pragma unroll
                    for (int r = 0; r < NREAD; ++r) { f32x4 v = *reinterpret_cast<const f32x4*>(lp + r * 64 + (it & 1) * 16); asm volatile("" ::"v"(v)); }
                }
            }
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        float s = 0;
        for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][3];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else if constexpr (SHAPE == 4) {
        // v_mfma_f32_4x4x1_16B_f32: 16 independent 4x4x1 blocks per instruction (256 MAC): the fine-grained shape a 4-position
        // tile granularity would use; independent accumulators, NREAD LDS reads per 128 instructions
        f32x4 acc[32];
        for (int i = 0; i < 32; ++i) acc[i] = f32x4{0, 0, 0, 0};
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[j], b[(i + j) & 3], acc[i], 0, 0, 0);
                if (j == 0) {
#pragma unroll
                    for (int r = 0; r < NREAD; ++r) { f32x4 v = *reinterpret_cast<const f32x4*>(lp + r * 64 + (it & 1) * 16); asm volatile("" ::"v"(v)); }
                }
            }
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        float s = 0;
        for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][3];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else {
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j & 3], b[(i + j) & 3], acc[i], 0, 0, 0);
                if (j == 0) {
#pragma unroll
                    for (int r = 0; r < NREAD; ++r) { f32x4 v = *reinterpret_cast<const f32x4*>(lp + r * 64 + (it & 1) * 16); asm volatile("" ::"v"(v)); }
                }
            }
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        float s = 0;
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int SHAPE, int NREAD>
void run(const char* name, int mfma_per_iter) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 4 * 8);
    const int iters = 2000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<SHAPE, NREAD>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<SHAPE, NREAD>), dim3(256), dim3(256), 150 * 1024, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[1024];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 1024; ++i) s += (double)h[i];
    // s_memtime ticks are shader-clock cycles here (16x16x4 comes out at its 32)
    printf("%-36s cycles per MFMA %.2f\n", name, s / 1024 / iters / mfma_per_iter);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<16, 0>("16x16x4, no LDS reads", 128);
    run<16, 8>("16x16x4, 8 ds_read/128", 128);
    run<32, 0>("32x32x2, no LDS reads", 64);
    run<32, 4>("32x32x2, 4 ds_read/64", 64);
    run<4, 0>("4x4x1 (16 blocks), no LDS reads", 128);
    run<4, 32>("4x4x1 (16 blocks), 32 ds_read/128", 128);
    return 0;
}
