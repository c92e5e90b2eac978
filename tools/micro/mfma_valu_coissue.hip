// Microbenchmark: can a SIMD's packed-FP32 vector ALU work while its matrix pipe runs exact-fp32 MFMAs of ANOTHER wave?
// One 512-thread workgroup per CU: waves 0..3 (one per SIMD) issue v_mfma_f32_16x16x4_f32 back to back, waves 4..7 (their SIMD
// partners) issue v_pk_fma_f32 back to back.  Three runs: MFMA waves alone, VALU waves alone, both.  If the two pipes were
// independent for fp32, "both" would take as long as the slower one alone; if the fp32 MFMA executes on the vector ALU's
// multipliers, the times add.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_coissue.hip -o /tmp/coissue && /tmp/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Second experiment, closer to the leaf kernel: TWO MFMA waves per SIMD (waves 0..7, as the kernel runs them) and a THIRD wave
// per SIMD (waves 8..11) on v_pk_fma_f32 whose operands come from LDS (one ds_read_b128 per two v_pk_fma_f32).
__global__ void __launch_bounds__(768) k3(float* out, unsigned long long* cyc, int iters, int mode) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16384; i += 768) reinterpret_cast<float*>(lds)[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    unsigned long long t0 = 0, t1 = 0;
    float res = 0.f;
    if (wave < 8) {
        if (mode & 1) {
            f32x4 acc[16];
            for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
            const float a = lane * 0.001f, b = lane * 0.002f - 1.f;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
            for (int it = 0; it < iters; ++it) {
                // what the leaf kernel's loop also issues per 16 MFMAs: LDS fragment reads (4 x b128)
                f32x4 v0 = *reinterpret_cast<const f32x4*>(lds + lane * 528 % 60000 + (it & 3) * 16);
                f32x4 v1 = *reinterpret_cast<const f32x4*>(lds + lane * 528 % 60000 + 64 + (it & 3) * 16);
                asm volatile("" ::"v"(v0), "v"(v1));
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            }
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
            for (int i = 0; i < 16; ++i) res += acc[i][0] + acc[i][3];
        }
    } else {
        if (mode & 2) {
            f32x2 acc[16];
            for (int i = 0; i < 16; ++i) acc[i] = f32x2{0.f, 1.f};
            const char* lp = lds + lane * 16;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(lp + ((it * 8 + i / 2) & 31) * 1024);
                    const f32x2 wa = f32x2{w[0], w[1]}, wb = f32x2{w[2], w[3]};
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(wa), "v"(wb));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i + 1]) : "v"(wb), "v"(wa));
                }
            }
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
            for (int i = 0; i < 16; ++i) res += acc[i][0] + acc[i][1];
        }
    }
    out[blockIdx.x * 768 + threadIdx.x] = res;
    if (lane == 0) cyc[blockIdx.x * 12 + wave] = t1 - t0;
}

__global__ void __launch_bounds__(512) k(float* out, unsigned long long* cyc, int iters, int mode) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool mfma_wave = wave < 4;
    unsigned long long t0 = 0, t1 = 0;
    float res = 0.f;
    if (mfma_wave) {
        if (mode & 1) {
            f32x4 acc[16];
            for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
            const float a = lane * 0.001f, b = lane * 0.002f - 1.f;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
            for (int i = 0; i < 16; ++i) res += acc[i][0] + acc[i][3];
        }
    } else {
        if (mode & 2) {
            f32x2 acc[16];
            for (int i = 0; i < 16; ++i) acc[i] = f32x2{0.f, 1.f};
            f32x2 a = f32x2{lane * 0.001f, 1.0001f}, b = f32x2{0.999f, lane * 0.002f};
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
            for (int i = 0; i < 16; ++i) res += acc[i][0] + acc[i][1];
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = res;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc;
    const int blocks = 256, iters = 4096;
    hipMalloc(&out, blocks * 512 * sizeof(float));
    hipMalloc(&cyc, blocks * 8 * sizeof(unsigned long long));
    unsigned long long h[256 * 8];
    const char* names[4] = {"", "MFMA waves alone", "VALU waves alone", "both together"};
    for (int mode = 1; mode <= 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, out, cyc, iters, mode);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
        double m = 0, v = 0;
        for (int b = 0; b < blocks; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += (double)h[b * 8 + w];
        m /= blocks * 4; v /= blocks * 4;
        // s_memtime counts at a constant 100 MHz: report ticks per instruction relative to each other
        printf("%-18s  MFMA wave: %8.1f ticks per 16 MFMA (16x16x4: 32 cycles each at the fp32 rate)   VALU wave: %8.1f ticks per 16 v_pk_fma_f32 (4 cycles each)\n",
               names[mode], m / iters, v / iters);
    }
    printf("\n3 waves per SIMD: two MFMA waves (with LDS fragment reads) + one v_pk_fma_f32 wave fed from LDS\n");
    float* out3; unsigned long long* cyc3;
    hipMalloc(&out3, blocks * 768 * sizeof(float));
    hipMalloc(&cyc3, blocks * 12 * sizeof(unsigned long long));
    static unsigned long long h3[256 * 12];
    for (int mode = 1; mode <= 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k3, dim3(blocks), dim3(768), 65536, 0, out3, cyc3, iters, mode);
            hipDeviceSynchronize();
        }
        hipMemcpy(h3, cyc3, sizeof h3, hipMemcpyDeviceToHost);
        double m = 0, v = 0;
        for (int b = 0; b < blocks; ++b) for (int w = 0; w < 12; ++w) (w < 8 ? m : v) += (double)h3[b * 12 + w];
        m /= blocks * 8; v /= blocks * 4;
        printf("%-18s  each MFMA wave: %8.1f ticks per 16 MFMA (two waves share a pipe: 1024 when both run at the fp32 rate)   VALU wave: %8.1f ticks per 16 v_pk_fma_f32\n",
               names[mode], m / iters, v / iters);
    }
    return 0;
}
