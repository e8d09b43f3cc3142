// Micro-benchmark 2: (a) does sustained fp64 MFMA throttle the clock?  (b) do fp64 VALU / fp32 VALU
// ops from ANOTHER wave on the same SIMD overlap with v_mfma_f64_16x16x4_f64?
// Build: hipcc --offload-arch=gfx950 -O3 tools/pipe_share.hip -o build/pipe_share
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

// waves 0-3 of the 512-thread block run MFMA (if mfma_on); waves 4-7 run VALU of `kind`
// kind: 0 none, 1 fp64 fma, 2 fp32 fma
__global__ void __launch_bounds__(512, 2) k(int iters, int mfma_on, int kind, double* sink) {
    const int half = threadIdx.x >> 8;
    double r = 0;
    if (half == 0) {
        if (mfma_on) {
            v4d c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0;
            const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
            for (int i = 0; i < iters; ++i) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, c3, 0, 0, 0);
            }
            r = c0[0] + c1[1] + c2[2] + c3[3];
        }
    } else {
        if (kind == 1) {  // 64 fp64 FMAs per iteration (16 independent chains x 4)
            double x[16];
            for (int j = 0; j < 16; ++j) x[j] = threadIdx.x * 1e-3 + j;
            const double m = 1.0000001, a = 1e-9;
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                    for (int j = 0; j < 16; ++j) x[j] = __builtin_fma(x[j], m, a);
            }
            for (int j = 0; j < 16; ++j) r += x[j];
        } else if (kind == 2) {
            float x[16];
            for (int j = 0; j < 16; ++j) x[j] = threadIdx.x * 1e-3f + j;
            const float m = 1.0000001f, a = 1e-9f;
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                    for (int j = 0; j < 16; ++j) x[j] = __builtin_fmaf(x[j], m, a);
            }
            for (int j = 0; j < 16; ++j) r += x[j];
        }
    }
    if (r == 12345.678) sink[threadIdx.x] = r;
}

float run(int blocks, int iters, int mfma_on, int kind, double* sink) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, iters, mfma_on, kind, sink);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms;
}

int main() {
    double* sink; hipMalloc(&sink, 1 << 16);
    const int blocks = 256, iters = 20000;
    run(blocks, 100, 1, 1, sink);
    float t_m = run(blocks, iters, 1, 0, sink);
    float t_v64 = run(blocks, iters, 0, 1, sink);
    float t_v32 = run(blocks, iters, 0, 2, sink);
    float t_m_v64 = run(blocks, iters, 1, 1, sink);
    float t_m_v32 = run(blocks, iters, 1, 2, sink);
    printf("MFMA only            : %.3f ms  (%.1f ns per MFMA per SIMD)\n", t_m, t_m * 1e6 / (iters * 4.0));
    printf("fp64 FMA only        : %.3f ms  (%.2f ns per wave-instr)\n", t_v64, t_v64 * 1e6 / (iters * 64.0));
    printf("fp32 FMA only        : %.3f ms  (%.2f ns per wave-instr)\n", t_v32, t_v32 * 1e6 / (iters * 64.0));
    printf("MFMA || fp64 FMA     : %.3f ms  (sum would be %.3f, max %.3f)\n", t_m_v64, t_m + t_v64, t_m > t_v64 ? t_m : t_v64);
    printf("MFMA || fp32 FMA     : %.3f ms  (sum would be %.3f, max %.3f)\n", t_m_v32, t_m + t_v32, t_m > t_v32 ? t_m : t_v32);
    for (int rep = 0; rep < 40; ++rep) {
        float ms = run(512, 40000, 1, 0, sink);
        double tf = 512.0 * 4 * 40000 * 4 * 2048.0 / (ms * 1e-3) * 1e-12;
        if (rep % 4 == 0) printf("sustained rep %2d: %.3f ms  %.1f TFLOP/s\n", rep, ms, tf);
    }
    return 0;
}
