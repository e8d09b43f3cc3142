// Micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 on gfx950 (cycles per instruction per SIMD and
// chip TFLOP/s) for different numbers of independent accumulators and waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o build/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k(int iters, double* sink, long long* cyc) {
    v4d c[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) c[i] = v4d{0, 0, 0, 0};
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    double v = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) v += c[i][0] + c[i][3];
    if (v == 12345.678) sink[threadIdx.x] = v;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC>
void run(int waves_per_simd, int cus) {
    double* sink; long long* cyc;
    hipMalloc(&sink, 4096 * 8); hipMalloc(&cyc, 8);
    const int iters = 20000;
    const int threads = 256;                   // 4 waves = 1 per SIMD
    const int blocks = cus * waves_per_simd;   // blocks per CU = waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, 200, sink, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, iters, sink, cyc);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long hc; hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    double nm = (double)blocks * 4 * iters * NACC;
    double tf = nm * 2048.0 / (ms * 1e-3) * 1e-12;
    double ns_per_mfma_simd = (ms * 1e6) / ((double)iters * NACC * waves_per_simd);
    printf("acc=%d waves/SIMD=%d : %.2f TFLOP/s, %.1f ns per MFMA per SIMD, memtime ticks per MFMA (one wave) %.1f, %.3f ms\n",
           NACC, waves_per_simd, tf, ns_per_mfma_simd, (double)hc / ((double)iters * NACC), ms);
    hipFree(sink); hipFree(cyc);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("device %s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    int cus = p.multiProcessorCount;
    run<1>(1, cus); run<2>(1, cus); run<4>(1, cus); run<8>(1, cus);
    run<1>(2, cus); run<2>(2, cus); run<4>(2, cus); run<8>(2, cus);
    run<4>(4, cus); run<2>(8, cus);
    return 0;
}
