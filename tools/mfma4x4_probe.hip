// Micro-benchmark 5 (round 2): v_mfma_f64_4x4x4_4b_f64 on gfx950 -- (a) which lane holds which element of A, B and D,
// (b) its issue cost next to v_mfma_f64_16x16x4_f64, (c) whether two of them with an all-ones operand reduce a value over the
// 16 lanes of a row (the 16-lane sums of the fused sweep, today 8 DPP moves + 4 adds per value).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma4x4_probe.hip -o build/mfma4x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void k_map(double* out) {
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            out[(la * 64 + lb) * 64 + lane] = d;
        }
}

__global__ void k_reduce(const double* v, double* out) {
    const int lane = threadIdx.x;
    const double x = v[lane];
    const double r_a = __builtin_amdgcn_mfma_f64_4x4x4f64(x, 1.0, 0.0, 0, 0, 0);
    const double r_b = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, x, 0.0, 0, 0, 0);
    out[0 * 64 + lane] = r_a;
    out[1 * 64 + lane] = r_b;
    out[2 * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(r_a, 1.0, 0.0, 0, 0, 0);
    out[3 * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, r_a, 0.0, 0, 0, 0);
    out[4 * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(r_b, 1.0, 0.0, 0, 0, 0);
    out[5 * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, r_b, 0.0, 0, 0, 0);
}

// mode 0: 8 independent 4x4x4 chains; 1: one dependent 4x4x4 chain (x 8 per iteration); 2: 8 independent 16x16x4;
// 3: 8 fp64 FMAs (reference);  4: 4x4x4 pairs where the second consumes the first (the reduction pattern), 4 pairs
__global__ void __launch_bounds__(256, 1) k_time(int iters, int mode, double* sink) {
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    double r = 0;
    if (mode == 0) {
        double c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[j], 0, 0, 0);
        }
        for (int j = 0; j < 8; ++j) r += c[j];
    } else if (mode == 1) {
        double c = 0;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
        }
        r = c;
    } else if (mode == 2) {
        v4d c[8];
        for (int j = 0; j < 8; ++j) c[j] = v4d{0, 0, 0, 0};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[j], 0, 0, 0);
        }
        for (int j = 0; j < 8; ++j) r += c[j][0] + c[j][3];
    } else if (mode == 3) {
        double c[8] = {0, 1, 2, 3, 4, 5, 6, 7};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) c[j] = __builtin_fma(c[j], a, b);
        }
        for (int j = 0; j < 8; ++j) r += c[j];
    } else if (mode == 4) {
        double c[4] = {a, b, a + b, a - b};
        for (int i = 0; i < iters; ++i) {
            double t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(c[j], 1.0, 0.0, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, t[j], 0.0, 0, 0, 0) * 0.0625;
        }
        for (int j = 0; j < 4; ++j) r += c[j];
    }
    if (r == 12345.678) sink[threadIdx.x] = r;
}

static float run(int iters, int mode, double* sink) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_time, dim3(256), dim3(256), 0, 0, iters, mode, sink);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    double *d_out, *d_v;
    hipMalloc(&d_out, 64 * 64 * 64 * 8);
    hipMalloc(&d_v, 64 * 8);
    std::vector<double> out(64 * 64 * 64);
    hipLaunchKernelGGL(k_map, dim3(1), dim3(64), 0, 0, d_out);
    hipMemcpy(out.data(), d_out, out.size() * 8, hipMemcpyDeviceToHost);
    // For every output lane: the list of (A lane, B lane) pairs that contribute.
    printf("# map: D lane <- sum over (A lane, B lane)\n");
    for (int ld = 0; ld < 64; ++ld) {
        printf("D%02d:", ld);
        for (int la = 0; la < 64; ++la)
            for (int lb = 0; lb < 64; ++lb)
                if (out[(la * 64 + lb) * 64 + ld] != 0.0) printf(" (%d,%d)", la, lb);
        printf("\n");
    }
    std::vector<double> v(64), red(6 * 64);
    srand(7);
    for (int i = 0; i < 64; ++i) v[i] = (rand() % 1000) / 8.0;
    hipMemcpy(d_v, v.data(), 64 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_reduce, dim3(1), dim3(64), 0, 0, d_v, d_out);
    hipMemcpy(red.data(), d_out, 6 * 64 * 8, hipMemcpyDeviceToHost);
    double rs[4] = {0, 0, 0, 0};
    for (int i = 0; i < 64; ++i) rs[i / 16] += v[i];
    printf("# row sums (lanes 16 r .. 16 r + 15): %.3f %.3f %.3f %.3f\n", rs[0], rs[1], rs[2], rs[3]);
    const char* names[6] = {"mfma(x,1)", "mfma(1,x)", "mfma(mfma(x,1),1)", "mfma(1,mfma(x,1))", "mfma(mfma(1,x),1)",
                            "mfma(1,mfma(1,x))"};
    for (int m = 0; m < 6; ++m) {
        int ok = 1;
        for (int i = 0; i < 64; ++i) ok &= red[m * 64 + i] == rs[i / 16];
        printf("%-20s row-sum in every lane: %s   lanes 0,1,4,5,16,20: %.3f %.3f %.3f %.3f %.3f %.3f\n", names[m], ok ? "YES" : "no",
               red[m * 64], red[m * 64 + 1], red[m * 64 + 4], red[m * 64 + 5], red[m * 64 + 16], red[m * 64 + 20]);
    }
    double* sink;
    hipMalloc(&sink, 1 << 16);
    const int iters = 100000;
    run(1000, 0, sink);
    const char* tn[5] = {"8 independent 4x4x4_4b", "8 dependent 4x4x4_4b", "8 independent 16x16x4", "8 independent fp64 FMA",
                         "4 reduction pairs (4x4x4 -> 4x4x4 -> mul)"};
    for (int m = 0; m < 5; ++m) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            float ms = run(iters, m, sink);
            if (ms < best) best = ms;
        }
        printf("%-44s %.3f ms  = %.2f ns per instruction (x 2.1-2.4 GHz = %.1f-%.1f cycles)\n", tn[m], best,
               best * 1e6 / (iters * 8.0), best * 1e6 / (iters * 8.0) * 2.1, best * 1e6 / (iters * 8.0) * 2.4);
    }
    return 0;
}
