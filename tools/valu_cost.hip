// Micro-benchmark 3: issue cost (cycles per wave-instruction per SIMD) of the fp64 VALU instructions an exp()
// is made of, at 1 / 2 / 4 waves per SIMD.  Each loop body is 32 independent inline-asm instructions.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_cost.hip -o build/valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#define REP8(x) x x x x x x x x
#define BODY(ASM, ...)                                                                  \
    for (int i = 0; i < iters; ++i) {                                                   \
        REP8(asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : __VA_ARGS__);) \
    }

template <int OP>
__global__ void k(int iters, double* sink, long long* cyc) {
    double a0 = threadIdx.x * 1e-3 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    double b = 1.0000001, c = 1e-9;
    int e = 1;
    long long t0 = __builtin_readcyclecounter();
    if (OP == 0) { BODY("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5", "v"(b), "v"(c)) }
    if (OP == 1) { BODY("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4", "v"(c)) }
    if (OP == 2) { BODY("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4", "v"(b)) }
    if (OP == 3) { BODY("v_max_f64 %0, %0, %4\n v_max_f64 %1, %1, %4\n v_max_f64 %2, %2, %4\n v_max_f64 %3, %3, %4", "v"(c)) }
    if (OP == 4) { BODY("v_rndne_f64 %0, %0\n v_rndne_f64 %1, %1\n v_rndne_f64 %2, %2\n v_rndne_f64 %3, %3", "v"(c)) }
    if (OP == 5) { BODY("v_ldexp_f64 %0, %0, %4\n v_ldexp_f64 %1, %1, %4\n v_ldexp_f64 %2, %2, %4\n v_ldexp_f64 %3, %3, %4", "v"(e)) }
    if (OP == 6) { BODY("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3", "v"(c)) }
    if (OP == 7) { BODY("v_fma_f64 %0, %0, %4, s[2:3]\n v_fma_f64 %1, %1, %4, s[2:3]\n v_fma_f64 %2, %2, %4, s[2:3]\n v_fma_f64 %3, %3, %4, s[2:3]", "v"(b)) }
    if (OP == 8) {
        int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
        for (int i = 0; i < iters; ++i) {
            REP8(asm volatile("v_cvt_i32_f64 %0, %4\n v_cvt_i32_f64 %1, %5\n v_cvt_i32_f64 %2, %6\n v_cvt_i32_f64 %3, %7"
                              : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));)
        }
        a0 += i0 + i1 + i2 + i3;
    }
    if (OP == 9) {  // 32-bit moves through DPP (row reductions), 2 per fp64 value
        int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
        for (int i = 0; i < iters; ++i) {
            REP8(asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                              "v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                              "v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                              "v_mov_b32_dpp %3, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1"
                              : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));)
        }
        a0 += i0 + i1 + i2 + i3;
    }
    if (OP == 10) {  // compare + cndmask
        int i0 = threadIdx.x, i1 = i0 + 1;
        for (int i = 0; i < iters; ++i) {
            REP8(asm volatile("v_cmp_gt_f64 vcc, %2, %4\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_gt_f64 vcc, %3, %4\n v_cndmask_b32 %1, %1, %0, vcc"
                              : "+v"(i0), "+v"(i1) : "v"(a0), "v"(a1), "v"(c) : "vcc");)
        }
        a0 += i0 + i1;
    }
    if (OP == 11 || OP == 12) {
        float f0 = threadIdx.x * 1e-3f, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, fb = 1.0000001f, fc = 1e-9f;
        for (int i = 0; i < iters; ++i) {
            if (OP == 11) {
                REP8(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                                  : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fb), "v"(fc));)
            } else {
                REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3"
                                  : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));)
            }
        }
        a0 += f0 + f1 + f2 + f3;
    }
    if (OP == 13) { BODY("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5", "v"(b), "v"(c)) }
    long long t1 = __builtin_readcyclecounter();
    double v = a0 + a1 + a2 + a3;
    if (v == 12345.678) sink[threadIdx.x] = v;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int OP>
void run(const char* name, int instr_per_body, double* sink, long long* cyc) {
    const int iters = 4000;
    printf("%-34s", name);
    for (int wps : {1, 2, 4}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(256 * wps), dim3(256), 0, 0, 10, sink, cyc);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(256 * wps), dim3(256), 0, 0, iters, sink, cyc);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long hc; hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
        const double n = (double)iters * 8 * instr_per_body;   // instructions per wave
        printf("  wps=%d: %6.2f cyc/instr/SIMD (one wave sees %6.2f)", wps, ms * 1e-3 * 2.4e9 / (n * wps), (double)hc / n);
    }
    printf("\n");
}

int main() {
    double* sink; long long* cyc;
    hipMalloc(&sink, 1 << 16); hipMalloc(&cyc, 8);
    run<0>("v_fma_f64 (vgpr operands)", 4, sink, cyc);
    run<7>("v_fma_f64 (sgpr addend)", 4, sink, cyc);
    run<1>("v_add_f64", 4, sink, cyc);
    run<2>("v_mul_f64", 4, sink, cyc);
    run<3>("v_max_f64", 4, sink, cyc);
    run<4>("v_rndne_f64", 4, sink, cyc);
    run<5>("v_ldexp_f64", 4, sink, cyc);
    run<6>("v_rcp_f64", 4, sink, cyc);
    run<8>("v_cvt_i32_f64", 4, sink, cyc);
    run<9>("v_mov_b32_dpp (+s_nop 1)", 4, sink, cyc);
    run<10>("v_cmp_gt_f64 + v_cndmask_b32 (pairs)", 4, sink, cyc);
    run<11>("v_fma_f32", 4, sink, cyc);
    run<12>("v_exp_f32", 4, sink, cyc);
    run<13>("v_pk_fma_f32", 4, sink, cyc);
    return 0;
}
