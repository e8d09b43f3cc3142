// Accuracy of v_rcp_f64 on gfx950 and of one / two Newton steps behind it (recip_fast of mbar_device.h takes two): maximum relative
// error over 2^22 arguments spread over 1e-300 .. 1e300.  hipcc --offload-arch=gfx950 -O3 tools/rcp_accuracy.hip -o /tmp/rcp && /tmp/rcp
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* e, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = x[i];
    double r0 = __builtin_amdgcn_rcp(d);
    double r1 = fma(fma(-d, r0, 1.0), r0, r0);
    double r2 = fma(fma(-d, r1, 1.0), r1, r1);
    const double t = 1.0 / d;  // IEEE division (div_scale / div_fmas / div_fixup sequence)
    e[3 * i + 0] = fabs(r0 - t) / fabs(t);
    e[3 * i + 1] = fabs(r1 - t) / fabs(t);
    e[3 * i + 2] = fabs(r2 - t) / fabs(t);
}
int main() {
    const int n = 1 << 22;
    std::vector<double> x(n), e(3 * n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u = (double)(s >> 11) / 9007199254740992.0;
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double v = (double)(s >> 11) / 9007199254740992.0;
        x[i] = (1.0 + u) * std::pow(10.0, 600.0 * v - 300.0);
    }
    double *dx, *de;
    (void)hipMalloc(&dx, n * 8); (void)hipMalloc(&de, 3 * n * 8);
    (void)hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, de, n);
    (void)hipMemcpy(e.data(), de, 3 * n * 8, hipMemcpyDeviceToHost);
    double m[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i) for (int j = 0; j < 3; ++j) m[j] = std::fmax(m[j], e[3 * i + j]);
    std::printf("max relative error against IEEE 1/d over %d arguments: v_rcp_f64 %.3e (2^%.1f), + one Newton step %.3e (2^%.1f), + two %.3e (2^%.1f)\n",
                n, m[0], std::log2(m[0]), m[1], std::log2(m[1] > 0 ? m[1] : 1e-300), m[2], std::log2(m[2] > 0 ? m[2] : 1e-300));
    return 0;
}
