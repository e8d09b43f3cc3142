// Micro-benchmark: HBM -> LDS staging bandwidth of one-wave tiles (128 state rows x 16 samples = 16 KB, moved by
// 16 LDS-DMA instructions of 1 KB) for two layouts of the K x N matrix in HBM:
//   row-major   u[k][n] (row pitch ld):   a tile is 128 segments of 128 bytes, one per row, rows ~80 MB apart
//   tile-major  u[n / 16][k][n % 16]:     a tile is one contiguous 16 KB block
// plus a plain coalesced register read of the same bytes as the streaming reference.
// Build: hipcc --offload-arch=gfx950 -O3 tools/dma_pattern.hip -o build/dma_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int ROWS = 128, TS = 16;

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void dma16(const char* base /*uniform*/, uint32_t voff, char* dst) {
    uint64_t ub = reinterpret_cast<uint64_t>(base);
    asm("" : "+s"(ub));
    asm("" : "+v"(voff));
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(ub) + voff),
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

// MODE 0: row-major, 1: tile-major.  WPB waves per block, each wave double-buffers its own tiles.
template <int MODE, int WPB, int NREAD = 4>
__global__ void __launch_bounds__(WPB * 64) k_dma(const double* __restrict__ u, int64_t ld, int64_t ntiles, double* sink,
                                                  int delay /* x64 clocks of s_sleep per tile: stands in for the compute */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* buf = smem + wave * (2 * ROWS * TS * 8);
    const int64_t gw = (int64_t)blockIdx.x * WPB + wave, W = (int64_t)gridDim.x * WPB;
    const uint32_t voff = MODE == 0 ? (uint32_t)(((int64_t)(lane >> 3) * ld + 2 * (lane & 7)) * 8) : (uint32_t)(lane * 16);
    auto stage = [&](int64_t t, char* dst) {
#pragma unroll
        for (int j = 0; j < ROWS / 8; ++j) {
            const char* base = MODE == 0 ? reinterpret_cast<const char*>(u + (int64_t)(8 * j) * ld + t * TS)
                                         : reinterpret_cast<const char*>(u + t * (ROWS * TS) + j * 128);
            dma16(base, voff, dst + j * 1024);
        }
    };
    double acc = 0.0;
    int64_t t = gw;
    int cur = 0;
    if (t < ntiles) stage(t, buf);
    for (; t < ntiles; t += W) {
        if (t + W < ntiles) {
            stage(t + W, buf + (cur ^ 1) * (ROWS * TS * 8));
            wait_vm<ROWS / 8>();
        } else {
            wait_vm<0>();
        }
        const char* cb = buf + cur * (ROWS * TS * 8);
#pragma unroll
        for (int i = 0; i < NREAD; ++i) acc += *reinterpret_cast<const double*>(cb + ((i * 64 + lane) * 8 * (32 / NREAD)) % (ROWS * TS * 8));
        for (int d = 0; d < delay; ++d) __builtin_amdgcn_s_sleep(1);  // 64 clocks each
        cur ^= 1;
    }
    if (acc == 12345.6789) sink[threadIdx.x] = acc;
}

__global__ void __launch_bounds__(256) k_stream(const double2* __restrict__ u, int64_t n2, double* sink) {
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) {
        const double2 v = u[i];
        acc += v.x + v.y;
    }
    if (acc == 12345.6789) sink[threadIdx.x] = acc;
}

template <int MODE, int WPB, int NREAD = 4>
void run(const double* u, int64_t ld, int64_t ntiles, double* sink, int blocks, const char* name, int delay = 0) {
    const size_t lds = (size_t)WPB * 2 * ROWS * TS * 8;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_dma<MODE, WPB, NREAD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_dma<MODE, WPB, NREAD>), dim3(blocks), dim3(WPB * 64), lds, 0, u, ld, ntiles, sink, delay);
        hipEventRecord(e1);
        hipDeviceSynchronize();
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)ntiles * ROWS * TS * 8;
    printf("%-34s blocks=%4d waves/block=%d delay=%3d x64clk : %.3f ms  %.0f GB/s  (%s)\n", name, blocks, WPB, delay, ms,
           bytes / ms * 1e-6, hipGetErrorString(hipGetLastError()));
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const int64_t N = 10'000'000, ld = N, ntiles = N / TS;
    double *u, *sink;
    hipMalloc(&u, (size_t)ROWS * N * 8);
    hipMalloc(&sink, 4096 * 8);
    hipMemset(u, 0, (size_t)ROWS * N * 8);
    printf("device %s CUs=%d, %.2f GB per sweep\n", p.gcnArchName, cus, ROWS * N * 8e-9);
    run<0, 4>(u, ld, ntiles, sink, cus, "row-major tiles (128 x 128 B)");
    run<1, 4>(u, ld, ntiles, sink, cus, "tile-major tiles (16 KB blocks)");
    for (int delay : {20, 40, 60, 80, 100, 120})  // 64-clock units: 100 = 6400 clocks = 3 us of "compute" per tile
        run<0, 4>(u, ld, ntiles, sink, cus, "row-major + per-tile delay", delay);
    for (int delay : {80, 90, 100, 110})
        run<1, 4>(u, ld, ntiles, sink, cus, "tile-major + per-tile delay", delay);
    for (int delay : {90, 100, 110})
        run<0, 4>(u, ld, ntiles, sink, cus, "row-major + per-tile delay", delay);
    for (int delay : {0, 60, 80, 90})  // the whole tile is also READ from LDS (32 ds_read_b64 per lane), like the sweeps do
        run<0, 4, 32>(u, ld, ntiles, sink, cus, "row-major + full LDS read + delay", delay);
    run<0, 2>(u, ld, ntiles, sink, 2 * cus, "row-major tiles (128 x 128 B)");
    run<1, 2>(u, ld, ntiles, sink, 2 * cus, "tile-major tiles (16 KB blocks)");
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int blocks : {cus * 4, cus * 8, cus * 16}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, 0, reinterpret_cast<const double2*>(u),
                               (int64_t)ROWS * N / 2, sink);
            hipEventRecord(e1);
            hipDeviceSynchronize();
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("coalesced register stream, %5d blocks : %.3f ms  %.0f GB/s\n", blocks, ms, ROWS * N * 8.0 / ms * 1e-6);
    }
    return 0;
}
