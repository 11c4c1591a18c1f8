// Microbenchmark: ds_read_b128 throughput of the swizzled 128-B-row tile image used by the tower kernels, alone and
// with LDS-DMA landing in the same LDS.  Build: hipcc -w --offload-arch=gfx950 -O3 -o lds_bench lds_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __forceinline__ u32x4 lds_read_b128(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
// mode 0: swizzled fragment pattern (row = lane&31, chunk = 2ks + lane>>5, chunk ^= (row>>1)&7)
// mode 1: unswizzled (same without the XOR)   mode 2: lane-linear (addr = lane*16)
// dma: number of 1-KB LDS-DMA per wave per iteration issued alongside (from an L2-resident buffer)
__global__ __launch_bounds__(512) void lds_kernel(int mode, int iters, int dma, const char* src, unsigned* sink, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[160 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int lrow = lane & 31, lhalf = lane >> 5;
    unsigned addr[4][4];
    for (int t = 0; t < 4; ++t)
        for (int ks = 0; ks < 4; ++ks) {
            const int row = ((wave + 2 * t) % 8) * 32 + lrow;
            const int ch = 2 * ks + lhalf;
            if (mode == 0) addr[t][ks] = base + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4);
            else if (mode == 1) addr[t][ks] = base + row * 128 + (ch << 4);
            else addr[t][ks] = base + (t * 4 + ks) * 1024 + lane * 16;
        }
    u32x4 acc = {0, 0, 0, 0};
    const char* s = src + ((size_t)blockIdx.x * 8 + wave) * 8192 + lane * 16;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        for (int d = 0; d < dma; ++d)
            __builtin_amdgcn_global_load_lds((gptr_t)(s + (d & 7) * 1024), (lptr_t)(lds + 64 * 1024 + (wave * 8 + (d & 7)) * 1024), 16, 0, 0);
        u32x4 v[16];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) v[t * 4 + ks] = lds_read_b128(addr[t][ks]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 16; ++i) acc ^= v[i];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc[0] == 0x1234567u) sink[0] = acc[1] ^ acc[2] ^ acc[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    char* src; unsigned* sink; unsigned long long* cyc;
    hipMalloc(&src, 256 * 8 * 8192); hipMemset(src, 1, 256 * 8 * 8192);
    hipMalloc(&sink, 64); hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    for (int dma : {0, 1, 2, 4})
        for (int mode : {0, 1, 2}) {
            hipLaunchKernelGGL(lds_kernel, dim3(256), dim3(512), 0, 0, mode, iters, dma, src, sink, cyc);
            hipDeviceSynchronize();
            unsigned long long h[256];
            hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            double m = 0;
            for (int i = 0; i < 256; ++i) m += h[i];
            m /= 256;
            const double bytes = 8.0 * 16 * 1024 * iters;
            printf("mode %d dma/wave/iter %d: %8.0f cycles, ds_read %6.1f B/clk/CU, dma %5.1f B/clk/CU\n", mode, dma, m, bytes / m, 8.0 * dma * 1024 * iters / m);
        }
    return 0;
}
