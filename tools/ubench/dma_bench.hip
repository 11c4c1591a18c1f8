// Microbenchmark: per-CU fill rate of LDS-DMA (global_load_lds_dwordx4) and plain global_load_dwordx4 as a function
// of waves per CU, requests in flight per wave, source footprint (L2 / Infinity Cache / HBM) and row stride.
// Build: hipcc --offload-arch=gfx950 -O3 -o dma_bench dma_bench.hip ; tuning aid, not part of the library.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

// each wave: `iters` rounds of DEPTH x (1 KB DMA); row segment = 128 B, 8 rows per instruction, row stride `stride` bytes
template <int DEPTH, bool DMA>
__global__ __launch_bounds__(1024) void fill_kernel(const char* src, size_t footprint, int stride, int iters, unsigned* sink, int mode = 0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    // block-private slice of the footprint so that different CUs read different lines
    const size_t slice = footprint / gridDim.x;
    const char* base = src + ((mode & 16) ? (size_t)0 : (mode & 32) ? (size_t)(blockIdx.x & 7) * slice : (size_t)blockIdx.x * slice);   // 16: every CU reads the same slice; 32: one slice per XCD
    const size_t wave_span = slice / nw;
    const char* wb = base + (size_t)wave * wave_span;
    const int row = lane >> 3;
    const size_t lane_off = (size_t)row * stride + (((lane & 7) ^ ((mode & 1) ? ((row * 5 + wave) & 7) : 0)) * 16);
    const size_t step = (size_t)8 * stride;     // bytes consumed per instruction (8 rows)
    const size_t wrap = wave_span / step * step;
    size_t off = 0;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const char* p = wb + off + lane_off;
            if ((mode & 2) && d % 6 == 5) p = src;
            if ((mode & 4) && row == 7) p = src;
            if constexpr (DMA) {
                if (mode & 8) __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(lds + (wave * DEPTH + d) * 1024), 16, 0, 16);   // sc1: bypass L1
                else __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(lds + (wave * DEPTH + d) * 1024), 16, 0, 0);
            } else {
                const uint4 v = *reinterpret_cast<const uint4*>(p);
                acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
            }
            off += step;
            if (off >= wrap) off = 0;
        }
        if constexpr (DMA) wait_vmcnt<DEPTH / 2>();   // keep half of the ring in flight
    }
    if constexpr (DMA) wait_vmcnt<0>();
    if (acc.x == 0x12345u) sink[0] = acc.y ^ acc.z ^ acc.w;
}

template <int DEPTH, bool DMA>
static double run(const char* src, size_t footprint, int stride, int waves, int iters, unsigned* sink, int ncu, int mode = 0) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t ldsb = DMA ? (size_t)waves * DEPTH * 1024 : 0;
    hipFuncSetAttribute((const void*)fill_kernel<DEPTH, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((fill_kernel<DEPTH, DMA>), dim3(ncu), dim3(waves * 64), ldsb, 0, src, footprint, stride, iters, sink, mode);
    hipEventRecord(e0);
    hipLaunchKernelGGL((fill_kernel<DEPTH, DMA>), dim3(ncu), dim3(waves * 64), ldsb, 0, src, footprint, stride, iters, sink, mode);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)ncu * waves * iters * DEPTH * 1024.0;
    return bytes / (ms * 1e-3) / 1e9;   // GB/s chip-wide
}

int main() {
    const int ncu = 256;
    const size_t cap = (size_t)8 << 30;
    char* src;
    unsigned* sink;
    hipMalloc(&src, cap);
    hipMalloc(&sink, 64);
    hipMemset(src, 1, cap);
    printf("%-5s %-6s %-9s %-7s %-6s %10s %12s\n", "kind", "waves", "footprint", "stride", "depth", "GB/s", "B/clk/CU@2.1");
    // (4) sharing: every CU streams its own slice (0) / all CUs the same slice (16) / one slice per XCD (32); row stride 512 and 8192
    for (int stride : {512, 8192})
        for (int mode : {0, 16, 32}) {
            const size_t fp = (size_t)256 * 64 * 1024 * (stride / 128);
            double a = fp <= cap ? run<8, true>(src, fp, stride, 8, 400, sink, 256, mode) : 0;
            printf("stride %-5d share-mode %-2d L2-resident: %8.0f GB/s %6.1f B/clk/CU\n", stride, mode, a, a / 256 / 2.1);
        }
    return 0;
}
