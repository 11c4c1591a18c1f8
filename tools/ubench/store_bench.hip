// Microbenchmark: how fast all CUs together can WRITE a [pixels][Cout] bf16 tensor when a wave-instruction (64 lanes x 16 B = 1 KB)
// covers SEG bytes of contiguous channels per pixel (SEG = 64: the store passes of conv1x1_wide / stage_first / the wide bottleneck
// kernels today — a wave owns one 32-channel tile; 128, 256, 512, 1024: wider row segments), with 8 waves per workgroup writing
// adjacent segments of the same pixels, one workgroup per CU, plain or non-temporal stores.  Tuning aid, not part of the library.
// Build: hipcc --offload-arch=gfx950 -O3 -o store_bench store_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// workgroup b writes pixel tiles b, b + grid, ...: a tile = 256 pixels x COUT channels (bf16).  Wave w, pass over the tile:
// SEG bytes per pixel per instruction -> 1024 / SEG pixels per instruction; the 8 waves split the COUT * 2 bytes of a pixel row
// into 8 column groups of COUT * 2 / 8 bytes, each written as (COUT * 2 / 8) / SEG segments.
template <int SEG, bool NT>
__global__ __launch_bounds__(512) void store_kernel(unsigned char* y, int cout, int tiles, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int LPP = SEG / 16;                    // lanes per pixel
    constexpr int PPI = 64 / LPP;                    // pixels per instruction
    const size_t rowb = (size_t)cout * 2;            // bytes per pixel
    const int colb = (int)(rowb / 8);                // bytes per pixel owned by one wave
    const int nseg = colb / SEG;                     // segments per wave and pixel (>= 1)
    const u32x4 v = {(unsigned)lane, (unsigned)wave, 0x3f803f80u, 0x3f803f80u};
    for (int it = 0; it < iters; ++it)
        for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
            unsigned char* base = y + (size_t)t * 256 * rowb + (size_t)wave * colb + (lane % LPP) * 16;
            for (int s = 0; s < nseg; ++s)
#pragma unroll 4
                for (int p0 = 0; p0 < 256; p0 += PPI) {
                    u32x4* dst = reinterpret_cast<u32x4*>(base + (size_t)(p0 + lane / LPP) * rowb + (size_t)s * SEG);
                    if (NT) __builtin_nontemporal_store(v, dst);
                    else *dst = v;
                }
        }
}

template <int SEG, bool NT>
static void run(unsigned char* y, int cout, int tiles, int ncu) {
    if (cout * 2 / 8 < SEG) return;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4;
    hipLaunchKernelGGL((store_kernel<SEG, NT>), dim3(ncu), dim3(512), 0, 0, y, cout, tiles, 1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((store_kernel<SEG, NT>), dim3(ncu), dim3(512), 0, 0, y, cout, tiles, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)iters * tiles * 256.0 * cout * 2;
    printf("CUs %3d  Cout %5d  seg %4d B  %s  %7.1f us  %7.1f GB/s  %5.1f GB/s per CU\n", ncu, cout, SEG, NT ? "nt   " : "plain", ms * 1e3 / iters, bytes / ms / 1e6,
           bytes / ms / 1e6 / ncu);
}

int main(int argc, char** argv) {
    int ncu = argc > 1 ? atoi(argv[1]) : 256;        // workgroups (= CUs storing at once): 256, or fewer to see the per-CU rate alone
    const size_t total = (size_t)256 << 20;          // 256 MB written per pass: the size of a res3 / res4 activation tensor at B = 256
    unsigned char* y;
    hipMalloc(&y, total);
    hipMemset(y, 0, total);
    for (int cout : {512, 1024, 2048}) {
        const int tiles = (int)(total / ((size_t)256 * cout * 2));
        run<64, false>(y, cout, tiles, ncu);  run<64, true>(y, cout, tiles, ncu);
        run<128, false>(y, cout, tiles, ncu); run<128, true>(y, cout, tiles, ncu);
        run<256, false>(y, cout, tiles, ncu); run<256, true>(y, cout, tiles, ncu);
        run<512, false>(y, cout, tiles, ncu); run<512, true>(y, cout, tiles, ncu);
    }
    hipFree(y);
    return 0;
}
