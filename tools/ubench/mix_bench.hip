// Does a wave overlap v_mfma_f32_4x4x1_16b_f32 with independent VALU work (fma, exp, rcp)?  ticks (s_memtime) per iteration for
// MFMA only / VALU only / both interleaved in program order, 1 and 2 waves per SIMD.
// Build: hipcc -w --offload-arch=gfx950 -O3 -o mix_bench mix_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void k(int iters, float* sink, unsigned long long* out) {
    f32x4 acc[4];
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = 0.37f * (threadIdx.x % 13) - 2.f, b = 0.11f * (threadIdx.x % 17) - 1.f;
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = 0.001f * (threadIdx.x + j);
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            if (MODE & 1) acc[u & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[u & 3], 0, 0, 0);
            if (MODE & 2) v[u & 7] = __builtin_fmaf(v[u & 7], 1.0001f, 0.5f);
            if ((MODE & 4) && (u & 7) == 0) v[(u >> 3) & 7] = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(v[(u >> 3) & 7]) + 1.0f);
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[j][r];
    for (int j = 0; j < 8; ++j) s += v[j];
    if (s == 1.2345e33f) sink[0] = s;
    if (threadIdx.x == 0) out[blockIdx.x] = c1 - c0;
}
template <int MODE>
static void run(const char* name, int threads, float* sink, unsigned long long* out) {
    const int iters = 200;
    hipLaunchKernelGGL(k<MODE>, dim3(64), dim3(threads), 0, 0, iters, sink, out);
    hipDeviceSynchronize();
    unsigned long long h[64];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0;
    for (int i = 0; i < 64; ++i) c += h[i];
    printf("%-44s %d waves/SIMD: %.0f ticks per 32-slot iteration\n", name, threads / 256, c / 64 / iters);
}
int main() {
    float* sink; unsigned long long* out;
    hipMalloc(&sink, 64); hipMalloc(&out, 64 * 8);
    for (int threads : {256, 512}) {
        run<1>("32 x mfma 4x4x1", threads, sink, out);
        run<2>("32 x v_fma", threads, sink, out);
        run<3>("32 x (mfma, v_fma) interleaved", threads, sink, out);
        run<4>("4 x (exp2, rcp)", threads, sink, out);
        run<7>("32 x (mfma, v_fma) + 4 x (exp2, rcp)", threads, sink, out);
    }
    return 0;
}
