// Microbenchmark: cycles per v_mfma_f32_16x16x4_f32 / 32x32x2_f32 in SHORT kernels (the DQN step's kernels run 5-60 us with
// one wave per SIMD), and the shader clock such kernels get (s_memtime ticks per wall_clock64 tick of 10 ns).
// Build: hipcc -w --offload-arch=gfx950 -O3 -o f32mfma_bench f32mfma_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k16(int iters, float* sink, unsigned long long* out) {
    f32x4 acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = 0.37f * (threadIdx.x % 13) - 2.f, b = 0.11f * (threadIdx.x % 17) - 1.f;
    const unsigned long long w0 = wall_clock64(), c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
    }
    const unsigned long long w1 = wall_clock64(), c1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 4; ++r) s += acc[j][r];
    if (s == 1.2345e33f) sink[0] = s;
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = w1 - w0; out[blockIdx.x * 2 + 1] = c1 - c0; }
}
__global__ __launch_bounds__(256) void k32(int iters, float* sink, unsigned long long* out) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float a = 0.37f * (threadIdx.x % 13) - 2.f, b = 0.11f * (threadIdx.x % 17) - 1.f;
    const unsigned long long w0 = wall_clock64(), c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    const unsigned long long w1 = wall_clock64(), c1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    if (s == 1.2345e33f) sink[0] = s;
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = w1 - w0; out[blockIdx.x * 2 + 1] = c1 - c0; }
}
__global__ void kvalu(int iters, float* sink, unsigned long long* out) {
    float x = threadIdx.x * 0.001f, y = 1.0001f;
    const unsigned long long w0 = wall_clock64(), c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 32; ++j) x = __builtin_fmaf(x, y, 0.5f);
    }
    const unsigned long long w1 = wall_clock64(), c1 = __builtin_amdgcn_s_memtime();
    if (x == 1.2345e33f) sink[0] = x;
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = w1 - w0; out[blockIdx.x * 2 + 1] = c1 - c0; }
}
static void report(const char* name, unsigned long long* out, int nwg, double n_inst, hipEvent_t e0, hipEvent_t e1, int reps) {
    unsigned long long h[1024];
    hipMemcpy(h, out, sizeof(unsigned long long) * 2 * nwg, hipMemcpyDeviceToHost);
    double w = 0, c = 0;
    for (int i = 0; i < nwg; ++i) { w += h[2 * i]; c += h[2 * i + 1]; }
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s in-kernel %.2f us, s_memtime ticks %.0f (%.0f MHz if shader clock), ticks per instruction %.1f, launch-to-launch %.2f us\n", name, w / nwg / 100.0,
           c / nwg, c / w * 100, c / nwg / n_inst, ms * 1e3 / reps);
}
int main() {
    float* sink; unsigned long long* out;
    hipMalloc(&sink, 64); hipMalloc(&out, 1024 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 200;
    for (int pass = 0; pass < 2; ++pass) {
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k16<24>, dim3(201), dim3(256), 0, 0, 40, sink, out);
        hipEventRecord(e1); hipDeviceSynchronize();
        report("16x16x4 f32, 24 acc, 960 per wave", out, 201, 960, e0, e1, reps);
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k16<6>, dim3(201), dim3(256), 0, 0, 64, sink, out);
        hipEventRecord(e1); hipDeviceSynchronize();
        report("16x16x4 f32, 6 acc, 384 per wave", out, 201, 384, e0, e1, reps);
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k16<1>, dim3(201), dim3(256), 0, 0, 384, sink, out);
        hipEventRecord(e1); hipDeviceSynchronize();
        report("16x16x4 f32, 1 acc (dependent)", out, 201, 384, e0, e1, reps);
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k32, dim3(201), dim3(256), 0, 0, 120, sink, out);
        hipEventRecord(e1); hipDeviceSynchronize();
        report("32x32x2 f32, 4 acc, 480 per wave", out, 201, 480, e0, e1, reps);
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kvalu, dim3(256), dim3(256), 0, 0, 300, sink, out);
        hipEventRecord(e1); hipDeviceSynchronize();
        report("dependent v_fma x 9600", out, 256, 9600, e0, e1, reps);
        hipEventRecord(e0);
        for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(k16<24>, dim3(201), dim3(256), 0, 0, 4000, sink, out);
        hipEventRecord(e1); hipDeviceSynchronize();
        report("16x16x4 f32, 24 acc, 96000/wave", out, 201, 96000, e0, e1, 20);
    }
    return 0;
}
