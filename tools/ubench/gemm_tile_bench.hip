// Step A of round 5: what does a one-wave-per-SIMD, 4 x 4 register-tile, both-operands-from-LDS MFMA
// kernel reach on RANDOM bf16 data at the tower's own GEMM shapes?  Kernel: ivos-w_amd/csrc/gemm_bt.h (the product's).
// Prints per shape and ablation: us per launch, TFLOP/s, shader clock (s_memtime ticks / s_memrealtime @ 100 MHz), K-loop
// share of a workgroup's cycles, and checks 4096 sampled outputs against a host fp64 contraction of the same bf16 inputs.
// Build: hipcc -w --offload-arch=gfx950 -O3 -std=c++17 -o gemm_tile_bench gemm_tile_bench.hip
#include "../../ivos-w_amd/csrc/gemm_bt.h"
#include <math.h>
#include <stdlib.h>
#include <vector>
using namespace ivosw;
namespace ivosw { void set_error(const char*, ...) {} }

static inline uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static inline float bf2f(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
static uint64_t rng_s = 0x9e3779b97f4a7c15ull;
static inline uint32_t rnd() { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return (uint32_t)(rng_s >> 32); }
static inline float urand() { return (rnd() >> 8) * (2.0f / 16777216.0f) - 1.0f; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int ABL>
static float time_kernel(const BtArgs& a, int grid, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm_bt_kernel<ABL>, dim3(grid), dim3(256), 0, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm_bt_kernel<ABL>, dim3(grid), dim3(256), 0, 0, a);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / reps;
}

template <int ABL>
static void run(const char* name, BtArgs a, unsigned long long* ts_d, int reps) {
    const int grid = (a.M / 256) * (a.N / 256);
    BtArgs b = a; b.ts = nullptr;
    const float us = time_kernel<ABL>(b, grid, reps);
    // one stamped launch behind a warm queue
    b.ts = ts_d;
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(gemm_bt_kernel<ABL>, dim3(grid), dim3(256), 0, 0, b);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> ts((size_t)grid * 4);
    CK(hipMemcpy(ts.data(), ts_d, ts.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0, kcyc = 0, real = 0;
    for (int g = 0; g < grid; ++g) { cyc += ts[4 * g + 2] - ts[4 * g]; kcyc += ts[4 * g + 1] - ts[4 * g]; real += ts[4 * g + 3]; }
    const double mhz = cyc / real * 100.0;
    const double fl = 2.0 * a.M * a.N * a.K;
    const double nmfma = (double)(a.K / 16) * 16;                        // per wave
    printf("  %-34s %8.1f us  %7.1f TFLOP/s  clock %4.0f MHz  wg %7.0f cyc (K loop %5.1f %%, %5.1f cyc/MFMA)\n", name, us, fl / us * 1e-6, mhz,
           cyc / grid, 100.0 * kcyc / cyc, kcyc / grid / nmfma);
}

static int check(const BtArgs& a, const std::vector<uint16_t>& A, const std::vector<uint16_t>& B, const std::vector<float>& bias) {
    std::vector<uint16_t> C((size_t)a.M * a.N);
    CK(hipMemcpy(C.data(), a.C, C.size() * 2, hipMemcpyDeviceToHost));
    int bad = 0;
    double worst = 0;
    for (int s = 0; s < 4096; ++s) {
        const int m = s < 512 ? (s % 256) + (s >= 256 ? a.M - 256 : 0) : rnd() % a.M;
        const int n = s < 512 ? (s * 7 + s / 256) % a.N : rnd() % a.N;
        double acc = bias[n];
        for (int k = 0; k < a.K; ++k) acc += (double)bf2f(A[(size_t)m * a.K + k]) * bf2f(B[(size_t)n * a.K + k]);
        if (a.relu && acc < 0) acc = 0;
        const double got = bf2f(C[(size_t)m * a.N + n]);
        const double err = fabs(got - acc) / (fabs(acc) + 1.0);
        if (err > worst) worst = err;
        if (err > 1e-2) { if (bad < 5) printf("    MISMATCH m %d n %d got %f want %f\n", m, n, got, acc); ++bad; }
    }
    printf("  check: 4096 samples, worst rel err %.2e, %d bad\n", worst, bad);
    return bad;
}

static inline float grand() {        // ~ N(0, 1): sum of 12 uniforms
    float s = 0.f;
    for (int i = 0; i < 12; ++i) s += (rnd() >> 8) * (1.0f / 16777216.0f);
    return s - 6.0f;
}

int main(int argc, char** argv) {
    const int shapes[][3] = {{65536, 1024, 768}, {65536, 256, 2304}, {262144, 512, 128}, {65536, 256, 1024}, {16384, 2048, 1536}, {8192, 8192, 8192}};
    const int nshape = argc > 1 ? atoi(argv[1]) : 6;
    // data: 0 = uniform [-1, 1) (the worst case for switching power, the guide's convention), 1 = what the tower feeds its contractions:
    // A = ReLU of a gaussian (half zeros, the rest positive: a post-ReLU bf16 activation), B = gaussian weights
    const int data = argc > 2 ? atoi(argv[2]) : 0;
    printf("data: %s\n", data ? "A = relu(gaussian), B = gaussian / sqrt(K)  (tower-like)" : "A, B uniform [-1, 1)  (random)");
    int bad = 0;
    for (int si = 0; si < nshape && si < 6; ++si) {
        const int M = shapes[si][0], N = shapes[si][1], K = shapes[si][2];
        printf("M %d N %d K %d (%.1f GFLOP, %d workgroups)\n", M, N, K, 2.0 * M * N * K * 1e-9, (M / 256) * (N / 256));
        std::vector<uint16_t> A((size_t)M * K), B((size_t)N * K);
        std::vector<float> bias(N);
        const float sc = 1.0f;
        if (data) {
            for (auto& v : A) { const float g = grand(); v = f2bf(g > 0.f ? g : 0.f); }
            for (auto& v : B) v = f2bf(grand() * (1.4f / sqrtf((float)K)));
        } else {
            for (auto& v : A) v = f2bf(urand() * sc);
            for (auto& v : B) v = f2bf(urand() * (4.0f / sqrtf((float)K)));
        }
        for (auto& v : bias) v = urand();
        BtArgs a{};
        void *dA, *dB, *dC, *dbias, *dts;
        CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dC, (size_t)M * N * 2)); CK(hipMalloc(&dbias, N * 4));
        CK(hipMalloc(&dts, (size_t)(M / 256) * (N / 256) * 32));
        CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dbias, bias.data(), N * 4, hipMemcpyHostToDevice));
        CK(hipMemset(dC, 0xff, (size_t)M * N * 2));
        a.A = (const bf16_t*)dA; a.B = (const bf16_t*)dB; a.bias = (const float*)dbias; a.C = (bf16_t*)dC; a.M = M; a.N = N; a.K = K; a.relu = 1;
        const int reps = 20;
        {
            BtArgs b = a; b.ts = nullptr;
            hipLaunchKernelGGL(gemm_bt_kernel<0>, dim3((M / 256) * (N / 256)), dim3(256), 0, 0, b);
            CK(hipDeviceSynchronize());
            bad += check(a, A, B, bias);
        }
        // a second of the full kernel first: the shader clock settles under sustained MFMA load
        { BtArgs b = a; time_kernel<0>(b, (M / 256) * (N / 256), 200); }
        run<0>("full", a, (unsigned long long*)dts, reps);
        run<4>("no stores", a, (unsigned long long*)dts, reps);
        run<8>("no epilogue", a, (unsigned long long*)dts, reps);
        run<8 | 1>("no epilogue, no DMA in loop", a, (unsigned long long*)dts, reps);
        run<8 | 2>("no epilogue, no fragment reads", a, (unsigned long long*)dts, reps);
        run<8 | 3>("no epilogue, MFMA only", a, (unsigned long long*)dts, reps);
        run<16>("full, DMA slots staggered by wave", a, (unsigned long long*)dts, reps);
        run<8 | 16>("no epilogue, staggered", a, (unsigned long long*)dts, reps);
        run<8 | 32>("no epilogue, DMA re-reads 1 KiB", a, (unsigned long long*)dts, reps);
        run<8 | 2 | 32>("no epi, no frag reads, DMA 1 KiB", a, (unsigned long long*)dts, reps);
        run<0>("full (again)", a, (unsigned long long*)dts, reps);
        if (K >= 128) {           // the persistent form: one workgroup per CU, the ring runs across tile boundaries
            const int grid = (M / 256) * (N / 256), G = grid < 256 ? grid : 256;
            BtArgs b = a; b.ts = nullptr;
            CK(hipMemset(dC, 0xff, (size_t)M * N * 2));
            hipLaunchKernelGGL(gemm_bt_persist_kernel<0>, dim3(G), dim3(256), 0, 0, b);
            CK(hipDeviceSynchronize());
            bad += check(a, A, B, bias);
            for (int rep = 0; rep < 2; ++rep) {
                hipEvent_t e0, e1;
                CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm_bt_persist_kernel<0>, dim3(G), dim3(256), 0, 0, b);
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm_bt_persist_kernel<0>, dim3(G), dim3(256), 0, 0, b);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                const float us = ms * 1000.f / reps;
                printf("  %-34s %8.1f us  %7.1f TFLOP/s  (%d workgroups, %d tiles each)\n", "PERSISTENT, ring across tiles", us, 2.0 * M * N * K / us * 1e-6, G, grid / G);
                run<0>("one tile per workgroup (again)", a, (unsigned long long*)dts, reps);
            }
        }
        hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dbias); hipFree(dts);
    }
    printf(bad ? "FAILED\n" : "ALL CHECKS PASSED\n");
    return bad != 0;
}
