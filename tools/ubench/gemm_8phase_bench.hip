// re-base the "attainable roof" on the CDNA4 guide's 256^2 8-phase plain-HIP template instead of this repo's own
// gemm_bt probe.  Kernel: ivos-w_amd/csrc/gemm_8phase.h (re-derived from the guide's geometry table; the example source is absent).
// Runs 4096^3, 8192^3 and the five tower shapes of profiles/r05_gemm_tile_bench.txt on RANDOM bf16 (uniform [-1, 1), the guide's
// convention; `data 1` = tower-like), with the ablation ladder (no stores / no epilogue / no DMA / no reads / MFMA only / no setprio /
// no stagger / linear LDS), interleaved rounds in one process, and a sampled host fp64 check per shape.
// Build: hipcc -w --offload-arch=gfx950 -O3 -std=c++17 -o gemm_8phase_bench gemm_8phase_bench.hip
// Usage: gemm_8phase_bench [nshapes=7] [data=0] [rounds=3]
#include "../../ivos-w_amd/csrc/gemm_8phase.h"
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
using namespace ivosw;
namespace ivosw { void set_error(const char*, ...) {} }

static inline uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static inline float bf2f(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
static uint64_t rng_s = 0x9e3779b97f4a7c15ull;
static inline uint32_t rnd() { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return (uint32_t)(rng_s >> 32); }
static inline float urand() { return (rnd() >> 8) * (2.0f / 16777216.0f) - 1.0f; }
static inline float grand() { float s = 0.f; for (int i = 0; i < 12; ++i) s += (rnd() >> 8) * (1.0f / 16777216.0f); return s - 6.0f; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int ABL>
static float time_kernel(const G8Args& a, int grid, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(gemm_8phase_kernel<ABL>, dim3(grid), dim3(512), 0, 0, a);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm_8phase_kernel<ABL>, dim3(grid), dim3(512), 0, 0, a);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms * 1000.f / reps;
}

template <int ABL>
static void run(const char* name, G8Args a, unsigned long long* ts_d, int reps, int rounds) {
    const int grid = (a.M / 256) * (a.N / 256);
    G8Args b = a; b.ts = nullptr;
    std::vector<float> us(rounds);
    for (int r = 0; r < rounds; ++r) us[r] = time_kernel<ABL>(b, grid, reps);
    std::sort(us.begin(), us.end());
    b.ts = ts_d;
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(gemm_8phase_kernel<ABL>, dim3(grid), dim3(512), 0, 0, b);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> ts((size_t)grid * 4);
    CK(hipMemcpy(ts.data(), ts_d, ts.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0, kcyc = 0, real = 0;
    for (int g = 0; g < grid; ++g) { cyc += ts[4 * g + 2] - ts[4 * g]; kcyc += ts[4 * g + 1] - ts[4 * g]; real += ts[4 * g + 3]; }
    const double mhz = cyc / real * 100.0;
    const double fl = 2.0 * a.M * a.N * a.K;
    const double nmfma = (double)(a.K / 64) * 64;      // per wave (16-cycle MFMAs; two waves share a SIMD)
    printf("  %-30s min %8.1f med %8.1f us  %7.1f / %7.1f TFLOP/s  clock %4.0f MHz  wg %7.0f cyc (K loop %5.1f %%, %5.2f cyc per MFMA of the SIMD)\n", name,
           us[0], us[rounds / 2], fl / us[0] * 1e-6, fl / us[rounds / 2] * 1e-6, mhz, cyc / grid, 100.0 * kcyc / cyc, kcyc / grid / (2 * nmfma));
}

static int check(const G8Args& a, const std::vector<uint16_t>& A, const std::vector<uint16_t>& B, const std::vector<float>& bias, const char* what) {
    std::vector<uint16_t> C((size_t)a.M * a.N);
    CK(hipMemcpy(C.data(), a.C, C.size() * 2, hipMemcpyDeviceToHost));
    int bad = 0;
    double worst = 0;
    for (int s = 0; s < 8192; ++s) {
        // the first 1024 samples walk every row and column residue of the first / last tile (a transposed or mis-mapped fragment
        // shows there), the rest are random
        const int m = s < 1024 ? (s % 256) + (s >= 512 ? a.M - 256 : 0) : rnd() % a.M;
        const int n = s < 1024 ? (s * 7 + s / 256) % a.N : rnd() % a.N;
        double acc = bias[n];
        for (int k = 0; k < a.K; ++k) acc += (double)bf2f(A[(size_t)m * a.K + k]) * bf2f(B[(size_t)n * a.K + k]);
        if (a.relu && acc < 0) acc = 0;
        const double got = bf2f(C[(size_t)m * a.N + n]);
        const double err = fabs(got - acc) / (fabs(acc) + 1.0);
        if (err > worst) worst = err;
        if (err > 1e-2) { if (bad < 5) printf("    MISMATCH m %d n %d got %f want %f\n", m, n, got, acc); ++bad; }
    }
    printf("  check (%s): 8192 samples, worst rel err %.2e, %d bad\n", what, worst, bad);
    return bad;
}

int main(int argc, char** argv) {
    const int shapes[][3] = {{4096, 4096, 4096}, {8192, 8192, 8192}, {65536, 1024, 768}, {65536, 256, 2304}, {262144, 512, 128}, {65536, 256, 1024}, {16384, 2048, 1536}};
    const int nshape = argc > 1 ? atoi(argv[1]) : 7;
    const int data = argc > 2 ? atoi(argv[2]) : 0;
    const int rounds = argc > 3 ? atoi(argv[3]) : 3;
    printf("data: %s\n", data ? "A = relu(gaussian), B = gaussian / sqrt(K)  (tower-like)" : "A, B uniform [-1, 1)  (random)");
    int bad = 0;
    for (int si = 0; si < nshape && si < 7; ++si) {
        const int M = shapes[si][0], N = shapes[si][1], K = shapes[si][2];
        const int grid = (M / 256) * (N / 256);
        printf("M %d N %d K %d (%.1f GFLOP, %d workgroups)\n", M, N, K, 2.0 * M * N * K * 1e-9, grid);
        std::vector<uint16_t> A((size_t)M * K), B((size_t)N * K);
        std::vector<float> bias(N);
        if (data) {
            for (auto& v : A) { const float g = grand(); v = f2bf(g > 0.f ? g : 0.f); }
            for (auto& v : B) v = f2bf(grand() * (1.4f / sqrtf((float)K)));
        } else {
            for (auto& v : A) v = f2bf(urand());
            for (auto& v : B) v = f2bf(urand() * (4.0f / sqrtf((float)K)));
        }
        for (auto& v : bias) v = urand();
        G8Args a{};
        void *dA, *dB, *dC, *dbias, *dts;
        CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dC, (size_t)M * N * 2)); CK(hipMalloc(&dbias, N * 4));
        CK(hipMalloc(&dts, (size_t)grid * 32));
        CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dbias, bias.data(), N * 4, hipMemcpyHostToDevice));
        a.A = (const bf16_t*)dA; a.B = (const bf16_t*)dB; a.bias = (const float*)dbias; a.C = (bf16_t*)dC; a.R = nullptr;
        a.M = M; a.N = N; a.K = K; a.ldc = N; a.relu = 1; a.K1 = K; a.A2 = nullptr;
        // correctness first, and again after the timed runs (a race that needs load to show); the un-staggered form too
        CK(hipMemset(dC, 0xff, (size_t)M * N * 2));
        hipLaunchKernelGGL(gemm_8phase_kernel<0>, dim3(grid), dim3(512), 0, 0, a);
        CK(hipDeviceSynchronize());
        bad += check(a, A, B, bias, "cold");
        const int reps = si < 2 ? 10 : 20;
        // ~ a second of the full kernel: the shader clock settles under sustained MFMA load
        { G8Args b = a; time_kernel<0>(b, grid, si == 1 ? 60 : 200); }
        run<0>("full", a, (unsigned long long*)dts, reps, rounds);
        run<4>("no stores", a, (unsigned long long*)dts, reps, rounds);
        run<8>("no epilogue (K loop)", a, (unsigned long long*)dts, reps, rounds);
        run<8 | 16>("K loop, no setprio", a, (unsigned long long*)dts, reps, rounds);
        run<8 | 32>("K loop, no stagger", a, (unsigned long long*)dts, reps, rounds);
        run<8 | 64>("K loop, linear LDS (conflicts)", a, (unsigned long long*)dts, reps, rounds);
        run<8 | 1>("K loop, no DMA", a, (unsigned long long*)dts, reps, rounds);
        run<8 | 2>("K loop, no fragment reads", a, (unsigned long long*)dts, reps, rounds);
        run<8 | 3>("MFMA + barriers only", a, (unsigned long long*)dts, reps, rounds);
        run<0>("full (again)", a, (unsigned long long*)dts, reps, rounds);
        // race screen: several launches under load, full check each
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(dC, 0xff, (size_t)M * N * 2));
            for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(gemm_8phase_kernel<0>, dim3(grid), dim3(512), 0, 0, a);
            CK(hipDeviceSynchronize());
            bad += check(a, A, B, bias, "warm");
        }
        hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dbias); hipFree(dts);
    }
    printf(bad ? "FAILED\n" : "ALL CHECKS PASSED\n");
    return bad != 0;
}
