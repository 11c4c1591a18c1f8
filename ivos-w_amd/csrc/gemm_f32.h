// Small-shape fp32 GEMM on the f32-input matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 fma chain).
// Serves the Brain's batched (non-recurrent) contractions: encoder_fc2, the input-side LSTM gates,
// decoder_fc1 and every dgrad / wgrad of the backward pass.  Shapes are tiny (M <= a few thousand,
// N,K <= 512), so the tile is 64x64x64 with generic operand strides; wgrad shapes (small MxN, long K)
// use split-K into slabs + a fixed-order reduce (deterministic, no atomics).
#pragma once
#include "common.h"

namespace ivosw {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmF32 {
    const float* A;   // A(m,k) = A[m*sam + k*sak]
    const float* B;   // B(k,n) = B[k*sbk + n*sbn]
    float* C;         // C[m*ldc + n]   (split-K: slab z at C + z*M*ldc)
    int M, N, K;
    long sam, sak, sbk, sbn;
    int ldc;
    const float* bias;   // [N] added in the epilogue, or nullptr
    const float* mask;   // same indexing as C; result *= (mask > 0), or nullptr
    int relu;            // max(.,0) epilogue
    int relu_a;          // max(.,0) applied to A on load
    int splitk;          // gridDim.z
};

// K-step 64: at these sizes a GEMM is a handful of dependent global-load round trips; 16-deep steps cost 8 of them for
// K = 128 (18 us per launch), 64-deep steps 2 with 16 loads per operand and thread in flight
constexpr int GB_M = 64, GB_N = 64, GB_K = 64, G_LD = 65;
constexpr int G_NI = GB_M * GB_K / 256;   // elements per operand per thread per K-step

__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmF32 g) {
    __shared__ float As[GB_K][G_LD];
    __shared__ float Bs[GB_K][G_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * GB_M, n0 = blockIdx.x * GB_N;
    const int kchunk = (((g.K + g.splitk - 1) / g.splitk) + GB_K - 1) / GB_K * GB_K;
    const int kbeg = blockIdx.z * kchunk;
    const int kend = min(g.K, kbeg + kchunk);
    const bool a_kfast = (g.sak == 1), b_kfast = (g.sbk == 1);

    float ra[G_NI], rb[G_NI];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < G_NI; ++i) {
            int m, k;
            if (a_kfast) { k = tid % GB_K; m = tid / GB_K + (256 / GB_K) * i; } else { m = tid & 63; k = (tid >> 6) + 4 * i; }
            const int gm = m0 + m, gk = k0 + k;
            float v = 0.f;
            if (gm < g.M && gk < kend) v = g.A[gm * g.sam + gk * g.sak];
            ra[i] = g.relu_a ? fmaxf(v, 0.f) : v;
            int n, kb;
            if (b_kfast) { kb = tid % GB_K; n = tid / GB_K + (256 / GB_K) * i; } else { n = tid & 63; kb = (tid >> 6) + 4 * i; }
            const int gn = n0 + n, gkb = k0 + kb;
            rb[i] = (gn < g.N && gkb < kend) ? g.B[gkb * g.sbk + gn * g.sbn] : 0.f;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < G_NI; ++i) {
            int m, k;
            if (a_kfast) { k = tid % GB_K; m = tid / GB_K + (256 / GB_K) * i; } else { m = tid & 63; k = (tid >> 6) + 4 * i; }
            As[k][m] = ra[i];
            int n, kb;
            if (b_kfast) { kb = tid % GB_K; n = tid / GB_K + (256 / GB_K) * i; } else { n = tid & 63; kb = (tid >> 6) + 4 * i; }
            Bs[kb][n] = rb[i];
        }
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    if (kbeg < kend) {
        fetch(kbeg);
        for (int k0 = kbeg; k0 < kend; k0 += GB_K) {
            __syncthreads();
            stash();
            __syncthreads();
            if (k0 + GB_K < kend) fetch(k0 + GB_K);
#pragma unroll
            for (int kk = 0; kk < GB_K / 2; ++kk) {
                const float a = As[kk * 2 + (lane >> 5)][wm * 32 + (lane & 31)];
                const float b = Bs[kk * 2 + (lane >> 5)][wn * 32 + (lane & 31)];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
        }
    }

    float* C = g.C + (size_t)blockIdx.z * g.M * g.ldc;
    const int col = n0 + wn * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < g.M && col < g.N) {
            float v = acc[r];
            if (g.splitk == 1) {
                if (g.bias) v += g.bias[col];
                if (g.mask) v = (g.mask[(size_t)row * g.ldc + col] > 0.f) ? v : 0.f;
                if (g.relu) v = fmaxf(v, 0.f);
            }
            C[(size_t)row * g.ldc + col] = v;
        }
    }
}

// out[i] = sum_z slabs[z*n + i], fixed order.
__global__ void splitk_reduce_kernel(const float* slabs, float* out, int n, int nslab) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int z = 0; z < nslab; ++z) s += slabs[(size_t)z * n + i];
    out[i] = s;
}

inline void launch_gemm_f32(const GemmF32& g, hipStream_t st) {
    dim3 grid((g.N + GB_N - 1) / GB_N, (g.M + GB_M - 1) / GB_M, g.splitk);
    hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, st, g);
}

// C[M,N] (contiguous) = A^T-style wgrad through split-K slabs.
inline void launch_gemm_f32_splitk(GemmF32 g, float* out, float* slabs, int nsplit, hipStream_t st) {
    g.C = slabs;
    g.splitk = nsplit;
    launch_gemm_f32(g, st);
    const int n = g.M * g.ldc;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, st, slabs, out, n, nsplit);
}

}  // namespace ivosw
