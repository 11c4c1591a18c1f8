// Small-shape fp32 GEMM on the f32-input matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 fma chain).
// Serves the Brain's batched (non-recurrent) contractions: encoder_fc2, the input-side LSTM gates,
// decoder_fc1 and every dgrad / wgrad of the backward pass.  Shapes are tiny (M <= a few thousand,
// N,K <= 512), so the tile is 64x64x64 with generic operand strides; wgrad shapes (small MxN, long K)
// use split-K into slabs + a fixed-order reduce (deterministic, no atomics).
#pragma once
#include "adam.h"
#include "common.h"
#include <stdlib.h>

namespace ivosw {

int tune_get(const char* key, int dflt);   // capi.cpp
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmF32 {
    const float* A;   // A(m,k) = A[m*sam + k*sak]
    const float* A2;  // optional: A(m,k) += A2[m*sam + k*sak] on load (dgx = dG[fw] + dG[bw] without a pass of its own)
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
            if (gm < g.M && gk < kend) { v = g.A[gm * g.sam + gk * g.sak]; if (g.A2) v += g.A2[gm * g.sam + gk * g.sak]; }
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

// Same tile, 16-byte operand loads: when an operand is contiguous along K (sak / sbk == 1) or along M / N (sam / sbn
// == 1), 16-byte aligned and its extents are multiples of 4, a thread fetches 4 float4 per operand and K-step instead
// of 16 scalars (each with its own 64-bit address arithmetic) — the scalar kernel spends 3.6 us per K-step against
// 1 us of MFMA issue.  Operand modes are uniform per launch (amode / bmode: 0 = contiguous along K, 1 = along M / N).
__device__ __forceinline__ void gemm_vec_tile(const GemmF32& g, int amode, int bmode, int bx, int by, int bz, float (*As)[G_LD], float (*Bs)[G_LD]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = by * GB_M, n0 = bx * GB_N;
    const int kchunk = (((g.K + g.splitk - 1) / g.splitk) + GB_K - 1) / GB_K * GB_K;
    const int kbeg = bz * kchunk;
    const int kend = min(g.K, kbeg + kchunk);
    const int q16 = tid & 15, q4 = tid >> 4;      // float4 slot: 16 per 64-float line, lines q4 + 16 i

    // per-thread base pointers (K offset added per step); a line is a row (mode 0: 64 k of one m) or a k (mode 1: 64 m of one k)
    const float* ap[4];
    const float* bp[4];
    bool aok[4], bok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int line = q4 + 16 * i;
        if (amode == 0) { aok[i] = m0 + line < g.M; ap[i] = g.A + (size_t)(m0 + line) * g.sam + 4 * q16; }        // + k0
        else { aok[i] = m0 + 4 * q16 < g.M; ap[i] = g.A + (size_t)line * g.sak + m0 + 4 * q16; }                 // + k0 * sak
        if (bmode == 0) { bok[i] = n0 + line < g.N; bp[i] = g.B + (size_t)(n0 + line) * g.sbn + 4 * q16; }
        else { bok[i] = n0 + 4 * q16 < g.N; bp[i] = g.B + (size_t)line * g.sbk + n0 + 4 * q16; }
    }
    float4 ra[4], rb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int line = q4 + 16 * i;
            const bool ka = amode == 0 ? (k0 + 4 * q16 < kend) : (k0 + line < kend);
            const bool kb = bmode == 0 ? (k0 + 4 * q16 < kend) : (k0 + line < kend);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f), u = v;
            if (aok[i] && ka) {
                const size_t ko = amode == 0 ? (size_t)k0 : (size_t)k0 * g.sak;
                v = *reinterpret_cast<const float4*>(ap[i] + ko);
                if (g.A2) {
                    const float4 v2 = *reinterpret_cast<const float4*>(g.A2 + (ap[i] - g.A) + ko);
                    v = make_float4(v.x + v2.x, v.y + v2.y, v.z + v2.z, v.w + v2.w);
                }
            }
            if (bok[i] && kb) u = *reinterpret_cast<const float4*>(bp[i] + (bmode == 0 ? (size_t)k0 : (size_t)k0 * g.sbk));
            if (g.relu_a) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
            ra[i] = v;
            rb[i] = u;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int line = q4 + 16 * i;
            if (amode == 0) { As[4 * q16][line] = ra[i].x; As[4 * q16 + 1][line] = ra[i].y; As[4 * q16 + 2][line] = ra[i].z; As[4 * q16 + 3][line] = ra[i].w; }
            else { As[line][4 * q16] = ra[i].x; As[line][4 * q16 + 1] = ra[i].y; As[line][4 * q16 + 2] = ra[i].z; As[line][4 * q16 + 3] = ra[i].w; }
            if (bmode == 0) { Bs[4 * q16][line] = rb[i].x; Bs[4 * q16 + 1][line] = rb[i].y; Bs[4 * q16 + 2][line] = rb[i].z; Bs[4 * q16 + 3][line] = rb[i].w; }
            else { Bs[line][4 * q16] = rb[i].x; Bs[line][4 * q16 + 1] = rb[i].y; Bs[line][4 * q16 + 2] = rb[i].z; Bs[line][4 * q16 + 3] = rb[i].w; }
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
    float* C = g.C + (size_t)bz * g.M * g.ldc;
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

__global__ __launch_bounds__(256) void gemm_f32_vec_kernel(GemmF32 g, int amode, int bmode) {
    __shared__ float As[GB_K][G_LD];
    __shared__ float Bs[GB_K][G_LD];
    gemm_vec_tile(g, amode, bmode, blockIdx.x, blockIdx.y, blockIdx.z, As, Bs);
}

// Several independent GEMMs in ONE launch (the weight gradients of a DQN step: each is a few dozen tiles, far too small to
// fill the chip alone, and each launch boundary costs as much as the GEMM): workgroup -> (problem, tile) by a prefix table.
constexpr int GROUP_MAX = 6;
struct GemmGroup {
    GemmF32 g[GROUP_MAX];
    int amode[GROUP_MAX], bmode[GROUP_MAX];
    int first[GROUP_MAX + 1];          // first workgroup of problem i; first[n] = grid size
    int n;
};
__global__ __launch_bounds__(256) void gemm_f32_group_kernel(GemmGroup grp) {
    __shared__ float As[GB_K][G_LD];
    __shared__ float Bs[GB_K][G_LD];
    int p = 0;
#pragma unroll
    for (int i = 1; i < GROUP_MAX; ++i)
        if (i < grp.n && (int)blockIdx.x >= grp.first[i]) p = i;
    const GemmF32& g = grp.g[p];
    const int local = blockIdx.x - grp.first[p];
    const int nx = (g.N + GB_N - 1) / GB_N, ny = (g.M + GB_M - 1) / GB_M;
    gemm_vec_tile(g, grp.amode[p], grp.bmode[p], local % nx, (local / nx) % ny, local / (nx * ny), As, Bs);
}

// out[i] = sum_z slabs[z*n + i], fixed order.
__global__ void splitk_reduce_kernel(const float* slabs, float* out, int n, int nslab) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int z = 0; z < nslab; ++z) s += slabs[(size_t)z * n + i];
    out[i] = s;
}

// the same for up to six slab sets in one launch (blockIdx.y = which)
__global__ void splitk_reduce_group_kernel(ReduceGroup r) {
    const int w = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= r.n[w]) return;
    const float* sl = r.slabs[w] + i;
    const size_t n = r.n[w];
    const int ns = r.nslab[w];
    // four interleaved partial sums, eight loads in flight: the plain loop was a chain of L2 round trips (12 us at 64 slabs);
    // the order is fixed, so the result is deterministic
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int z = 0;
    for (; z + 8 <= ns; z += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = sl[(size_t)(z + u) * n];
        s0 += v[0]; s1 += v[1]; s2 += v[2]; s3 += v[3];
        s0 += v[4]; s1 += v[5]; s2 += v[6]; s3 += v[7];
    }
    for (; z < ns; ++z) s0 += sl[(size_t)z * n];
    r.out[w][i] = (s0 + s1) + (s2 + s3);
}

// operand mode for the 16-byte path: 0 = contiguous along K, 1 = contiguous along the M / N extent, -1 = not eligible
inline int gemm_vec_mode(const float* p, long s_ext, long s_k, int ext, int K) {
    if ((reinterpret_cast<uintptr_t>(p) & 15) != 0 || K % 4 != 0) return -1;
    if (s_k == 1 && s_ext % 4 == 0) return 0;
    if (s_ext == 1 && s_k % 4 == 0 && ext % 4 == 0) return 1;
    return -1;
}

inline void launch_gemm_f32(const GemmF32& g, hipStream_t st) {
    dim3 grid((g.N + GB_N - 1) / GB_N, (g.M + GB_M - 1) / GB_M, g.splitk);
    const int am = gemm_vec_mode(g.A, g.sam, g.sak, g.M, g.K), bm = gemm_vec_mode(g.B, g.sbn, g.sbk, g.N, g.K);
    static const bool vec_off = getenv("IVOSW_GEMM_SCALAR") != nullptr;
    const bool a2ok = !g.A2 || (reinterpret_cast<uintptr_t>(g.A2) & 15) == 0;
    if (am >= 0 && bm >= 0 && a2ok && !vec_off) hipLaunchKernelGGL(gemm_f32_vec_kernel, grid, dim3(256), 0, st, g, am, bm);
    else hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, st, g);
}

// C[M,N] (contiguous) = A^T-style wgrad through split-K slabs.
inline void launch_gemm_f32_splitk(GemmF32 g, float* out, float* slabs, int nsplit, hipStream_t st) {
    g.C = slabs;
    g.splitk = nsplit;
    launch_gemm_f32(g, st);
    const int n = g.M * g.ldc;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, st, slabs, out, n, nsplit);
}

// n GEMMs (every one eligible for the 16-byte path, else they are launched one by one) as one grouped launch
inline void launch_gemm_f32_group(const GemmF32* gs, int n, hipStream_t st) {
    GemmGroup grp{};
    bool ok = n <= GROUP_MAX;
    int total = 0;
    for (int i = 0; ok && i < n; ++i) {
        const GemmF32& g = gs[i];
        const int am = gemm_vec_mode(g.A, g.sam, g.sak, g.M, g.K), bm = gemm_vec_mode(g.B, g.sbn, g.sbk, g.N, g.K);
        ok = am >= 0 && bm >= 0 && (!g.A2 || (reinterpret_cast<uintptr_t>(g.A2) & 15) == 0);
        grp.g[i] = g; grp.amode[i] = am; grp.bmode[i] = bm; grp.first[i] = total;
        total += ((g.N + GB_N - 1) / GB_N) * ((g.M + GB_M - 1) / GB_M) * g.splitk;
    }
    if (!ok) {
        for (int i = 0; i < n; ++i) launch_gemm_f32(gs[i], st);
        return;
    }
    grp.n = n;
    for (int i = n; i <= GROUP_MAX; ++i) grp.first[i] = total;
    hipLaunchKernelGGL(gemm_f32_group_kernel, dim3(total), dim3(256), (size_t)tune_get("GEMM_DYNLDS", 0), st, grp);
}

}  // namespace ivosw
