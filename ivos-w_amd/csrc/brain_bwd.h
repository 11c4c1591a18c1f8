// The backward tail of a DQN step after the BPTT recurrence, as register-resident fp32 MFMA kernels (tunable DQN_TAIL,
// default on).  Reference arithmetic: what autograd derives for Brain.forward (models/agent.py:33-64) under
// Agent.update_agent's loss (:128-155).
//
// The generic 64x64x64 LDS-staged GEMM (gemm_f32.h) ran this part as two grouped launches of 620 + 164 workgroups, 52 us for
// ~1.8 GFLOP (12 us of matrix-pipe time): every K-step paid a global -> register -> LDS -> register round trip and two
// barriers for 32 MFMAs per wave.  Here no operand goes through LDS:
//
//  * weight gradients  C[M,N] = sum_k A[k,:]^T B[k,:]  (dW_hh, dW_ih, dW2, dW3): both operands are row-major over k, so a lane
//    loads 16 bytes of A (four neighbouring m at its k) and 2 x 16 bytes of B (eight n) — with v_mfma_f32_16x16x4_f32's
//    "lane (i, kq) carries row/column i at k-slot kq" that is the operand of FOUR m-tiles and EIGHT n-tiles at once (tile j
//    holds the columns 4 i + j): 3 loads feed 32 MFMAs.  A workgroup owns a 64 x 128 tile of C over a K-slab, its four waves
//    take interleaved 4-row groups of the slab and add their accumulators through LDS in a fixed order at the end;
//    slabs are summed by the existing fixed-order reduction (deterministic, no atomics).
//  * the dgrad chain  de = (dG_fw + dG_bw) W_ih,  da1 = (de W2) . (a1 > 0)  per 16-row tile: the tile's de stays in LDS
//    between the two contractions, the waves split K and exchange partial sums through LDS.
//
// One launch runs the dgrad tiles next to the three weight gradients that do not depend on them.
#pragma once
#include "brain_fused.h"

namespace ivosw {

struct WgradJob {
    const float* A;     // A[k*lda + m]
    const float* A2;    // optional second operand summed on load (same indexing), or nullptr
    const float* B;     // B[k*ldb + n]
    float* C;           // slab z at C + z*M*N, element [m*N + n]
    int lda, ldb;
    int M, N, K;        // M % 64 == 0, N % 128 == 0
    int nslab;          // K split
};
struct DgradArgs {
    const float* dG;    // [rows,512] forward direction
    const float* dG2;   // [rows,512] backward direction (summed on load)
    const float* wih;   // [512,128]
    const float* w2;    // [128,128]
    const float* a1;    // [rows,128] relu output (mask)
    float* de;          // [rows,128]
    float* da1;         // [rows,128]
    int rows;
};
// column sums out[n] = sum_m A[m*ld + n] (bias gradients), optionally with the two sums weighted by x[m*2 + c] on top
// (encoder_fc1's weight gradient dW1[j][c] = sum_rows da1[row][j] x[row][c] beside db1); rows split nsplit ways into slabs
struct CsumJob {
    const float* A;
    const float* x;      // [M,2] or nullptr
    float* out;          // [N]            (slab z at out + z*N)
    float* outw;         // [N,2] when x   (slab z at outw + z*2N)
    int M, N, ld, nsplit;
};
constexpr int TAIL_MAX = 4;
struct TailGroup {
    DgradArgs dg;
    int n_dgrad;                 // workgroups [0, n_dgrad) run dgrad tiles (0: none)
    WgradJob w[TAIL_MAX];
    int first[TAIL_MAX + 1];     // first workgroup of weight-gradient job i (after the dgrad tiles); first[n] = their total
    int n;
    CsumJob cs[TAIL_MAX];
    int cs_first[TAIL_MAX + 1];  // first workgroup of column-sum job i (after the weight gradients)
    int n_cs;
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// ---------------------------------------------------------------- weight gradient tile
__device__ __forceinline__ void wgrad_tile(const WgradJob& jb, int local, float* lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, kq = lane >> 4;
    const int nmb = jb.M / 64, nnb = jb.N / 128;
    const int nb = local % nnb, mb = (local / nnb) % nmb, z = local / (nnb * nmb);
    const int m0 = mb * 64, n0 = nb * 128;
    const int per = ((jb.K + jb.nslab - 1) / jb.nslab + 15) / 16 * 16;
    const int kbeg = z * per, kend = min(jb.K, kbeg + per);
    const int nsteps = (kend - kbeg + 15) / 16;

    f32x4 acc[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[j][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float* ap = jb.A + m0 + 4 * i16;
    const bool has2 = jb.A2 != nullptr;
    const float* ap2 = (has2 ? jb.A2 : jb.A) + m0 + 4 * i16;
    const float* bp = jb.B + n0 + 4 * i16;
    // Branch-free fetch: rows past the slab read the slab's last row and are zeroed by a select when they are consumed (a
    // guarded load is a branch, and hipcc then drains vmcnt(0) at every block boundary: one memory round trip per step).
    struct Ops { float4 a, a2, b0, b1; };
    auto fetch = [&](int k, Ops& o) {
        const int kc = min(k, kend - 1);
        o.a = ld4(ap + (size_t)kc * jb.lda);
        o.a2 = ld4(ap2 + (size_t)kc * jb.lda);
        o.b0 = ld4(bp + (size_t)kc * jb.ldb);
        o.b1 = ld4(bp + (size_t)kc * jb.ldb + 64);
    };
    int k = kbeg + 4 * wave + kq;           // the waves take interleaved 4-row groups: step s covers rows kbeg + 16 s .. + 15
    // Two steps in flight ahead of the one being multiplied, in THREE fixed register sets (a rotating copy o1 = o2 reads the
    // load's destination, i.e. waits for it): the loop is unrolled by three, steps past the slab multiply zeros.
    Ops o[3];
    auto mul = [&](const Ops& q, int kk) {
        const float live = kk < kend ? 1.0f : 0.0f, live2 = has2 ? live : 0.0f;
        const float4 a = make_float4(fmaf(q.a2.x, live2, q.a.x * live), fmaf(q.a2.y, live2, q.a.y * live), fmaf(q.a2.z, live2, q.a.z * live),
                                     fmaf(q.a2.w, live2, q.a.w * live));
        const float4 b0 = q.b0, b1 = q.b1;
#define WG_ROW(j, av)                                                                        \
        acc[j][0] = MFMA16(av, b0.x, acc[j][0]); acc[j][1] = MFMA16(av, b0.y, acc[j][1]);    \
        acc[j][2] = MFMA16(av, b0.z, acc[j][2]); acc[j][3] = MFMA16(av, b0.w, acc[j][3]);    \
        acc[j][4] = MFMA16(av, b1.x, acc[j][4]); acc[j][5] = MFMA16(av, b1.y, acc[j][5]);    \
        acc[j][6] = MFMA16(av, b1.z, acc[j][6]); acc[j][7] = MFMA16(av, b1.w, acc[j][7]);
        WG_ROW(0, a.x) WG_ROW(1, a.y) WG_ROW(2, a.z) WG_ROW(3, a.w)
#undef WG_ROW
    };
    fetch(k, o[0]);
    fetch(k + 16, o[1]);
    for (int s = 0; s < nsteps; s += 3) {
        fetch(k + 32, o[2]);
        __builtin_amdgcn_sched_barrier(0);  // keep the loads above the MFMAs (hipcc otherwise sinks them to their first use)
        mul(o[0], k);
        fetch(k + 48, o[0]);
        __builtin_amdgcn_sched_barrier(0);
        mul(o[1], k + 16);
        fetch(k + 64, o[1]);
        __builtin_amdgcn_sched_barrier(0);
        mul(o[2], k + 32);
        k += 48;
    }

    // (w0 + w2) + (w1 + w3) through two 32 KB LDS images, register r of a lane at [r][lane]
    float* img = lds + (size_t)(wave & 1) * 8192;
    if (wave >= 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) img[((j * 8 + t) * 4 + r) * 64 + lane] = acc[j][t][r];
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[j][t][r] += img[((j * 8 + t) * 4 + r) * 64 + lane];
    }
    __syncthreads();
    if (wave == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) lds[((j * 8 + t) * 4 + r) * 64 + lane] = acc[j][t][r];
    }
    __syncthreads();
    if (wave == 0) {
        float* C = jb.C + (size_t)z * jb.M * jb.N;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* o = C + (size_t)(m0 + 4 * (4 * kq + r) + j) * jb.N + n0 + 4 * i16;
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    float4 v;
                    v.x = acc[j][4 * g + 0][r] + lds[((j * 8 + 4 * g + 0) * 4 + r) * 64 + lane];
                    v.y = acc[j][4 * g + 1][r] + lds[((j * 8 + 4 * g + 1) * 4 + r) * 64 + lane];
                    v.z = acc[j][4 * g + 2][r] + lds[((j * 8 + 4 * g + 2) * 4 + r) * 64 + lane];
                    v.w = acc[j][4 * g + 3][r] + lds[((j * 8 + 4 * g + 3) * 4 + r) * 64 + lane];
                    *reinterpret_cast<float4*>(o + 64 * g) = v;
                }
            }
    }
}

// ---------------------------------------------------------------- dgrad chain tile (16 rows)
constexpr int DG_LD = 128 + 4;
__device__ __forceinline__ void dgrad_tile(const DgradArgs& p, int tile, float* lds) {
    float (*part_s)[16][DG_LD] = reinterpret_cast<float (*)[16][DG_LD]>(lds);              // [4][16][132]
    float (*de_s)[DG_LD] = reinterpret_cast<float (*)[DG_LD]>(lds + 4 * 16 * DG_LD);       // [16][132]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, kq = lane >> 4;
    const int r0 = tile * 16;
    const int arow = min(r0 + i16, p.rows - 1);

    // W2's K-quarter of this wave (phase 2) and the a1 mask of the reduction are requested before anything else
    float4 w2r[2][4][2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* wp = p.w2 + (size_t)(32 * wave + 16 * c + 4 * kq + i) * 128 + 4 * i16;
            w2r[c][i][0] = ld4(wp);
            w2r[c][i][1] = ld4(wp + 64);
        }
    float4 mask[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i, row = min(r0 + (idx >> 5), p.rows - 1);
        mask[i] = ld4(p.a1 + (size_t)row * 128 + 4 * (idx & 31));
    }

    f32x4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    {   // de = dgx W_ih, this wave's K-quarter [128 w, 128 w + 128) of the 512 gate columns
        const float* pa = p.dG + (size_t)arow * 512 + 128 * wave + 4 * kq;
        const float* pa2 = p.dG2 + (size_t)arow * 512 + 128 * wave + 4 * kq;
        const float* pb = p.wih + (size_t)(128 * wave + 4 * kq) * 128 + 4 * i16;
        float4 an, an2, bn[4][2];
        auto fetch = [&](int c) {
            an = ld4(pa + 16 * c);
            an2 = ld4(pa2 + 16 * c);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bn[i][0] = ld4(pb + (size_t)(16 * c + i) * 128);
                bn[i][1] = ld4(pb + (size_t)(16 * c + i) * 128 + 64);
            }
        };
        fetch(0);
        for (int c = 0; c < 8; ++c) {
            const float4 a = add4(an, an2);
            float4 b[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) { b[i][0] = bn[i][0]; b[i][1] = bn[i][1]; }
            if (c + 1 < 8) fetch(c + 1);
            __builtin_amdgcn_sched_barrier(0);
#define DG_STEP(i, av)                                                                         \
            acc[0] = MFMA16(av, b[i][0].x, acc[0]); acc[1] = MFMA16(av, b[i][0].y, acc[1]);    \
            acc[2] = MFMA16(av, b[i][0].z, acc[2]); acc[3] = MFMA16(av, b[i][0].w, acc[3]);    \
            acc[4] = MFMA16(av, b[i][1].x, acc[4]); acc[5] = MFMA16(av, b[i][1].y, acc[5]);    \
            acc[6] = MFMA16(av, b[i][1].z, acc[6]); acc[7] = MFMA16(av, b[i][1].w, acc[7]);
            DG_STEP(0, a.x) DG_STEP(1, a.y) DG_STEP(2, a.z) DG_STEP(3, a.w)
        }
    }
    auto put_partial = [&]() {      // n-tile (g, j) holds columns 64 g + 4 i + j: four tiles = 16 contiguous bytes per lane
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int g = 0; g < 2; ++g)
                *reinterpret_cast<float4*>(&part_s[wave][4 * kq + r][64 * g + 4 * i16]) =
                    make_float4(acc[4 * g][r], acc[4 * g + 1][r], acc[4 * g + 2][r], acc[4 * g + 3][r]);
    };
    put_partial();
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i, row = idx >> 5, c4 = 4 * (idx & 31);
        const float4 v = add4(add4(ld4(&part_s[0][row][c4]), ld4(&part_s[1][row][c4])), add4(ld4(&part_s[2][row][c4]), ld4(&part_s[3][row][c4])));
        *reinterpret_cast<float4*>(&de_s[row][c4]) = v;
        if (r0 + row < p.rows) *reinterpret_cast<float4*>(p.de + (size_t)(r0 + row) * 128 + c4) = v;
    }
    __syncthreads();
    {   // da1 = (de W2) . (a1 > 0), this wave's K-quarter [32 w, 32 w + 32) of de's columns
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float4 a = ld4(&de_s[i16][32 * wave + 16 * c + 4 * kq]);
            const float4 (*b)[2] = w2r[c];
            DG_STEP(0, a.x) DG_STEP(1, a.y) DG_STEP(2, a.z) DG_STEP(3, a.w)
        }
#undef DG_STEP
    }
    put_partial();      // (part_s was last read before the barrier above)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i, row = idx >> 5, c4 = 4 * (idx & 31);
        float4 v = add4(add4(ld4(&part_s[0][row][c4]), ld4(&part_s[1][row][c4])), add4(ld4(&part_s[2][row][c4]), ld4(&part_s[3][row][c4])));
        const float4 m = mask[i];
        v = make_float4(m.x > 0.f ? v.x : 0.f, m.y > 0.f ? v.y : 0.f, m.z > 0.f ? v.z : 0.f, m.w > 0.f ? v.w : 0.f);
        if (r0 + row < p.rows) *reinterpret_cast<float4*>(p.da1 + (size_t)(r0 + row) * 128 + c4) = v;
    }
}

// ---------------------------------------------------------------- column-sum unit: 32 columns x one row split
__device__ __forceinline__ void csum_unit(const CsumJob& jb, int local, float* lds) {
    float (*red)[8][33] = reinterpret_cast<float (*)[8][33]>(lds);      // [3][8][33]
    const int ncb = (jb.N + 31) / 32;
    const int cb = local % ncb, z = local / ncb;
    const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int n = cb * 32 + c;
    const bool wx = jb.x != nullptr;
    const int per = ((jb.M + jb.nsplit - 1) / jb.nsplit + 7) / 8 * 8;
    const int mbeg = z * per, mend = min(jb.M, mbeg + per);
    float s[4] = {0.f, 0.f, 0.f, 0.f}, s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    if (n < jb.N) {
        int m = mbeg + rg;
        for (; m + 56 < mend; m += 64) {        // eight loads in flight per thread, four partial sums (fixed order)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = jb.A[(size_t)(m + 8 * u) * jb.ld + n];
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u & 3] += v[u];
            if (wx) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float2 xv = *reinterpret_cast<const float2*>(jb.x + (size_t)(m + 8 * u) * 2);
                    s0[u & 3] = fmaf(v[u], xv.x, s0[u & 3]);
                    s1[u & 3] = fmaf(v[u], xv.y, s1[u & 3]);
                }
            }
        }
        for (; m < mend; m += 8) {
            const float v = jb.A[(size_t)m * jb.ld + n];
            s[0] += v;
            if (wx) {
                const float2 xv = *reinterpret_cast<const float2*>(jb.x + (size_t)m * 2);
                s0[0] = fmaf(v, xv.x, s0[0]);
                s1[0] = fmaf(v, xv.y, s1[0]);
            }
        }
    }
    red[0][rg][c] = (s[0] + s[1]) + (s[2] + s[3]);
    red[1][rg][c] = (s0[0] + s0[1]) + (s0[2] + s0[3]);
    red[2][rg][c] = (s1[0] + s1[1]) + (s1[2] + s1[3]);
    __syncthreads();
    if (rg < 3 && n < jb.N && (rg == 0 || wx)) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += red[rg][i][c];
        if (rg == 0) jb.out[(size_t)z * jb.N + n] = t;
        else jb.outw[(size_t)z * 2 * jb.N + n * 2 + (rg - 1)] = t;
    }
}

// (256, 2): two workgroups per CU — the loop is bound by operand latency as much as by the matrix pipe, and with 128 accumulator
// registers a wave still fits in 256
__global__ __launch_bounds__(256, 2) void bwd_tail_kernel(TailGroup grp) {
    __shared__ __attribute__((aligned(16))) float lds[16384];      // 64 KB: two wave images (wgrad) / partials + de tile (dgrad)
    const int bid = blockIdx.x;
    if (bid < grp.n_dgrad) {
        if (bid < (grp.dg.rows + 15) / 16) dgrad_tile(grp.dg, bid, lds);
        return;
    }
    // The (M / 64) workgroups of one K-slab read the same rows of B: consecutive LOGICAL ids are made to share an XCD (the
    // hardware deals physical ids round-robin over the eight XCDs; n_dgrad is a multiple of 8), so B comes out of that XCD's
    // L2 for all but the first of them instead of crossing the fabric eight times.
    const int n_w = grp.first[TAIL_MAX];
    if (bid >= grp.n_dgrad + n_w) {
        const int cbid = bid - grp.n_dgrad - n_w;
        int q = 0;
#pragma unroll
        for (int i = 1; i < TAIL_MAX; ++i)
            if (i < grp.n_cs && cbid >= grp.cs_first[i]) q = i;
        csum_unit(grp.cs[q], cbid - grp.cs_first[q], lds);
        return;
    }
    const int wb = xcd_remap(bid - grp.n_dgrad, n_w);
    int p = 0;
#pragma unroll
    for (int i = 1; i < TAIL_MAX; ++i)
        if (i < grp.n && wb >= grp.first[i]) p = i;
    wgrad_tile(grp.w[p], wb - grp.first[p], lds);
}

}  // namespace ivosw
