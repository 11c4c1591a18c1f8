// Brain (bidirectional shared-cell LSTM Q-network) forward, Double-DQN loss and hand-derived backward.
//
// Reference arithmetic: Brain.forward (models/agent.py:33-64), Agent.update_agent (:128-155).
//
// Layout: the shared cell's input-side gates W_ih*e_t do not depend on the direction, so they are one
// batched GEMM over all (n,t); only h*W_hh^T is sequential.  The recurrence kernel keeps the whole
// 512x128 fp32 W_hh in VGPRs (one gate column per thread, 512 threads = the CU's register file) and
// walks the T steps with h broadcast from LDS: rows (sample x direction) map to workgroups, so a
// 128-sample minibatch fills all 256 CUs.  fp32 throughout (exact fma chains); MFMA is used for the
// dense batched contractions only (gemm_f32.h).
#ifdef IVOSW_PROBES
#include "../../include/ivosw_probe.h"
#endif
#include <mutex>

#include "gemm_f32.h"
#include "brain_fused.h"
#include "brain_bwd.h"

namespace ivosw {

// offsets (floats) into the parameter / gradient arena, state_dict order
constexpr int O_W1 = 0, O_B1 = 256, O_W2 = 384, O_B2 = 16768, O_WIH = 16896, O_WHH = 82432, O_W3 = 147968,
              O_B3 = 180736, O_W4 = 180864, O_B4 = 180992;
static_assert(O_B4 + 1 == IVOSW_BRAIN_NPARAMS, "arena layout");
constexpr int HD = 128;
int tune_get(const char* key, int dflt);   // capi.cpp

// ---------------------------------------------------------------- small kernels
// a1[row][j] = relu(W1[j,:].x[row] + b1[j])        (encoder_fc1 + relu, agent.py:49)
// rows [0, rows0) read x, rows [rows0, rows) read x2 (the policy pass runs [s'; s] as one batch without concatenating them)
__global__ void enc1_kernel(const float* __restrict__ prm, const float* __restrict__ x, const float* __restrict__ x2, int rows0, int rows,
                            float* __restrict__ a1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * HD) return;
    const int row = i >> 7, j = i & 127;
    const float* xr = row < rows0 ? x + (size_t)row * 2 : x2 + (size_t)(row - rows0) * 2;
    const float x0 = xr[0], x1 = xr[1];
    const float v = fmaf(x1, prm[O_W1 + j * 2 + 1], fmaf(x0, prm[O_W1 + j * 2], prm[O_B1 + j]));
    a1[i] = fmaxf(v, 0.f);
}

// q[row] = d1[row,:].w4 + b4                       (decoder_fc2, agent.py:61)
__global__ void dec2_kernel(const float* __restrict__ prm, const float* __restrict__ d1, int rows, float* __restrict__ q) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* r = d1 + (size_t)row * HD;
    float s = r[lane] * prm[O_W4 + lane] + r[lane + 64] * prm[O_W4 + lane + 64];
    s = wave_sum(s);
    if (lane == 0) q[row] = s + prm[O_B4];
}

// out[n] = sum_m A[m*ld + n]; grid = ceil(N/32), block = 1024 (32 cols x 32 row groups)
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ A, int M, int N, int ld, float* __restrict__ out) {
    __shared__ float red[32][33];
    const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int n = blockIdx.x * 32 + c;
    // four independent partial sums and 8 loads in flight per thread: the loop is a chain of global-load latencies
    // otherwise (M/32 = 100 dependent round trips, 15 us); the summation order stays fixed (deterministic)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (n < N) {
        int m = rg;
        for (; m + 224 < M; m += 256) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = A[(size_t)(m + 32 * u) * ld + n];
            s0 += v[0]; s1 += v[1]; s2 += v[2]; s3 += v[3];
            s0 += v[4]; s1 += v[5]; s2 += v[6]; s3 += v[7];
        }
        for (; m < M; m += 32) s0 += A[(size_t)m * ld + n];
    }
    const float s = (s0 + s1) + (s2 + s3);
    red[rg][c] = s;
    __syncthreads();
    if (rg == 0 && n < N) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) t += red[i][c];
        out[n] = t;
    }
}

// The same reduction for up to two matrices in one launch (blockIdx.y = which), each optionally with two weighted sums on top:
// out[n] = sum_m A[m][n]; outw[n*2 + c] = sum_m A[m][n] * x[m*2 + c] (c = 0, 1) — encoder_fc1's weight gradient
// dW1[j][c] = sum_rows da1[row][j] * x[row][c] is a weighted column sum of the matrix whose plain column sum is db1.
struct ColsumJob {
    const float* A;
    const float* x;      // [M,2] or nullptr
    float* out;          // [N]            (split over blockIdx.z: slab z at out + z * N; the caller reduces the slabs)
    float* outw;         // [N,2] when x   (slab z at outw + z * 2N)
    int M, N, ld;
    int nsplit;          // row splits of this job (0 = gridDim.z); workgroups with blockIdx.z >= nsplit have nothing to do
};
struct ColsumGroup { ColsumJob j[4]; };
__global__ __launch_bounds__(1024) void colsum_group_kernel(ColsumGroup grp) {
    __shared__ float red[3][32][33];
    const ColsumJob& jb = grp.j[blockIdx.y];
    const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int n = blockIdx.x * 32 + c;
    const int nsplit = jb.nsplit > 0 ? jb.nsplit : (int)gridDim.z;
    if (blockIdx.x * 32 >= jb.N || (int)blockIdx.z >= nsplit) return;
    float s[4] = {0.f, 0.f, 0.f, 0.f}, s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    const bool wx = jb.x != nullptr;
    // rows [mbeg, mend) of this split (multiples of 32 rows, so every row group sees whole strides)
    const int per = ((jb.M + nsplit - 1) / nsplit + 31) / 32 * 32;
    const int mbeg = blockIdx.z * per, mend = min(jb.M, mbeg + per);
    if (n < jb.N) {
        int m = mbeg + rg;
        for (; m + 224 < mend; m += 256) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = jb.A[(size_t)(m + 32 * u) * jb.ld + n];
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u & 3] += v[u];
            if (wx) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float2 xv = *reinterpret_cast<const float2*>(jb.x + (size_t)(m + 32 * u) * 2);
                    s0[u & 3] = fmaf(v[u], xv.x, s0[u & 3]);
                    s1[u & 3] = fmaf(v[u], xv.y, s1[u & 3]);
                }
            }
        }
        for (; m < mend; m += 32) {
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
        for (int i = 0; i < 32; ++i) t += red[rg][i][c];
        if (rg == 0) jb.out[(size_t)blockIdx.z * jb.N + n] = t;
        else jb.outw[(size_t)blockIdx.z * 2 * jb.N + n * 2 + (rg - 1)] = t;
    }
}

__global__ void add2_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = a[i] + b[i];
}

// ---------------------------------------------------------------- forward recurrence
struct LstmFwd {
    const float* whh;  // [512,128]
    const float* gx;   // [N,T,512]  W_ih*e_t
    float* hs;         // [N,T,256]  (fw | bw) h after consuming frame t
    float* gates;      // [2,N,T,512] post-activation i,f,g,o   (nullable)
    float* cs;         // [2,N,T,128]                           (nullable)
    float* hprev;      // [2,N,T,128] h before consuming frame t (nullable)
    int N, T;
    int keep_from;     // samples >= keep_from store gates/cs/hprev
    unsigned long long* ts;   // phase probe (ivosw_lstm_probe, tools/lstm_probe.py): s_memtime stamps of step T/2 per workgroup, or nullptr
};

template <int R>
__global__ __launch_bounds__(512) void lstm_fwd_kernel(LstmFwd p) {
    __shared__ __attribute__((aligned(16))) float h_s[R][HD];
    __shared__ float c_s[R][HD];
    __shared__ float g_s[R][4 * HD];
    const int tid = threadIdx.x;
    float w[HD];
    {
        const float4* wp = reinterpret_cast<const float4*>(p.whh + (size_t)tid * HD);
#pragma unroll
        for (int i = 0; i < HD / 4; ++i) {
            const float4 v = wp[i];
            w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
        }
    }
    if (tid < HD) {
#pragma unroll
        for (int r = 0; r < R; ++r) { h_s[r][tid] = 0.f; c_s[r][tid] = 0.f; }
    }
    __syncthreads();
    // g gate: tanh(x) = 2*sigmoid(2x) - 1, so all four gate types share one branch-free exp path
    const float gk = ((tid >> 7) == 2) ? 2.0f : 1.0f;
    const int Nk = p.N - p.keep_from;  // kept buffers are indexed by (sample - keep_from)
    const int row0 = blockIdx.x * R;
    // the input-side gate term of the NEXT step is fetched while this step computes: its global-load latency sat at the
    // head of every step's fma chain (the arithmetic order is unchanged: the chain still starts from that term)
    float a_nx[R];
    auto gx_fetch = [&](int s) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = min(row0 + r, 2 * p.N - 1);
            const int d = row / p.N, n = row - d * p.N;
            const int t = d ? p.T - 1 - s : s;
            a_nx[r] = p.gx[((size_t)n * p.T + t) * 512 + tid];
        }
    };
    gx_fetch(0);
    for (int s = 0; s < p.T; ++s) {
        float a_cur[R];
#pragma unroll
        for (int r = 0; r < R; ++r) a_cur[r] = a_nx[r];
        if (s + 1 < p.T) gx_fetch(s + 1);
        // rows one at a time: W_hh owns the register file, per-row state lives in LDS
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = row0 + r;
            if (row >= 2 * p.N) break;
            const int d = row / p.N, n = row - d * p.N;
            const int t = d ? p.T - 1 - s : s;
            float a = a_cur[r];
            // h == 0 at s == 0, so the first step needs no special case (fma(0,w,a) == a)
#pragma unroll
            for (int kc = 0; kc < HD / 4; ++kc) {
                const float4 h4 = *reinterpret_cast<const float4*>(&h_s[r][kc * 4]);
                a = fmaf(h4.x, w[4 * kc], a);
                a = fmaf(h4.y, w[4 * kc + 1], a);
                a = fmaf(h4.z, w[4 * kc + 2], a);
                a = fmaf(h4.w, w[4 * kc + 3], a);
            }
            const float sg = 1.0f / (1.0f + expf(-gk * a));
            const float act = fmaf(sg, gk, 1.0f - gk);
            g_s[r][tid] = act;
            if (p.gates && n >= p.keep_from)
                p.gates[(((size_t)d * Nk + (n - p.keep_from)) * p.T + t) * 512 + tid] = act;
        }
        __syncthreads();
        {
            // state update: 128 threads per row, the R rows side by side (they used to queue behind one another on the
            // first two waves while the other six idled)
            const int r = tid >> 7, j = tid & (HD - 1);
            const int row = row0 + r;
            if (r < R && row < 2 * p.N) {
                const int d = row / p.N, n = row - d * p.N;
                const int t = d ? p.T - 1 - s : s;
                const float gi = g_s[r][j], gf = g_s[r][HD + j], gg = g_s[r][2 * HD + j], go = g_s[r][3 * HD + j];
                const bool keep = p.gates && n >= p.keep_from;
                const size_t sidx = (((size_t)d * Nk + (n - p.keep_from)) * p.T + t) * HD + j;
                if (keep) p.hprev[sidx] = h_s[r][j];
                const float c = fmaf(gf, c_s[r][j], gi * gg);
                const float h = go * tanhf_(c);
                c_s[r][j] = c;
                h_s[r][j] = h;
                p.hs[((size_t)n * p.T + t) * 256 + d * HD + j] = h;
                if (keep) p.cs[sidx] = c;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- forward recurrence, quad layout
// The same recurrence with the work of a step laid out so that nothing but h crosses a thread: thread = (hidden unit j,
// K-quarter q), the four threads of a unit are the four lanes of a DPP quad.  A thread keeps W_hh[g*128 + j][32q .. 32q+31]
// for all four gates g (128 registers, like the kernel above), reads ITS quarter of h (8 ds_read_b128 instead of 32: the LDS
// return path was the bound), runs four independent 32-long fma chains, and the quad sums the partials with DPP adds — no
// LDS, no barrier.  Lane q then activates gate q (one exp per lane), the quad broadcasts the four activations back, and
// every lane updates the unit's cell state, which lives in a REGISTER.  One barrier per step (h is double-buffered),
// against two barriers + a 2 KB LDS round trip of the gates before.  Summation order: four K-quarters in k order, combined
// as (q0 + q1) + (q2 + q3), then + the input-side term.
struct QuadDpp {
    template <int CTRL>
    static __device__ __forceinline__ float mov(float v) {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
    }
};

template <int R>
__global__ __launch_bounds__(512) void lstm_fwd_quad_kernel(LstmFwd p) {
    constexpr int HQ = 36;                                   // a K-quarter of h: 32 floats + 4 of padding (the four quarters of a
    __shared__ __attribute__((aligned(16))) float h_s[2][R][4 * HQ];   // wave's reads fall on disjoint banks)
    const int tid = threadIdx.x, j = tid >> 2, q = tid & 3;
    float w[4][32];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4* wp = reinterpret_cast<const float4*>(p.whh + (size_t)(g * HD + j) * HD + 32 * q);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 v = wp[i];
            w[g][4 * i] = v.x; w[g][4 * i + 1] = v.y; w[g][4 * i + 2] = v.z; w[g][4 * i + 3] = v.w;
        }
    }
    for (int i = tid; i < R * 4 * HQ; i += 512) (&h_s[0][0][0])[i] = 0.f;
    __syncthreads();
    const float gk = (q == 2) ? 2.0f : 1.0f;                  // tanh(x) = 2*sigmoid(2x) - 1: one branch-free exp path for all gates
    const int Nk = p.N - p.keep_from;
    const int row0 = blockIdx.x * R;
    int dd[R], nn[R];
    bool ok[R], keep[R];
    float c[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = row0 + r;
        ok[r] = row < 2 * p.N;
        const int rc = ok[r] ? row : 2 * p.N - 1;
        dd[r] = rc / p.N;
        nn[r] = rc - dd[r] * p.N;
        keep[r] = ok[r] && p.gates && nn[r] >= p.keep_from;
        c[r] = 0.f;
    }
    float a_nx[R];
    auto gx_fetch = [&](int s) {                              // lane q fetches the input-side term of gate q
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int t = dd[r] ? p.T - 1 - s : s;
            a_nx[r] = p.gx[((size_t)nn[r] * p.T + t) * 512 + q * HD + j];
        }
    };
    gx_fetch(0);
    for (int s = 0; s < p.T; ++s) {
        const int cur = s & 1;
        float a_cur[R];
#pragma unroll
        for (int r = 0; r < R; ++r) a_cur[r] = a_nx[r];
        if (s + 1 < p.T) gx_fetch(s + 1);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int t = dd[r] ? p.T - 1 - s : s;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            const float* hq = &h_s[cur][r][q * HQ];
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                const float4 h4 = *reinterpret_cast<const float4*>(hq + 4 * kc);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    acc[g] = fmaf(h4.x, w[g][4 * kc], acc[g]);
                    acc[g] = fmaf(h4.y, w[g][4 * kc + 1], acc[g]);
                    acc[g] = fmaf(h4.z, w[g][4 * kc + 2], acc[g]);
                    acc[g] = fmaf(h4.w, w[g][4 * kc + 3], acc[g]);
                }
            }
            // quad all-reduce of the four partial sums (lanes q ^ 1, then q ^ 2): every lane ends with the same four totals
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                acc[g] += QuadDpp::mov<0xB1>(acc[g]);         // quad_perm [1,0,3,2]
                acc[g] += QuadDpp::mov<0x4E>(acc[g]);         // quad_perm [2,3,0,1]
            }
            const float mine = q == 0 ? acc[0] : q == 1 ? acc[1] : q == 2 ? acc[2] : acc[3];
            const float pre = mine + a_cur[r];
            const float sg = fast_rcp(1.0f + fast_exp(-gk * pre));
            const float act = fmaf(sg, gk, 1.0f - gk);        // lane q: gate q (i, f, g, o) of unit j
            const float gi = QuadDpp::mov<0x00>(act), gf = QuadDpp::mov<0x55>(act), gg = QuadDpp::mov<0xAA>(act), go = QuadDpp::mov<0xFF>(act);
            const float cn = fmaf(gf, c[r], gi * gg);
            const float h = go * fast_tanh(cn);
            c[r] = cn;
            if (ok[r]) {
                if (q == 0) {
                    h_s[cur ^ 1][r][(j >> 5) * HQ + (j & 31)] = h;
                    p.hs[((size_t)nn[r] * p.T + t) * 256 + dd[r] * HD + j] = h;
                }
                if (keep[r]) {
                    const size_t base = ((size_t)dd[r] * Nk + (nn[r] - p.keep_from)) * p.T + t;
                    p.gates[base * 512 + q * HD + j] = act;
                    if (q == 1) p.hprev[base * HD + j] = h_s[cur][r][(j >> 5) * HQ + (j & 31)];   // h before this frame
                    if (q == 2) p.cs[base * HD + j] = cn;
                }
            }
        }
        __syncthreads();                                      // h of step s+1 is complete; buffer `cur` is free for step s+2
    }
}

// ---------------------------------------------------------------- forward recurrence on the 4x4x1 matrix instruction
// The quad kernel above is VALU-bound (SQ counters: 340 VALU instructions per wave and step for two rows, 262 of them the fma
// chains, each a 4-cycle issue; two waves per SIMD keep the VALU 76 % busy).  v_mfma_f32_4x4x1_16b_f32 computes sixteen
// independent 4x4 outer products (K = 1) in 8 cycles: with block = hidden unit, the four B lanes of a block = that unit's four
// gate columns (the DPP quad of the kernel above) and the four A rows = FOUR rows of the batch, one instruction does for four
// rows what four v_fmac do for one — 128 instructions per wave and step (K = 128) for 4 rows x 64 columns against 524 fmas,
// exact fp32 (an fma chain in k order per accumulator).  A lane keeps its column's 128 weights in registers as before; h comes
// from LDS ([4 rows][128 + 4 pad]: lane (block, i) reads row i — four addresses per wave read, on disjoint banks); four
// accumulators (k mod 4) break the 128-deep dependent chain.  The result layout IS the quad layout: lane (b, j) holds gate j of
// unit b for the four rows in four registers, so activation, quad broadcast and state update are the code above per row.
// 4 rows per workgroup: 128 workgroups for the policy pass (2B = 256 samples x 2 directions), 64 for the target: one round.
typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lstm_fwd_mfma_body(const LstmFwd& p, const int bid) {
    constexpr int R = 4, HP = HD + 4;
    __shared__ __attribute__((aligned(16))) float h_s[2][R][HP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (p.ts && tid == 0) p.ts[(size_t)bid * 8 + 4] = __builtin_amdgcn_s_memtime();
    const int q = lane & 3, j = wave * 16 + (lane >> 2);      // gate q of hidden unit j; as an A lane: batch row q
    // (tried: the wave's rows through coalesced 8-KB reads + a wave-local LDS transpose instead of one row per lane — 18.1 k
    // against 15.9 k cycles for these loads: what they wait for is 256 KB per workgroup out of L2 lines that all 192
    // workgroups want at the same moment, not the access pattern)
    float w[HD];
    {
        const float4* wp = reinterpret_cast<const float4*>(p.whh + (size_t)(q * HD + j) * HD);
#pragma unroll
        for (int i = 0; i < HD / 4; ++i) {
            const float4 v = wp[i];
            w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
        }
    }
    for (int i = tid; i < 2 * R * HP; i += 512) (&h_s[0][0][0])[i] = 0.f;
    __syncthreads();
    const float gk = (q == 2) ? 2.0f : 1.0f;
    const int Nk = p.N - p.keep_from;
    const int row0 = bid * R;
    int dd[R], nn[R];
    bool ok[R], keep[R];
    float c[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = row0 + r;
        ok[r] = row < 2 * p.N;
        const int rc = ok[r] ? row : 2 * p.N - 1;
        dd[r] = rc / p.N;
        nn[r] = rc - dd[r] * p.N;
        keep[r] = ok[r] && p.gates && nn[r] >= p.keep_from;
        c[r] = 0.f;
    }
    // After the gate activations a 4 x 4 transpose inside the quad hands lane q the four gates of batch row q, so the cell
    // update (two of the four transcendentals per element, the state register, the h / c stores) is done ONCE per (unit, row)
    // by that lane instead of by every lane of the quad for every row: 64 VALU instructions (10 transcendental) per step and
    // lane instead of 88 (16) in a phase that is VALU-throughput-bound.  This lane's row:
    const bool my_ok = q == 0 ? ok[0] : q == 1 ? ok[1] : q == 2 ? ok[2] : ok[3];
    const bool my_keep = q == 0 ? keep[0] : q == 1 ? keep[1] : q == 2 ? keep[2] : keep[3];
    const int my_d = q == 0 ? dd[0] : q == 1 ? dd[1] : q == 2 ? dd[2] : dd[3];
    const int my_n = q == 0 ? nn[0] : q == 1 ? nn[1] : q == 2 ? nn[2] : nn[3];
    const bool q_odd = (q & 1) != 0, q_hi = (q & 2) != 0;
    float cq = 0.f;                                            // cell state of (unit j, row q)
    float a_nx[R];
    auto gx_fetch = [&](int s) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int t = dd[r] ? p.T - 1 - s : s;
            a_nx[r] = p.gx[((size_t)nn[r] * p.T + t) * 512 + q * HD + j];
        }
    };
    gx_fetch(0);
    if (p.ts && tid == 0) p.ts[(size_t)bid * 8 + 5] = __builtin_amdgcn_s_memtime() + (unsigned long long)(w[0] == 1.2345e33f);
    for (int s = 0; s < p.T; ++s) {
        const int cur = s & 1;
        float a_cur[R];
#pragma unroll
        for (int r = 0; r < R; ++r) a_cur[r] = a_nx[r];
        if (s + 1 < p.T) gx_fetch(s + 1);
        const bool probe = p.ts && tid == 0 && s == p.T / 2;
        if (probe) p.ts[(size_t)bid * 8 + 0] = __builtin_amdgcn_s_memtime();
        f32x4v acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4v{0.f, 0.f, 0.f, 0.f};
        const float* hrow = &h_s[cur][q][0];                   // A operand: this lane supplies h[row q][k]
        // h in four batches of eight 16-byte reads, a batch ahead of the MFMAs that use it (left to itself hipcc keeps two reads
        // in flight: ~70 cycles of MFMAs against an LDS round trip of twice that)
        float4 hb[2][8];
#pragma unroll
        for (int i = 0; i < 8; ++i) hb[0][i] = *reinterpret_cast<const float4*>(hrow + 4 * i);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g + 1 < 4) {
#pragma unroll
                for (int i = 0; i < 8; ++i) hb[(g + 1) & 1][i] = *reinterpret_cast<const float4*>(hrow + 4 * (8 * (g + 1) + i));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int kc = 8 * g + i;
                const float4 h4 = hb[g & 1][i];
                acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(h4.x, w[4 * kc], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(h4.y, w[4 * kc + 1], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(h4.z, w[4 * kc + 2], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(h4.w, w[4 * kc + 3], acc[3], 0, 0, 0);
            }
        }
        if (probe) p.ts[(size_t)bid * 8 + 1] = __builtin_amdgcn_s_memtime() + (unsigned long long)(acc[0][0] == 1.2345e33f);   // (waits for the MFMAs)
        // (VALU-throughput-bound with two waves per SIMD — ~650 issue cycles each, 16 transcendentals at quarter rate among
        // them — not latency-bound: computing the four rows as one branch-free block ahead of the stores changed nothing)
        float m0[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float pre = ((acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r])) + a_cur[r];
            const float sg = fast_rcp(1.0f + fast_exp(-gk * pre));
            m0[r] = fmaf(sg, gk, 1.0f - gk);                  // lane q: gate q (i, f, g, o) of unit j, batch row r
        }
        // transpose: m2[k] of lane q = gate k of row q (2 x 2 blocks across lane ^ 1, then blocks across lane ^ 2)
        float m1[R], m2[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float x = QuadDpp::mov<0xB1>(m0[r ^ 1]);     // quad_perm [1,0,3,2]
            m1[r] = (((r & 1) != 0) == q_odd) ? m0[r] : x;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float x = QuadDpp::mov<0x4E>(m1[r ^ 2]);     // quad_perm [2,3,0,1]
            m2[r] = (((r & 2) != 0) == q_hi) ? m1[r] : x;
        }
        cq = fmaf(m2[1], cq, m2[0] * m2[2]);
        const float hq = m2[3] * fast_tanh(cq);
        const int tq = my_d ? p.T - 1 - s : s;
        if (my_ok) {
            const float hold = h_s[cur][q][j];                 // h before this frame
            h_s[cur ^ 1][q][j] = hq;
            p.hs[((size_t)my_n * p.T + tq) * 256 + my_d * HD + j] = hq;
            if (my_keep) {
                const size_t base = ((size_t)my_d * Nk + (my_n - p.keep_from)) * p.T + tq;
                p.hprev[base * HD + j] = hold;
                p.cs[base * HD + j] = cq;
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (keep[r]) {
                const int t = dd[r] ? p.T - 1 - s : s;
                const size_t base = ((size_t)dd[r] * Nk + (nn[r] - p.keep_from)) * p.T + t;
                p.gates[base * 512 + q * HD + j] = m0[r];
            }
        }
        if (probe) p.ts[(size_t)bid * 8 + 2] = __builtin_amdgcn_s_memtime();
        __syncthreads();
        if (probe) p.ts[(size_t)bid * 8 + 3] = __builtin_amdgcn_s_memtime();
    }
    if (p.ts && tid == 0) p.ts[(size_t)bid * 8 + 6] = __builtin_amdgcn_s_memtime();
}

__global__ __launch_bounds__(512) void lstm_fwd_mfma_kernel(LstmFwd p) { lstm_fwd_mfma_body(p, blockIdx.x); }

// the policy pass and the target pass of a DQN step in ONE launch (they used to run side by side on two streams, with an
// event fork / join around them): workgroups [0, first1) run job 0, the rest job 1
struct LstmFwdPair {
    LstmFwd j[2];
    int first1;
};
__global__ __launch_bounds__(512) void lstm_fwd_mfma_pair_kernel(LstmFwdPair pr) {
    const int which = (int)blockIdx.x >= pr.first1;
    lstm_fwd_mfma_body(pr.j[which], (int)blockIdx.x - (which ? pr.first1 : 0));
}

// ---------------------------------------------------------------- recurrence + decoder in one kernel ("sample-pair" layout)
// A workgroup owns BOTH directions of TWO samples (rows: (fw, n0) (bw, n0) (fw, n1) (bw, n1)) instead of four rows of one
// direction, so when the recurrence ends it holds [h_fw | h_bw] of its samples' frames — the decoder's input — in LDS and
// runs relu -> decoder_fc1 -> relu -> decoder_fc2 on them itself (the W_hh registers are free by then): the decoder launch
// and the round trip of hs through memory are gone.  Same recurrence arithmetic as lstm_fwd_mfma_kernel; the decoder is
// dec_fused_kernel's (16x16x4 MFMAs, a wave owns 16 output columns).  hs goes to memory only for the samples a consumer
// reads them from (the loss rows of the policy's `state` half).
struct FwdMega {
    const float* prm;
    const float* gx;            // [N,T,512]
    float* hs;                  // [N,T,256], written for samples >= hs_from
    float *gates, *cs, *hprev;  // kept for samples >= keep_from (nullable)
    float* d1;                  // [N*T,128], written for samples >= keep_from
    float* q;                   // [N*T]
    int N, T, keep_from, hs_from;
    int dbg;
};
struct FwdMegaPair {
    FwdMega j[2];
    int first1;
};
constexpr int MEGA_HLD = 2 * HD + 4;                                  // padded [h_fw | h_bw] row in LDS
static size_t mega_lds_bytes(int T) { return (size_t)2 * T * MEGA_HLD * sizeof(float); }      // dynamic part: the pair's hs rows

__global__ __launch_bounds__(512) void brain_fwd_mega_kernel(FwdMegaPair pr) {
    extern __shared__ __attribute__((aligned(16))) float mega_lds[];
    constexpr int R = 4, HP = HD + 4;
    const int which = (int)blockIdx.x >= pr.first1;
    const FwdMega& p = pr.j[which];
    const int bid = (int)blockIdx.x - (which ? pr.first1 : 0);
    __shared__ __attribute__((aligned(16))) float h_s[2][R][HP];
    __shared__ float q_s[8][64];
    float (*hs_l)[MEGA_HLD] = reinterpret_cast<float (*)[MEGA_HLD]>(mega_lds);     // [2T][260]: row a*T + t of sample a
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane & 3, j = wave * 16 + (lane >> 2);      // gate q of hidden unit j; after the transpose: row q
    const float* __restrict__ prm = p.prm;
    float w[HD];
    {
        const float4* wp = reinterpret_cast<const float4*>(prm + O_WHH + (size_t)(q * HD + j) * HD);
#pragma unroll
        for (int i = 0; i < HD / 4; ++i) {
            const float4 v = wp[i];
            w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
        }
    }
    for (int i = tid; i < 2 * R * HP; i += 512) (&h_s[0][0][0])[i] = 0.f;
    __syncthreads();
    const float gk = (q == 2) ? 2.0f : 1.0f;
    const int Nk = p.N - p.keep_from;
    int dd[R], nn[R];
    bool ok[R], keep[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int n = 2 * bid + (r >> 1);
        ok[r] = n < p.N;
        nn[r] = ok[r] ? n : p.N - 1;
        dd[r] = r & 1;
        keep[r] = ok[r] && p.gates && nn[r] >= p.keep_from;
    }
    const int my_a = q >> 1, my_d = q & 1;                    // this lane's row after the transpose: sample my_a of the pair, direction my_d
    const int my_n = 2 * bid + my_a;
    const bool my_ok = my_n < p.N;
    const bool my_keep = my_ok && p.gates && my_n >= p.keep_from;
    const bool my_hs = my_ok && my_n >= p.hs_from;
    const bool q_odd = (q & 1) != 0, q_hi = (q & 2) != 0;
    float cq = 0.f;
    float a_nx[R];
    auto gx_fetch = [&](int s) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int t = dd[r] ? p.T - 1 - s : s;
            a_nx[r] = p.gx[((size_t)nn[r] * p.T + t) * 512 + q * HD + j];
        }
    };
    gx_fetch(0);
    for (int s = 0; s < p.T; ++s) {
        const int cur = s & 1;
        float a_cur[R];
#pragma unroll
        for (int r = 0; r < R; ++r) a_cur[r] = a_nx[r];
        if (s + 1 < p.T) gx_fetch(s + 1);
        f32x4v acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4v{0.f, 0.f, 0.f, 0.f};
        const float* hrow = &h_s[cur][q][0];
        float4 hb[2][8];
#pragma unroll
        for (int i = 0; i < 8; ++i) hb[0][i] = *reinterpret_cast<const float4*>(hrow + 4 * i);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g + 1 < 4) {
#pragma unroll
                for (int i = 0; i < 8; ++i) hb[(g + 1) & 1][i] = *reinterpret_cast<const float4*>(hrow + 4 * (8 * (g + 1) + i));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int kc = 8 * g + i;
                const float4 h4 = hb[g & 1][i];
                acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(h4.x, w[4 * kc], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(h4.y, w[4 * kc + 1], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(h4.z, w[4 * kc + 2], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(h4.w, w[4 * kc + 3], acc[3], 0, 0, 0);
            }
        }
        float m0[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float pre = ((acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r])) + a_cur[r];
            const float sg = fast_rcp(1.0f + fast_exp(-gk * pre));
            m0[r] = fmaf(sg, gk, 1.0f - gk);
        }
        float m1[R], m2[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float x = QuadDpp::mov<0xB1>(m0[r ^ 1]);
            m1[r] = (((r & 1) != 0) == q_odd) ? m0[r] : x;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float x = QuadDpp::mov<0x4E>(m1[r ^ 2]);
            m2[r] = (((r & 2) != 0) == q_hi) ? m1[r] : x;
        }
        cq = fmaf(m2[1], cq, m2[0] * m2[2]);
        const float hq = m2[3] * fast_tanh(cq);
        const int tq = my_d ? p.T - 1 - s : s;
        if (my_ok) {
            const float hold = h_s[cur][q][j];
            h_s[cur ^ 1][q][j] = hq;
            hs_l[my_a * p.T + tq][my_d * HD + j] = hq;
            if (my_hs) p.hs[((size_t)my_n * p.T + tq) * 256 + my_d * HD + j] = hq;
            if (my_keep) {
                const size_t base = ((size_t)my_d * Nk + (my_n - p.keep_from)) * p.T + tq;
                p.hprev[base * HD + j] = hold;
                p.cs[base * HD + j] = cq;
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (keep[r]) {
                const int t = dd[r] ? p.T - 1 - s : s;
                const size_t base = ((size_t)dd[r] * Nk + (nn[r] - p.keep_from)) * p.T + t;
                p.gates[base * 512 + q * HD + j] = m0[r];
            }
        }
        __syncthreads();
    }
    // ---- decoder on the pair's 2 T rows (row m = a * T + t), 64 rows at a time; wave w owns output columns [16 w, 16 w + 16)
    if (p.dbg) return;
    const int m16 = lane & 15, kq = lane >> 4;
    const int M = (2 * bid + 1 < p.N ? 2 : 1) * p.T;
    float4 w3r[16];
    {
        const float* wb = prm + O_W3 + (size_t)(16 * wave + m16) * 256 + 4 * kq;
#pragma unroll
        for (int c = 0; c < 16; ++c) w3r[c] = *reinterpret_cast<const float4*>(wb + 16 * c);
    }
    const float b3v = prm[O_B3 + 16 * wave + m16], w4v = prm[O_W4 + 16 * wave + m16], b4v = prm[O_B4];
    for (int g0 = 0; g0 < M; g0 += 64) {
        f32x4 acc[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* arow[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) arow[mt] = &hs_l[min(g0 + mt * 16 + m16, M - 1)][4 * kq];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            float4 a[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const float4 v = *reinterpret_cast<const float4*>(arow[mt] + 16 * c);
                a[mt] = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
            }
            const float4 b = w3r[c];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt] = MFMA16(a[mt].x, b.x, acc[mt]);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt] = MFMA16(a[mt].y, b.y, acc[mt]);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt] = MFMA16(a[mt].z, b.z, acc[mt]);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt] = MFMA16(a[mt].w, b.w, acc[mt]);
        }
        const int col = 16 * wave + m16;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = g0 + mt * 16 + 4 * kq + r;
                const float v = fmaxf(acc[mt][r] + b3v, 0.f);
                if (m < M) {
                    const int n = 2 * bid + m / p.T, t = m - (m / p.T) * p.T;
                    if (p.d1 && n >= p.keep_from) p.d1[((size_t)n * p.T + t) * HD + col] = v;
                }
                const float sum = row16_sum(v * w4v);
                if (m16 == 0) q_s[wave][mt * 16 + 4 * kq + r] = sum;
            }
        __syncthreads();
        if (tid < 64 && g0 + tid < M) {
            const int m = g0 + tid, a = m / p.T;
            p.q[((size_t)(2 * bid + a)) * p.T + (m - a * p.T)] =
                (((q_s[0][tid] + q_s[1][tid]) + (q_s[2][tid] + q_s[3][tid])) + ((q_s[4][tid] + q_s[5][tid]) + (q_s[6][tid] + q_s[7][tid]))) + b4v;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- backward recurrence (BPTT)
struct LstmBwd {
    const float* whh;    // [512,128]
    const float* gates;  // [2,Nk,T,512]   (Nk kept samples, sample index relative to keep_from)
    const float* cs;     // [2,Nk,T,128]
    const float* dhc;    // [Nk,256] dL/d(h_fw|h_bw) at frame action[n] (the only frame with loss)
    const int64_t* action;  // [Nk]
    float* dG;           // [2,Nk,T,512] dL/d(gate pre-activation)
    int N, T;            // N = Nk
    unsigned long long* ts;   // phase probe (ivosw_lstm_probe): stamps of the step 3 below the row's first, per workgroup, or nullptr
};

template <int R>
__global__ __launch_bounds__(512) void lstm_bwd_kernel(LstmBwd p) {
    __shared__ __attribute__((aligned(16))) float dp_s[R][4 * HD];
    __shared__ float part_s[R][4][HD];
    const int tid = threadIdx.x, k = tid & 127, part = tid >> 7;
    float w[HD];
#pragma unroll
    for (int j = 0; j < HD; ++j) w[j] = p.whh[(size_t)(part * HD + j) * HD + k];
    int dn[R], nn[R], s0[R];
    bool ok[R];
    int smax = -1;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = blockIdx.x * R + r;
        ok[r] = row < 2 * p.N;
        dn[r] = ok[r] ? row / p.N : 0;
        nn[r] = ok[r] ? row % p.N : 0;
        int a = ok[r] ? (int)p.action[nn[r]] : 0;
        a = min(max(a, 0), p.T - 1);
        s0[r] = ok[r] ? (dn[r] ? p.T - 1 - a : a) : -1;  // step at which frame `action` is consumed
        smax = max(smax, s0[r]);
    }
    // steps after the loss frame carry no gradient
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (!ok[r]) continue;
        for (int s = s0[r] + 1; s < p.T; ++s) {
            const int t = dn[r] ? p.T - 1 - s : s;
            p.dG[(((size_t)dn[r] * p.N + nn[r]) * p.T + t) * 512 + tid] = 0.f;
        }
    }
    float dh_rec[R], dc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { dh_rec[r] = 0.f; dc[r] = 0.f; }

    for (int s = smax; s >= 0; --s) {
        if (tid < HD) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float di = 0.f, df = 0.f, dg = 0.f, dO = 0.f;
                if (ok[r] && s <= s0[r]) {
                    const int t = dn[r] ? p.T - 1 - s : s;
                    const size_t rt = ((size_t)dn[r] * p.N + nn[r]) * p.T + t;
                    const float* gp = p.gates + rt * 512;
                    const float gi = gp[tid], gf = gp[HD + tid], gg = gp[2 * HD + tid], go = gp[3 * HD + tid];
                    const float cc = p.cs[rt * HD + tid];
                    const int tprev = dn[r] ? t + 1 : t - 1;
                    const float cprev = (s > 0) ? p.cs[(((size_t)dn[r] * p.N + nn[r]) * p.T + tprev) * HD + tid] : 0.f;
                    float dh = dh_rec[r];
                    if (s == s0[r]) dh += p.dhc[(size_t)nn[r] * 256 + dn[r] * HD + tid];
                    const float tc = tanhf_(cc);
                    dc[r] = fmaf(dh * go, 1.f - tc * tc, dc[r]);
                    di = dc[r] * gg * gi * (1.f - gi);
                    df = dc[r] * cprev * gf * (1.f - gf);
                    dg = dc[r] * gi * (1.f - gg * gg);
                    dO = dh * tc * go * (1.f - go);
                    dc[r] *= gf;
                    float* o = p.dG + rt * 512;
                    o[tid] = di; o[HD + tid] = df; o[2 * HD + tid] = dg; o[3 * HD + tid] = dO;
                }
                dp_s[r][tid] = di; dp_s[r][HD + tid] = df; dp_s[r][2 * HD + tid] = dg; dp_s[r][3 * HD + tid] = dO;
            }
        }
        __syncthreads();
        if (s > 0) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float a = 0.f;
#pragma unroll
                for (int jc = 0; jc < HD / 4; ++jc) {
                    const float4 d4 = *reinterpret_cast<const float4*>(&dp_s[r][part * HD + jc * 4]);
                    a = fmaf(d4.x, w[4 * jc], a);
                    a = fmaf(d4.y, w[4 * jc + 1], a);
                    a = fmaf(d4.z, w[4 * jc + 2], a);
                    a = fmaf(d4.w, w[4 * jc + 3], a);
                    if ((jc & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                }
                part_s[r][part][k] = a;
            }
        }
        __syncthreads();
        if (s > 0 && tid < HD) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                dh_rec[r] = (part_s[r][0][tid] + part_s[r][1][tid]) + (part_s[r][2][tid] + part_s[r][3][tid]);
        }
    }
}

// ---------------------------------------------------------------- backward recurrence, quad layout
// Thread = (hidden unit k, gate q); the four threads of a unit are a DPP quad.  Lane q computes dL/d(pre-activation) of
// gate q of unit k (the kernel above leaves that to 128 of its 512 threads, each behind a chain of global loads), keeps
// W_hh[q*128 + j][k] (j = 0..127) in registers, and the quad sums the four gates' contributions to dL/dh_prev[k] with DPP
// adds: one barrier per step (the gate gradients are double-buffered in LDS), no partial-sum round trip.  The
// activations of step s-1 are requested before the matvec of step s.  Same arithmetic per element as lstm_bwd_kernel;
// the 512-long dot product is summed as four interleaved 32-long chains per gate, gates combined (i + f) + (g + o).
template <int R>
__global__ __launch_bounds__(512) void lstm_bwd_quad_kernel(LstmBwd p) {
    constexpr int GQ = HD + 4;                               // one gate's 128 gradients + 4 floats of padding (bank spread)
    __shared__ __attribute__((aligned(16))) float dp_s[2][R][4 * GQ];
    const int tid = threadIdx.x, k = tid >> 2, q = tid & 3;
    float w[HD];
#pragma unroll
    for (int j = 0; j < HD; ++j) w[j] = p.whh[(size_t)(q * HD + j) * HD + k];
    int dn[R], nn[R], s0[R];
    bool ok[R];
    int smax = -1;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = blockIdx.x * R + r;
        ok[r] = row < 2 * p.N;
        dn[r] = ok[r] ? row / p.N : 0;
        nn[r] = ok[r] ? row % p.N : 0;
        int a = ok[r] ? (int)p.action[nn[r]] : 0;
        a = min(max(a, 0), p.T - 1);
        s0[r] = ok[r] ? (dn[r] ? p.T - 1 - a : a) : -1;      // step at which frame `action` is consumed
        smax = max(smax, s0[r]);
    }
    // steps after the loss frame carry no gradient
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (!ok[r]) continue;
        for (int s = s0[r] + 1; s < p.T; ++s) {
            const int t = dn[r] ? p.T - 1 - s : s;
            p.dG[(((size_t)dn[r] * p.N + nn[r]) * p.T + t) * 512 + q * HD + k] = 0.f;
        }
    }
    auto rt_of = [&](int r, int s) {                          // (direction, sample, frame) row of the kept activations at step s
        const int sc = min(max(s, 0), p.T - 1);
        const int t = dn[r] ? p.T - 1 - sc : sc;
        return ((size_t)dn[r] * p.N + nn[r]) * p.T + t;
    };
    // The kept activations of a step (its gate, its cell state, the cell state before it) are requested TWO steps ahead into
    // a ring of three register sets with fixed roles — the loop is unrolled by three, so nothing is copied: a rotating copy
    // (g_cur = g_nx) reads the load's destination, i.e. waits for a load that was issued at the top of the same step.
    struct Kept { float g, c, cp; };
    Kept ring[3][R];
    auto fetch = [&](int s, Kept (&dst)[R]) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            dst[r].g = p.gates[rt_of(r, s) * 512 + q * HD + k];
            dst[r].c = p.cs[rt_of(r, s) * HD + k];
            dst[r].cp = p.cs[rt_of(r, s - 1) * HD + k];
        }
    };
    float dh_rec[R], dc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { dh_rec[r] = 0.f; dc[r] = 0.f; }
    fetch(smax, ring[0]);
    fetch(smax - 1, ring[1]);
    if (p.ts && tid == 0) p.ts[(size_t)blockIdx.x * 8 + 4] = __builtin_amdgcn_s_memtime() + (unsigned long long)(w[0] == 1.2345e33f);
    auto step = [&](int s, Kept (&cur)[R], Kept (&ahead)[R]) {
        const int buf = s & 1;
        const bool probe = p.ts && tid == 0 && s == smax - 3;
        if (probe) p.ts[(size_t)blockIdx.x * 8 + 0] = __builtin_amdgcn_s_memtime();
        fetch(s - 2, ahead);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool act = ok[r] && s <= s0[r];
            const float gq = cur[r].g;
            const float gi = QuadDpp::mov<0x00>(gq), gf = QuadDpp::mov<0x55>(gq), gg = QuadDpp::mov<0xAA>(gq), go = QuadDpp::mov<0xFF>(gq);
            const float cc = cur[r].c;
            const float cprev = (s > 0) ? cur[r].cp : 0.f;
            float dh = dh_rec[r];
            if (s == s0[r]) dh += p.dhc[(size_t)nn[r] * 256 + dn[r] * HD + k];
            const float tc = fast_tanh(cc);
            const float dcn = fmaf(dh * go, 1.f - tc * tc, dc[r]);
            const float di = dcn * gg * gi * (1.f - gi);
            const float df = dcn * cprev * gf * (1.f - gf);
            const float dg = dcn * gi * (1.f - gg * gg);
            const float dO = dh * tc * go * (1.f - go);
            float mine = q == 0 ? di : q == 1 ? df : q == 2 ? dg : dO;
            mine = act ? mine : 0.f;
            if (act) {
                dc[r] = dcn * gf;
                p.dG[rt_of(r, s) * 512 + q * HD + k] = mine;
            }
            dp_s[buf][r][q * GQ + k] = mine;
        }
        if (probe) p.ts[(size_t)blockIdx.x * 8 + 1] = __builtin_amdgcn_s_memtime();
        __syncthreads();                                       // gate gradients of step s complete; buffer buf^1 (step s+1) is free
        if (probe) p.ts[(size_t)blockIdx.x * 8 + 2] = __builtin_amdgcn_s_memtime();
        if (s > 0) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                const float* dq = &dp_s[buf][r][q * GQ];
#pragma unroll
                for (int jc = 0; jc < HD / 4; ++jc) {
                    const float4 d4 = *reinterpret_cast<const float4*>(dq + 4 * jc);
                    a0 = fmaf(d4.x, w[4 * jc], a0);
                    a1 = fmaf(d4.y, w[4 * jc + 1], a1);
                    a2 = fmaf(d4.z, w[4 * jc + 2], a2);
                    a3 = fmaf(d4.w, w[4 * jc + 3], a3);
                    if ((jc & 7) == 7) __builtin_amdgcn_sched_barrier(0);   // keeps hipcc from hoisting all 32 reads (128 registers) at once
                }
                float a = (a0 + a1) + (a2 + a3);
                a += QuadDpp::mov<0xB1>(a);
                a += QuadDpp::mov<0x4E>(a);
                dh_rec[r] = a;
            }
        }
        if (probe) p.ts[(size_t)blockIdx.x * 8 + 3] = __builtin_amdgcn_s_memtime() + (unsigned long long)(dh_rec[0] == 1.2345e33f);
    };
    for (int s = smax; s >= 0; s -= 3) {
        step(s, ring[0], ring[2]);
        if (s - 1 >= 0) step(s - 1, ring[1], ring[0]);
        if (s - 2 >= 0) step(s - 2, ring[2], ring[1]);
    }
    if (p.ts && tid == 0) { p.ts[(size_t)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_memtime(); p.ts[(size_t)blockIdx.x * 8 + 6] = (unsigned long long)(smax + 1); }
}

// ---------------------------------------------------------------- forward driver
struct FwdBufs {
    float *a1, *e, *gx, *hs, *d1, *q;
    float *gates, *cs, *hprev;  // nullable; indexed by (sample - keep_from)
    int keep_from;
};

static size_t fwd_bufs_floats(int N, int T, int nkeep) {
    const size_t r = (size_t)N * T;
    size_t n = r * 128 + r * 128 + r * 512 + r * 256 + r * 128 + r;  // a1,e,gx,hs,d1,q
    n += (size_t)2 * nkeep * T * (512 + 128 + 128);
    return n + 64 * 16;  // alignment slack
}

static FwdBufs carve_fwd(Arena& ar, int N, int T, int keep_from) {
    const size_t r = (size_t)N * T;
    FwdBufs b{};
    b.a1 = ar.take<float>(r * 128);
    b.e = ar.take<float>(r * 128);
    b.gx = ar.take<float>(r * 512);
    b.hs = ar.take<float>(r * 256);
    b.d1 = ar.take<float>(r * 128);
    b.q = ar.take<float>(r);
    b.keep_from = keep_from;
    const int nkeep = (keep_from >= 0 && keep_from < N) ? N - keep_from : 0;
    if (nkeep > 0) {
        b.gates = ar.take<float>((size_t)2 * nkeep * T * 512);
        b.cs = ar.take<float>((size_t)2 * nkeep * T * 128);
        b.hprev = ar.take<float>((size_t)2 * nkeep * T * 128);
    }
    return b;
}

static int rows_per_wg(int rows) {
    // one workgroup per CU holds W_hh in registers; more rows than CUs -> several rows share a WG
    if (rows <= 256) return 1;
    if (rows <= 512) return 2;
    return 4;
}

// ---------------------------------------------------------------- fused forward (DQN_FUSED): three launches per pass,
// or three launches for a policy pass AND a target pass together (see brain_fused.h)
struct FwdPass {
    const float* prm;
    const float* x;      // samples [0, N0)
    const float* x2;     // samples [N0, N) (nullptr: all from x)
    int N, N0;
    const FwdBufs* b;
};
static unsigned long long* g_lstm_probe_bwd = nullptr;
static unsigned long long* g_lstm_probe = nullptr;     // set by ivosw_lstm_probe (tuning aid; never in a product call path)
static LstmFwd lstm_fwd_args(const FwdPass& f, int T) {
    LstmFwd lf{};
    lf.ts = g_lstm_probe;
    lf.whh = f.prm + O_WHH; lf.gx = f.b->gx; lf.hs = f.b->hs; lf.N = f.N; lf.T = T;
    lf.keep_from = f.b->gates ? f.b->keep_from : f.N;
    lf.gates = f.b->gates; lf.cs = f.b->cs; lf.hprev = f.b->hprev;
    return lf;
}
static bool fused_forward_ok(const FwdPass* f, int n) {
    if (!tune_get("DQN_FUSED", 1) || !tune_get("LSTM_QUAD", 1)) return false;
    if (n == 2) return tune_get("LSTM_MFMA", 1) && 2 * f[0].N >= 8 && 2 * f[1].N >= 8;   // the pair launch is the MFMA recurrence
    return true;
}
static void brain_forward_fused(const FwdPass* f, int n, int T, hipStream_t st, const EncDraw* draw = nullptr) {
    EncGroup eg{};
    if (draw) eg.dr = *draw;
    DecGroup dg{};
    int tiles[2] = {0, 0};
    for (int i = 0; i < n; ++i) {
        const int rows = f[i].N * T, keep_row = f[i].b->gates ? f[i].b->keep_from * T : rows;
        tiles[i] = (rows + FM - 1) / FM;
        eg.j[i] = EncJob{f[i].prm, f[i].x, f[i].x2 ? f[i].x2 : f[i].x, f[i].b->gx, f[i].b->a1, f[i].b->e,
                         f[i].x2 ? f[i].N0 * T : rows, rows, keep_row};
        dg.j[i] = DecJob{f[i].prm, f[i].b->hs, f[i].b->d1, f[i].b->q, rows, keep_row};
    }
    eg.first1 = dg.first1 = tiles[0];
    hipLaunchKernelGGL(enc_fused_kernel, dim3(tiles[0] + tiles[1]), dim3(256), 0, st, eg, O_W1, O_B1, O_W2, O_B2, O_WIH);
    // recurrence + decoder in one kernel (FWD_MEGA, default OFF: measured 201 us per DQN step against 185.5 us for the separate
    // decoder launch - the decoder's 16 us run on 192 workgroups after their recurrences instead of on the whole chip) when every
    // pass is batched enough for the MFMA recurrence and the pair's 2 T rows of [h_fw | h_bw] fit LDS
    bool mega = tune_get("FWD_MEGA", 0) && tune_get("LSTM_MFMA", 1) && mega_lds_bytes(T) <= 150 * 1024;
    for (int i = 0; i < n; ++i) mega = mega && 2 * f[i].N >= 8;
    if (mega) {
        static std::once_flag once;
        std::call_once(once, [] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(brain_fwd_mega_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); (void)hipGetLastError(); });
        FwdMegaPair mp{};
        int wgs[2] = {0, 0};
        for (int i = 0; i < n; ++i) {
            const FwdBufs& b = *f[i].b;
            const int keep_from = b.gates ? b.keep_from : f[i].N;
            mp.j[i] = FwdMega{f[i].prm, b.gx, b.hs, b.gates, b.cs, b.hprev, b.d1, b.q, f[i].N, T, keep_from, keep_from, tune_get("MEGA_DBG", 0)};
            wgs[i] = (f[i].N + 1) / 2;
        }
        mp.first1 = wgs[0];
        hipLaunchKernelGGL(brain_fwd_mega_kernel, dim3(wgs[0] + wgs[1]), dim3(512), mega_lds_bytes(T), st, mp);
        return;
    }
    if (n == 2) {
        LstmFwdPair pr{};
        pr.j[0] = lstm_fwd_args(f[0], T);
        pr.j[1] = lstm_fwd_args(f[1], T);
        pr.first1 = (2 * f[0].N + 3) / 4;
        hipLaunchKernelGGL(lstm_fwd_mfma_pair_kernel, dim3(pr.first1 + (2 * f[1].N + 3) / 4), dim3(512), 0, st, pr);
    } else {
        const LstmFwd lf = lstm_fwd_args(f[0], T);
        const int N = f[0].N, R = rows_per_wg(2 * N), nwg = (2 * N + R - 1) / R;
        if (tune_get("LSTM_MFMA", 1) && 2 * N >= 8) hipLaunchKernelGGL(lstm_fwd_mfma_kernel, dim3((2 * N + 3) / 4), dim3(512), 0, st, lf);
        else if (R == 1) hipLaunchKernelGGL(lstm_fwd_quad_kernel<1>, dim3(nwg), dim3(512), 0, st, lf);
        else if (R == 2) hipLaunchKernelGGL(lstm_fwd_quad_kernel<2>, dim3(nwg), dim3(512), 0, st, lf);
        else hipLaunchKernelGGL(lstm_fwd_quad_kernel<4>, dim3(nwg), dim3(512), 0, st, lf);
    }
    hipLaunchKernelGGL(dec_fused_kernel, dim3(tiles[0] + tiles[1]), dim3(256), 0, st, dg, O_W3, O_B3, O_W4, O_B4);
}

// x2 != nullptr: samples [0, N0) come from x, [N0, N) from x2
static void brain_forward_internal(const float* prm, const float* x, int N, int T, const FwdBufs& b, hipStream_t st,
                                   const float* x2 = nullptr, int N0 = 0) {
    const int rows = N * T;
    hipLaunchKernelGGL(enc1_kernel, dim3((rows * HD + 255) / 256), dim3(256), 0, st, prm, x, x2 ? x2 : x, x2 ? N0 * T : rows, rows, b.a1);
    GemmF32 g{};
    // e = a1 * W2^T + b2
    g.A = b.a1; g.sam = 128; g.sak = 1; g.B = prm + O_W2; g.sbk = 1; g.sbn = 128; g.C = b.e; g.ldc = 128;
    g.M = rows; g.N = 128; g.K = 128; g.bias = prm + O_B2; g.splitk = 1;
    launch_gemm_f32(g, st);
    // gx = e * Wih^T
    g = GemmF32{};
    g.A = b.e; g.sam = 128; g.sak = 1; g.B = prm + O_WIH; g.sbk = 1; g.sbn = 128; g.C = b.gx; g.ldc = 512;
    g.M = rows; g.N = 512; g.K = 128; g.splitk = 1;
    launch_gemm_f32(g, st);
    LstmFwd lf{};
    lf.whh = prm + O_WHH; lf.gx = b.gx; lf.hs = b.hs; lf.N = N; lf.T = T;
    lf.keep_from = b.gates ? b.keep_from : N;
    lf.gates = b.gates; lf.cs = b.cs; lf.hprev = b.hprev;
    const int R = rows_per_wg(2 * N);
    const int nwg = (2 * N + R - 1) / R;
    if (tune_get("LSTM_QUAD", 1) && tune_get("LSTM_MFMA", 1) && 2 * N >= 8) {
        hipLaunchKernelGGL(lstm_fwd_mfma_kernel, dim3((2 * N + 3) / 4), dim3(512), 0, st, lf);
    } else if (tune_get("LSTM_QUAD", 1)) {
        if (R == 1) hipLaunchKernelGGL(lstm_fwd_quad_kernel<1>, dim3(nwg), dim3(512), 0, st, lf);
        else if (R == 2) hipLaunchKernelGGL(lstm_fwd_quad_kernel<2>, dim3(nwg), dim3(512), 0, st, lf);
        else hipLaunchKernelGGL(lstm_fwd_quad_kernel<4>, dim3(nwg), dim3(512), 0, st, lf);
    } else if (R == 1) hipLaunchKernelGGL(lstm_fwd_kernel<1>, dim3(nwg), dim3(512), 0, st, lf);
    else if (R == 2) hipLaunchKernelGGL(lstm_fwd_kernel<2>, dim3(nwg), dim3(512), 0, st, lf);
    else hipLaunchKernelGGL(lstm_fwd_kernel<4>, dim3(nwg), dim3(512), 0, st, lf);
    // d1 = relu(relu(hcat) * W3^T + b3)
    g = GemmF32{};
    g.A = b.hs; g.sam = 256; g.sak = 1; g.relu_a = 1; g.B = prm + O_W3; g.sbk = 1; g.sbn = 256; g.C = b.d1; g.ldc = 128;
    g.M = rows; g.N = 128; g.K = 256; g.bias = prm + O_B3; g.relu = 1; g.splitk = 1;
    launch_gemm_f32(g, st);
    hipLaunchKernelGGL(dec2_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, prm, b.d1, rows, b.q);
}

// ---------------------------------------------------------------- DQN head: targets, loss, dQ
// one block; q_next_pol/q_next_tgt/q_state are [B,T]
__global__ __launch_bounds__(256) void dqn_head_kernel(const float* __restrict__ q_np, const float* __restrict__ q_nt,
                                                       const float* __restrict__ q_s, const int64_t* __restrict__ action,
                                                       const float* __restrict__ r_step, const float* __restrict__ r_done,
                                                       int B, int T, float gamma, float* __restrict__ dq,
                                                       float* __restrict__ loss, float* __restrict__ db4) {
    __shared__ float red[2][256];
    float l = 0.f, sdq = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        const float* qp = q_np + (size_t)b * T;
        int am = 0;
        float best = qp[0];
        for (int t = 1; t < T; ++t) {
            const float v = qp[t];
            if (v > best) { best = v; am = t; }  // first maximum (torch CPU max(1)[1])
        }
        const float qn = q_nt[(size_t)b * T + am];
        const float y1 = qn * gamma + r_step[b] * 0.1f;
        const float y2 = r_done[b] * 0.1f;
        int a = (int)action[b];
        a = min(max(a, 0), T - 1);
        const float qsa = q_s[(size_t)b * T + a];
        const float e1 = qsa - y1, e2 = qsa - y2;
        l += e1 * e1 + e2 * e2;
        const float d = (2.0f / (float)B) * (e1 + e2);
        dq[b] = d;
        sdq += d;
    }
    red[0][threadIdx.x] = l;
    red[1][threadIdx.x] = sdq;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            red[0][threadIdx.x] += red[0][threadIdx.x + o];
            red[1][threadIdx.x] += red[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *loss = red[0][0] / (float)B;
        *db4 = red[1][0];
    }
}

// per sample b (grid B, 128 threads): the decoder backward touches only row (b, action[b])
__global__ __launch_bounds__(128) void dec_bwd_rows_kernel(const float* __restrict__ prm, const float* __restrict__ dq,
                                                           const int64_t* __restrict__ action, const float* __restrict__ d1,
                                                           const float* __restrict__ hs, int T, float* __restrict__ dd1c,
                                                           float* __restrict__ w4term, float* __restrict__ hcc) {
    const int b = blockIdx.x, j = threadIdx.x;
    int a = (int)action[b];
    a = min(max(a, 0), T - 1);
    const size_t row = (size_t)b * T + a;
    const float d = dq[b], v = d1[row * HD + j];
    w4term[(size_t)b * HD + j] = d * v;
    dd1c[(size_t)b * HD + j] = (v > 0.f) ? d * prm[O_W4 + j] : 0.f;
    hcc[(size_t)b * 256 + j] = fmaxf(hs[row * 256 + j], 0.f);
    hcc[(size_t)b * 256 + HD + j] = fmaxf(hs[row * 256 + HD + j], 0.f);
}

struct DqnWs {
    FwdBufs pol, tgt;
    float *xcat, *dq, *dd1c, *w4term, *hcc, *dhc, *dG, *dgx, *de, *da1, *slabs;
};

int tune_get(const char* key, int dflt);   // capi.cpp

static constexpr int WG_SPLIT = 16;
static constexpr int WG_SPLIT_MAX = 40;   // slab sets of the two long-K weight gradients are sized for this (tunables DQN_SPLIT_HH / _IH)

// One helper stream + four events per device: the only state the library keeps between calls (documented in
// INTEGRATION.md).  Creation is serialised and all-or-nothing; ivosw_dqn_loss_grad holds the device's mutex while it
// enqueues, so two host threads (or two caller streams) on one device cannot interleave their fork / join pairs — an
// event re-recorded by a later call does not disturb waits that were already enqueued on it.
struct Side {
    std::mutex mu;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {};
    bool tried = false;
};
static Side* side_for_current_device() {
    static Side sides[64];
    static std::mutex init_mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    Side& s = sides[dev];
    std::lock_guard<std::mutex> lk(init_mu);
    if (!s.tried) {
        s.tried = true;
        bool ok = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) == hipSuccess;
        int made = 0;
        for (; ok && made < 4; ++made) ok = hipEventCreateWithFlags(&s.ev[made], hipEventDisableTiming) == hipSuccess;
        if (!ok) {      // partial failure: give everything back, the single-stream path is used from now on
            for (int i = 0; i < made; ++i) if (s.ev[i]) (void)hipEventDestroy(s.ev[i]);
            if (s.stream) (void)hipStreamDestroy(s.stream);
            s.stream = nullptr;
            (void)hipGetLastError();
        }
    }
    return s.stream ? &s : nullptr;
}

static size_t dqn_ws_floats(int B, int T) {
    const size_t r = (size_t)B * T;
    size_t n = fwd_bufs_floats(2 * B, T, B) + fwd_bufs_floats(B, T, 0);
    n += 2 * r * 2;                        // xcat
    n += B + (size_t)B * 128 * 2 + (size_t)B * 256 * 2;  // dq, dd1c, w4term, hcc, dhc
    n += 2 * r * 512 + r * 512 + r * 128 + r * 128;       // dG, dgx, de, da1
    n += (size_t)(WG_SPLIT + 2 * WG_SPLIT_MAX) * 512 * 128; // split-K slabs: one set per concurrently reduced weight gradient
    return n + 8 * 512 + 64 * 32;
}

}  // namespace ivosw

using namespace ivosw;

extern "C" size_t ivosw_brain_ws_bytes(int N, int T) {
    if (N <= 0 || T <= 0) return 0;
    return fwd_bufs_floats(N, T, 0) * sizeof(float);
}

extern "C" int ivosw_brain_forward(const float* params, const float* x, int N, int T, float* q, void* ws,
                                   size_t ws_bytes, ivosw_stream_t stream) {
    IVOSW_REQUIRE(params && x && q && ws, "null pointer");
    IVOSW_ON_DEVICE_OF(q);
    IVOSW_REQUIRE(N > 0 && T > 0, "N and T must be positive");
    if (ws_bytes < ivosw_brain_ws_bytes(N, T)) {
        set_error("ivosw_brain_forward: workspace %zu < %zu", ws_bytes, ivosw_brain_ws_bytes(N, T));
        return IVOSW_ERR_WS;
    }
    hipStream_t st = as_stream(stream);
    Arena ar(ws);
    FwdBufs b = carve_fwd(ar, N, T, -1);
    b.q = q;                              // the decoder writes the caller's buffer (a device-to-device copy was a launch of its own)
    const FwdPass fp{params, x, nullptr, N, 0, &b};
    if (fused_forward_ok(&fp, 1)) brain_forward_fused(&fp, 1, T, st);
    else brain_forward_internal(params, x, N, T, b, st);
    IVOSW_CHECK_LAUNCH();
    return IVOSW_OK;
}

// first maximum of each row (numpy / torch-CPU argmax): one wave per row, lanes stride the frames, ties go to the lower index
__global__ __launch_bounds__(64) void argmax_rows_kernel(const float* __restrict__ q, int N, int T, int64_t* __restrict__ idx) {
    const int n = blockIdx.x, lane = threadIdx.x;
    const float* r = q + (size_t)n * T;
    float best = -INFINITY;
    int am = 0x7fffffff;
    for (int t = lane; t < T; t += 64) {
        const float v = r[t];
        if (v > best || am == 0x7fffffff) { best = v; am = t; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oa = __shfl_xor(am, o, 64);
        if (oa != 0x7fffffff && (am == 0x7fffffff || ob > best || (ob == best && oa < am))) { best = ob; am = oa; }
    }
    if (lane == 0) idx[n] = am;
}

extern "C" int ivosw_brain_argmax(const float* q, int N, int T, int64_t* idx, ivosw_stream_t stream) {
    IVOSW_REQUIRE(q && idx, "null pointer");
    IVOSW_ON_DEVICE_OF(idx);
    IVOSW_REQUIRE(N > 0 && T > 0, "N and T must be positive");
    hipLaunchKernelGGL(argmax_rows_kernel, dim3(N), dim3(64), 0, as_stream(stream), q, N, T, idx);
    IVOSW_CHECK_LAUNCH();
    return IVOSW_OK;
}

extern "C" size_t ivosw_dqn_ws_bytes(int B, int T) {
    if (B <= 0 || T <= 0) return 0;
    return dqn_ws_floats(B, T) * sizeof(float);
}

// draw != nullptr: the minibatch is drawn and gathered by the encoder launch (state / new_state / action / rewards are then OUTPUTS
// of the step, written before anything reads them).  defer != nullptr: the fixed-order slab reduction is NOT launched; *defer
// describes it and the caller folds it into clamp + Adam (ivosw_dqn_step_drawn).  Either needs the fused launch chain
// (DQN_FUSED / DQN_GROUP / DQN_TAIL at their defaults): *folded says whether it was taken.
static int dqn_loss_grad_impl(const float* policy, const float* target, const float* state, const float* new_state, const int64_t* action,
                              const float* reward_step, const float* reward_done, int B, int T, float gamma, float* grads, float* loss,
                              void* ws, size_t ws_bytes, ivosw_stream_t stream, const EncDraw* draw, ReduceGroup* defer, bool* folded) {
    IVOSW_REQUIRE(policy && target && state && new_state && action && reward_step && reward_done && grads && loss && ws,
                  "null pointer");
    IVOSW_REQUIRE(B > 0 && T > 0, "B and T must be positive");
    if (ws_bytes < ivosw_dqn_ws_bytes(B, T)) {
        set_error("ivosw_dqn_loss_grad: workspace %zu < %zu", ws_bytes, ivosw_dqn_ws_bytes(B, T));
        return IVOSW_ERR_WS;
    }
    if (folded) *folded = false;
    hipStream_t st = as_stream(stream);
    const int rows = B * T;
    Arena ar(ws);
    DqnWs w{};
    w.pol = carve_fwd(ar, 2 * B, T, B);   // samples [0,B) = s', [B,2B) = s (kept for backward)
    w.tgt = carve_fwd(ar, B, T, -1);
    w.xcat = ar.take<float>((size_t)2 * rows * 2);
    w.dq = ar.take<float>(B);
    w.dd1c = ar.take<float>((size_t)B * 128);
    w.w4term = ar.take<float>((size_t)B * 128);
    w.hcc = ar.take<float>((size_t)B * 256);
    w.dhc = ar.take<float>((size_t)B * 256);
    w.dG = ar.take<float>((size_t)2 * rows * 512);
    w.dgx = ar.take<float>((size_t)rows * 512);
    w.de = ar.take<float>((size_t)rows * 128);
    w.da1 = ar.take<float>((size_t)rows * 128);
    w.slabs = ar.take<float>((size_t)WG_SPLIT * 512 * 128);
    float* slabs2 = ar.take<float>((size_t)WG_SPLIT_MAX * 512 * 128);
    float* slabs3 = ar.take<float>((size_t)WG_SPLIT_MAX * 512 * 128);
    float* slabs4 = ar.take<float>(8 * 512);                  // row-split column sums

    // The step is a chain of ~40 launches that each occupy a fraction of the chip for 5-30 us: independent branches run
    // on a second stream (fork / join with events), so their kernels overlap instead of queueing behind each other.
    Side* sd = tune_get("DQN_STREAMS", 1) ? side_for_current_device() : nullptr;
    std::unique_lock<std::mutex> side_lock;
    if (sd) side_lock = std::unique_lock<std::mutex>(sd->mu);
    hipStream_t s2 = sd ? sd->stream : st;
    bool sync_ok = true;
    auto fork = [&](int k) {          // s2 continues after everything enqueued on st so far
        if (!sd) return;
        sync_ok &= hipEventRecord(sd->ev[k], st) == hipSuccess;
        sync_ok &= hipStreamWaitEvent(s2, sd->ev[k], 0) == hipSuccess;
    };
    auto join = [&](int k) {          // st continues after everything enqueued on s2 so far
        if (!sd) return;
        sync_ok &= hipEventRecord(sd->ev[k], s2) == hipSuccess;
        sync_ok &= hipStreamWaitEvent(st, sd->ev[k], 0) == hipSuccess;
    };

    // ---- forward: policy on [s'; s] in one batch, target on s' (agent.py:135-137,144); the target pass runs beside it
    // (host order matters: the policy chain is the critical path, so it is enqueued first)
    const FwdPass passes[2] = {{policy, new_state, state, 2 * B, B, &w.pol}, {target, new_state, nullptr, B, 0, &w.tgt}};
    const bool fused = fused_forward_ok(passes, 2);
    const bool fold = (draw || defer) && fused && tune_get("DQN_GROUP", 1) && tune_get("DQN_TAIL", 1);
    if ((draw || defer) && !fold) return IVOSW_OK;    // the caller runs the un-folded sequence instead (nothing was launched)
    if (folded) *folded = fold;
    const float* q_np = w.pol.q;
    const float* q_s = w.pol.q + rows;
    const float* d1_s = w.pol.d1 + (size_t)rows * 128;
    const float* hs_s = w.pol.hs + (size_t)rows * 256;
    GemmF32 g{};
    if (fused) {
        // both nets per launch, then head + decoder backward + dL/dh in one: 4 launches, one stream, no events
        brain_forward_fused(passes, 2, T, st, draw);
        hipLaunchKernelGGL(head_fused_kernel, dim3(B), dim3(256), 0, st, policy, O_W3, O_W4, q_np, w.tgt.q, q_s, action, reward_step,
                           reward_done, B, T, gamma, d1_s, hs_s, w.dq, w.dd1c, w.w4term, w.hcc, w.dhc, loss, grads + O_B4);
    } else {
        fork(0);
        brain_forward_internal(policy, new_state, 2 * B, T, w.pol, st, state, B);
        brain_forward_internal(target, new_state, B, T, w.tgt, s2);
        join(1);

        // ---- head: Double-DQN targets, loss, dL/dQsa (agent.py:136-151)
        hipLaunchKernelGGL(dqn_head_kernel, dim3(1), dim3(256), 0, st, q_np, w.tgt.q, q_s, action, reward_step, reward_done,
                           B, T, gamma, w.dq, loss, grads + O_B4);

        // ---- decoder backward on the B rows that carry loss
        hipLaunchKernelGGL(dec_bwd_rows_kernel, dim3(B), dim3(128), 0, st, policy, w.dq, action, d1_s, hs_s, T, w.dd1c,
                           w.w4term, w.hcc);
        // dhc[B,256] = (dd1c * W3) . (hcat > 0)
        g.A = w.dd1c; g.sam = 128; g.sak = 1; g.B = policy + O_W3; g.sbk = 256; g.sbn = 1; g.C = w.dhc; g.ldc = 256;
        g.M = B; g.N = 256; g.K = 128; g.mask = w.hcc; g.splitk = 1;
        launch_gemm_f32(g, st);
    }

    // ---- BPTT through the shared cell
    LstmBwd lb{};
    lb.whh = policy + O_WHH; lb.gates = w.pol.gates; lb.cs = w.pol.cs; lb.dhc = w.dhc; lb.action = action; lb.dG = w.dG;
    lb.N = B; lb.T = T; lb.ts = g_lstm_probe_bwd;
    {
        const int R = rows_per_wg(2 * B);
        const int nwg = (2 * B + R - 1) / R;
        if (tune_get("LSTM_QUAD", 1)) {
            // one row per workgroup whatever the batch: with two or more rows the per-row state no longer fits beside W_hh
            // (hipcc spills 320+ registers), and a second round of workgroups costs what the second row would
            hipLaunchKernelGGL(lstm_bwd_quad_kernel<1>, dim3(2 * B), dim3(512), 0, st, lb);
        } else if (R == 1) hipLaunchKernelGGL(lstm_bwd_kernel<1>, dim3(nwg), dim3(512), 0, st, lb);
        else if (R == 2) hipLaunchKernelGGL(lstm_bwd_kernel<2>, dim3(nwg), dim3(512), 0, st, lb);
        else hipLaunchKernelGGL(lstm_bwd_kernel<4>, dim3(nwg), dim3(512), 0, st, lb);
    }
    const float* e_s = w.pol.e + (size_t)rows * 128;
    const float* a1_s = w.pol.a1 + (size_t)rows * 128;
    const float* dG_bw = w.dG + (size_t)rows * 512;            // dgx = dG[fw] + dG[bw] (the same e_t feeds both directions): summed
                                                                // on load by the two GEMMs that consume it (GemmF32::A2)
    if (!tune_get("DQN_GROUP", 1)) {
        const size_t n = (size_t)rows * 512;
        hipLaunchKernelGGL(add2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w.dG, dG_bw, w.dgx, n);
    }
    const bool grp = tune_get("DQN_GROUP", 1) != 0;
    const float* dgx_a = grp ? w.dG : w.dgx;
    const float* dgx_a2 = grp ? dG_bw : nullptr;

    GemmF32 gw[4];
    // dW3[128,256] = dd1c^T * hcc
    gw[0] = GemmF32{};
    gw[0].A = w.dd1c; gw[0].sam = 1; gw[0].sak = 128; gw[0].B = w.hcc; gw[0].sbk = 256; gw[0].sbn = 1; gw[0].C = grads + O_W3; gw[0].ldc = 256;
    gw[0].M = 128; gw[0].N = 256; gw[0].K = B; gw[0].splitk = 1;
    // dWhh[512,128] = sum_{d,n,t} dG^T * hprev          (K = 2*B*T), split-K slabs
    gw[1] = GemmF32{};
    gw[1].A = w.dG; gw[1].sam = 1; gw[1].sak = 512; gw[1].B = w.pol.hprev; gw[1].sbk = 128; gw[1].sbn = 1; gw[1].ldc = 128;
    const int split_hh = std::min(WG_SPLIT_MAX, std::max(1, tune_get("DQN_SPLIT_HH", WG_SPLIT))), split_ih = std::min(WG_SPLIT_MAX, std::max(1, tune_get("DQN_SPLIT_IH", WG_SPLIT)));
    gw[1].M = 512; gw[1].N = 128; gw[1].K = 2 * rows; gw[1].C = slabs2; gw[1].splitk = split_hh;
    // dWih[512,128] = dgx^T * e
    gw[2] = GemmF32{};
    gw[2].A = dgx_a; gw[2].A2 = dgx_a2; gw[2].sam = 1; gw[2].sak = 512; gw[2].B = e_s; gw[2].sbk = 128; gw[2].sbn = 1; gw[2].ldc = 128;
    gw[2].M = 512; gw[2].N = 128; gw[2].K = rows; gw[2].C = slabs3; gw[2].splitk = split_ih;
    // de[rows,128] = dgx * Wih
    GemmF32 gde = GemmF32{};
    gde.A = dgx_a; gde.A2 = dgx_a2; gde.sam = 512; gde.sak = 1; gde.B = policy + O_WIH; gde.sbk = 128; gde.sbn = 1; gde.C = w.de; gde.ldc = 128;
    gde.M = rows; gde.N = 128; gde.K = 512; gde.splitk = 1;
    // da1[rows,128] = (de * W2) . (a1 > 0)
    GemmF32 ga = GemmF32{};
    ga.A = w.de; ga.sam = 128; ga.sak = 1; ga.B = policy + O_W2; ga.sbk = 128; ga.sbn = 1; ga.C = w.da1; ga.ldc = 128;
    ga.M = rows; ga.N = 128; ga.K = 128; ga.mask = a1_s; ga.splitk = 1;
    // dW2[128,128] = de^T * a1 ; db2 = colsum(de)
    GemmF32 gb = GemmF32{};
    gb.A = w.de; gb.sam = 1; gb.sak = 128; gb.B = a1_s; gb.sbk = 128; gb.sbn = 1; gb.ldc = 128;
    gb.M = 128; gb.N = 128; gb.K = rows;
    if (grp) {
        // Everything after the recurrence on ONE stream, as four launches (a fork / join pair costs ~20 us of stream time and
        // the side stream's GEMMs took the CUs of the chain that everything else waits for):
        //   1. {de (first in the grid: the chain below depends on it), dW3, dWhh slabs, dWih slabs}
        //   2. {da1, dW2 slabs}: both read de and nothing of each other
        //   3. column sums {dW4 <- w4term, db3 <- dd1c} and, split 8 ways over the rows, {db2 <- de} {db1, dW1 <- da1 weighted by x}
        //   4. one fixed-order reduction of every slab set
        int split_hh_used = split_hh, split_ih_used = split_ih, split_w2_used = WG_SPLIT;
        bool tail_colsums_done = false;
        if (tune_get("DQN_TAIL", 1)) {
            // register-resident kernels (brain_bwd.h): {dgrad chain de -> da1 per 16-row tile, dW_hh, dW_ih, dW3} in one launch,
            // then dW2 (it needs de)
            auto tiles = [](const WgradJob& j) { return (j.M / 64) * (j.N / 128) * j.nslab; };
            TailGroup t1{};
            t1.dg = DgradArgs{w.dG, dG_bw, policy + O_WIH, policy + O_W2, a1_s, w.de, w.da1, rows};
            t1.n_dgrad = ((rows + 15) / 16 + 7) / 8 * 8;        // padded to a multiple of 8 (XCD-aligned weight-gradient section)
            // K-slabs of a multiple of 48 rows where the sizes allow (the kernel's loop is unrolled by three 16-row steps)
            auto slabs_for = [](int K, int want, int cap) {
                const int per = std::max(48, (K / want + 47) / 48 * 48);
                return std::min(cap, std::max(1, (K + per - 1) / per));
            };
            split_hh_used = slabs_for(2 * rows, 23, WG_SPLIT_MAX); split_ih_used = slabs_for(rows, 12, WG_SPLIT_MAX);   // B = 128, T = 25: 200 + 184 + 96 + 4 workgroups, one round at two per CU
            t1.w[0] = WgradJob{w.dG, nullptr, w.pol.hprev, slabs2, 512, 128, 512, 128, 2 * rows, split_hh_used};
            t1.w[1] = WgradJob{w.dG, dG_bw, e_s, slabs3, 512, 128, 512, 128, rows, split_ih_used};
            t1.w[2] = WgradJob{w.dd1c, nullptr, w.hcc, grads + O_W3, 128, 256, 128, 256, B, 1};
            t1.n = 3;
            for (int i = 0; i < 3; ++i) t1.first[i + 1] = t1.first[i] + tiles(t1.w[i]);
            for (int i = 4; i <= TAIL_MAX; ++i) t1.first[i] = t1.first[3];
            hipLaunchKernelGGL(bwd_tail_kernel, dim3(t1.n_dgrad + t1.first[3]), dim3(256), 0, st, t1);
            TailGroup t2{};
            split_w2_used = slabs_for(rows, 32, 64);            // 48-row slabs: this launch is latency-bound, so many small workgroups
            t2.w[0] = WgradJob{w.de, nullptr, a1_s, w.slabs, 128, 128, 128, 128, rows, split_w2_used};
            t2.n = 1;
            t2.first[1] = tiles(t2.w[0]);
            for (int i = 2; i <= TAIL_MAX; ++i) t2.first[i] = t2.first[1];
            // ... and the four column sums beside it (they used to be a launch of their own)
            t2.cs[0] = CsumJob{w.de, nullptr, slabs4, nullptr, rows, 128, 128, 8};                                   // db2 slabs
            t2.cs[1] = CsumJob{w.da1, state, slabs4 + 8 * 128, slabs4 + 16 * 128, rows, 128, 128, 8};                // db1, dW1 slabs
            t2.cs[2] = CsumJob{w.w4term, nullptr, grads + O_W4, nullptr, B, 128, 128, 1};
            t2.cs[3] = CsumJob{w.dd1c, nullptr, grads + O_B3, nullptr, B, 128, 128, 1};
            t2.n_cs = 4;
            for (int i = 0; i < 4; ++i) t2.cs_first[i + 1] = t2.cs_first[i] + ((t2.cs[i].N + 31) / 32) * t2.cs[i].nsplit;
            hipLaunchKernelGGL(bwd_tail_kernel, dim3(t2.first[1] + t2.cs_first[4]), dim3(256), 0, st, t2);
            tail_colsums_done = true;
        } else {
            GemmF32 g1[4] = {gde, gw[0], gw[1], gw[2]};
            launch_gemm_f32_group(g1, 4, st);
            gb.C = w.slabs; gb.splitk = WG_SPLIT;
            GemmF32 g2[2] = {ga, gb};
            launch_gemm_f32_group(g2, 2, st);
        }
        constexpr int CS = 8;                           // row splits of the two long column sums
        float* cs_b2 = slabs4;                          // [CS][128]
        float* cs_b1 = slabs4 + CS * 128;               // [CS][128]
        float* cs_w1 = slabs4 + 2 * CS * 128;           // [CS][256]
        if (!tail_colsums_done) {
        ColsumGroup c1{};                               // all four column sums in one launch (the two short ones unsplit)
        c1.j[0] = ColsumJob{w.de, nullptr, cs_b2, nullptr, rows, 128, 128, CS};
        c1.j[1] = ColsumJob{w.da1, state, cs_b1, cs_w1, rows, 128, 128, CS};
        c1.j[2] = ColsumJob{w.w4term, nullptr, grads + O_W4, nullptr, B, 128, 128, 1};
        c1.j[3] = ColsumJob{w.dd1c, nullptr, grads + O_B3, nullptr, B, 128, 128, 1};
        hipLaunchKernelGGL(colsum_group_kernel, dim3(4, 4, CS), dim3(1024), 0, st, c1);
        }
        ReduceGroup rg{};
        rg.slabs[0] = slabs2; rg.out[0] = grads + O_WHH; rg.n[0] = 512 * 128; rg.nslab[0] = split_hh_used;
        rg.slabs[1] = slabs3; rg.out[1] = grads + O_WIH; rg.n[1] = 512 * 128; rg.nslab[1] = split_ih_used;
        rg.slabs[2] = w.slabs; rg.out[2] = grads + O_W2; rg.n[2] = 128 * 128; rg.nslab[2] = split_w2_used;
        rg.slabs[3] = cs_b2; rg.out[3] = grads + O_B2; rg.n[3] = 128; rg.nslab[3] = CS;
        rg.slabs[4] = cs_b1; rg.out[4] = grads + O_B1; rg.n[4] = 128; rg.nslab[4] = CS;
        rg.slabs[5] = cs_w1; rg.out[5] = grads + O_W1; rg.n[5] = 256; rg.nslab[5] = CS;
        if (defer) *defer = rg;
        else hipLaunchKernelGGL(splitk_reduce_group_kernel, dim3(512 * 128 / 256, 6), dim3(256), 0, st, rg);
    } else {
        fork(2);
        hipLaunchKernelGGL(colsum_kernel, dim3(4), dim3(1024), 0, s2, w.w4term, B, 128, 128, grads + O_W4);
        hipLaunchKernelGGL(colsum_kernel, dim3(4), dim3(1024), 0, s2, w.dd1c, B, 128, 128, grads + O_B3);
        launch_gemm_f32(gw[0], s2);
        launch_gemm_f32_splitk(gw[1], grads + O_WHH, slabs2, split_hh, s2);
        launch_gemm_f32_splitk(gw[2], grads + O_WIH, slabs2, split_ih, s2);
        launch_gemm_f32(gde, st);
        launch_gemm_f32_splitk(gb, grads + O_W2, w.slabs, WG_SPLIT, st);
        hipLaunchKernelGGL(colsum_kernel, dim3(4), dim3(1024), 0, st, w.de, rows, 128, 128, grads + O_B2);
        launch_gemm_f32(ga, st);
        // dW1[128,2] = da1^T * x ; db1 = colsum(da1)
        g = GemmF32{};
        g.A = w.da1; g.sam = 1; g.sak = 128; g.B = state; g.sbk = 2; g.sbn = 1; g.ldc = 2;
        g.M = 128; g.N = 2; g.K = rows;
        launch_gemm_f32_splitk(g, grads + O_W1, w.slabs, WG_SPLIT, st);
        hipLaunchKernelGGL(colsum_kernel, dim3(4), dim3(1024), 0, st, w.da1, rows, 128, 128, grads + O_B1);
        join(3);
    }
    if (!sync_ok) {
        (void)hipGetLastError();
        set_error("ivosw_dqn_loss_grad: a fork/join event between the two streams failed");
        return IVOSW_ERR_LAUNCH;
    }
    IVOSW_CHECK_LAUNCH();
    return IVOSW_OK;
}

extern "C" int ivosw_dqn_loss_grad(const float* policy, const float* target, const float* state,
                                   const float* new_state, const int64_t* action, const float* reward_step,
                                   const float* reward_done, int B, int T, float gamma, float* grads, float* loss,
                                   void* ws, size_t ws_bytes, ivosw_stream_t stream) {
    IVOSW_REQUIRE(grads, "null pointer");
    IVOSW_ON_DEVICE_OF(grads);
    return dqn_loss_grad_impl(policy, target, state, new_state, action, reward_step, reward_done, B, T, gamma, grads, loss, ws, ws_bytes,
                              stream, nullptr, nullptr, nullptr);
}

// Clamp + Adam (clamp_adam_dev_kernel's expressions: same bits) with the step's split-K slab reduction folded in: an element of
// a slabbed tensor is summed from its slabs ON LOAD in splitk_reduce_group_kernel's order (four interleaved partial sums over
// z, eight loads in flight, (s0 + s1) + (s2 + s3)) and written to the gradient arena on the way, so the arena holds what the
// separate reduction would have left there.  off[k] = element offset of slab set k in the arena.
struct ReduceOffsets { int off[REDUCE_MAX]; };
// One element per lane, 177 workgroups: the slabs (13.5 MB at B = 128, T = 25, fresh in L2) are pulled by the whole chip — with the
// 45 workgroups of the 16-byte form the launch took 11 us, more than the reduction + update launches it replaces (5.3 + 5.1).
__global__ __launch_bounds__(1024) void clamp_adam_dev_reduce_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                                     float* __restrict__ v, int n, AdamDevState* __restrict__ st, ReduceGroup rg,
                                                                     ReduceOffsets ro, float lr, float beta1, float beta2, float eps, float wd,
                                                                     float clampv, float gscale) {
    const int step = st->step + 1;
    const double b1t = ipow((double)beta1, step), b2t = ipow((double)beta2, step);
    const float step_size = (float)((double)lr / (1.0 - b1t)), bc2_sqrt = (float)sqrt(1.0 - b2t);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        int w = -1;
#pragma unroll
        for (int k = 0; k < REDUCE_MAX; ++k)
            if (rg.nslab[k] > 0 && i >= ro.off[k] && i < ro.off[k] + rg.n[k]) w = k;
        float gi;
        if (w >= 0) {
            const float* sl = rg.slabs[w] + (i - ro.off[w]);
            const size_t nn = rg.n[w];
            const int ns = rg.nslab[w];
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            int z = 0;
            for (; z + 8 <= ns; z += 8) {
                float q[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) q[u] = sl[(size_t)(z + u) * nn];
                s0 += q[0]; s1 += q[1]; s2 += q[2]; s3 += q[3];
                s0 += q[4]; s1 += q[5]; s2 += q[6]; s3 += q[7];
            }
            for (; z < ns; ++z) s0 += sl[(size_t)z * nn];
            gi = (s0 + s1) + (s2 + s3);
            g[i] = gi;
        } else {
            gi = g[i];
        }
        float mi = m[i], vi = v[i];
        p[i] = clamp_adam_elem(gi, p[i], mi, vi, step_size, bc2_sqrt, beta1, beta2, eps, wd, clampv, gscale);
        m[i] = mi; v[i] = vi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (atomicAdd(&st->ticket, 1u) == gridDim.x - 1) {
            st->b1t = b1t; st->b2t = b2t; st->step = step; st->step_size = step_size; st->bc2_sqrt = bc2_sqrt;
            atomicExch(&st->ticket, 0u);
        }
    }
}

extern "C" int ivosw_replay_draw_gather(const float* old_iou, const float* new_iou, const float* annotated, const float* next_annotated,
                                        const int64_t* action, const float* reward_step, const float* reward_done, void* draw_state, int n,
                                        int B, int T, int64_t* idx_out, float* state, float* new_state, int64_t* action_out,
                                        float* reward_step_out, float* reward_done_out, ivosw_stream_t stream);
extern "C" int ivosw_clamp_adam_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n, void* adam_state, float lr,
                                    float beta1, float beta2, float eps, float weight_decay, float clamp, float grad_scale,
                                    ivosw_stream_t stream);

extern "C" int ivosw_dqn_step_drawn(float* policy, const float* target, const float* old_iou, const float* new_iou, const float* annotated,
                                    const float* next_annotated, const int64_t* action, const float* reward_step, const float* reward_done,
                                    void* draw_state, int n, int B, int T, float gamma, int64_t* idx_out, float* state, float* new_state,
                                    int64_t* action_out, float* reward_step_out, float* reward_done_out, float* grads, float* loss, void* ws,
                                    size_t ws_bytes, float* exp_avg, float* exp_avg_sq, void* adam_state, float lr, float beta1, float beta2,
                                    float eps, float weight_decay, float clamp, float grad_scale, ivosw_stream_t stream) {
    IVOSW_REQUIRE(policy && target && old_iou && new_iou && annotated && next_annotated && action && reward_step && reward_done && draw_state &&
                      idx_out && state && new_state && action_out && reward_step_out && reward_done_out && grads && loss && ws && exp_avg &&
                      exp_avg_sq && adam_state,
                  "null pointer");
    IVOSW_ON_DEVICE_OF(grads);
    IVOSW_REQUIRE(n > 0 && B > 0 && T > 0, "n, B and T must be positive");
    const bool aligned = ((reinterpret_cast<uintptr_t>(policy) | reinterpret_cast<uintptr_t>(grads) | reinterpret_cast<uintptr_t>(exp_avg) |
                           reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15) == 0;
    bool folded = false;
    if (aligned && tune_get("DQN_ONECALL", 1)) {
        const EncDraw dr{old_iou, new_iou, annotated, next_annotated, action, reward_step, reward_done, static_cast<DrawState*>(draw_state), n, B, T,
                         idx_out, state, new_state, action_out, reward_step_out, reward_done_out};
        ReduceGroup rg{};
        const int rc = dqn_loss_grad_impl(policy, target, state, new_state, action_out, reward_step_out, reward_done_out, B, T, gamma, grads, loss,
                                          ws, ws_bytes, stream, &dr, &rg, &folded);
        if (rc != IVOSW_OK) return rc;
        if (folded) {
            ReduceOffsets ro{};
            for (int k = 0; k < REDUCE_MAX; ++k) ro.off[k] = rg.out[k] ? (int)(rg.out[k] - grads) : 0;
            const int nprm = IVOSW_BRAIN_NPARAMS;
            hipLaunchKernelGGL(clamp_adam_dev_reduce_kernel, dim3((nprm + 1023) / 1024), dim3(1024), 0, as_stream(stream), policy, grads,
                               exp_avg, exp_avg_sq, nprm, static_cast<AdamDevState*>(adam_state), rg, ro, lr, beta1, beta2, eps, weight_decay,
                               clamp, grad_scale);
            IVOSW_CHECK_LAUNCH();
            return IVOSW_OK;
        }
    }
    // the un-folded sequence (a tunable moved the step off the fused launch chain): the same three entries the caller would have made
    int rc = ivosw_replay_draw_gather(old_iou, new_iou, annotated, next_annotated, action, reward_step, reward_done, draw_state, n, B, T, idx_out,
                                      state, new_state, action_out, reward_step_out, reward_done_out, stream);
    if (rc == IVOSW_OK)
        rc = ivosw_dqn_loss_grad(policy, target, state, new_state, action_out, reward_step_out, reward_done_out, B, T, gamma, grads, loss, ws,
                                 ws_bytes, stream);
    if (rc == IVOSW_OK)
        rc = ivosw_clamp_adam_dev(policy, grads, exp_avg, exp_avg_sq, IVOSW_BRAIN_NPARAMS, adam_state, lr, beta1, beta2, eps, weight_decay, clamp,
                                  grad_scale, stream);
    return rc;
}

/* Tuning probe: subsequent fused forwards stamp s_memtime at four points of recurrence step T/2 per workgroup into ts
 * ([workgroups, 8] uint64 on the device: 0-3 the step's four points, 4 kernel entry, 5 weights in registers, 6 last step done; NULL switches the probe off).  tools/lstm_probe.py. */
#ifdef IVOSW_PROBES
extern "C" int ivosw_lstm_probe(unsigned long long* ts, unsigned long long* ts_bwd) {
    g_lstm_probe = ts;
    g_lstm_probe_bwd = ts_bwd;          // BPTT kernel: 0 step start, 1 gate gradients written, 2 barrier passed, 3 matvec done, 4 / 5 loop start / end, 6 steps
    return IVOSW_OK;
}
#endif  // IVOSW_PROBES
