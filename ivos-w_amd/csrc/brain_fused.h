// Launch-collapsed pieces of the Brain forward and the Double-DQN head (tunable DQN_FUSED, default on).
//
// A DQN step at B = 128, T = 25 is ~10 GFLOP spread over a chain of dependent launches, each a few microseconds of work
// on a fraction of the chip: what it costs is the NUMBER of links in the chain (every kernel pays its own ramp-up, drain and
// a round trip of its operands through L2).  The kernels below collapse the non-recurrent links:
//
//   enc_fused_kernel   encoder_fc1 + relu + encoder_fc2 + the input-side LSTM gates W_ih e_t   (models/agent.py:49-52)
//                      for the policy batch [s'; s] AND the target batch s' in one launch (was 2 x 3 launches on two streams)
//   dec_fused_kernel   relu -> decoder_fc1 -> relu -> decoder_fc2                              (models/agent.py:60-62)
//                      for both nets in one launch (was 2 x 2)
//   head_fused_kernel  Double-DQN targets, loss, dL/dQ, the decoder backward on the B rows that carry loss and
//                      dL/d(h_fw|h_bw) at the loss frame                                       (models/agent.py:135-151)
//                      (was dqn_head + dec_bwd_rows + a 128 x 256 x 128 GEMM)
//
// Contractions run on v_mfma_f32_16x16x4_f32 (exact fp32: an fma chain per accumulator).  A workgroup owns 48 rows (three
// 16-row MFMA tiles: 9 600 rows of a step = 200 workgroups on 256 CUs; 64-row tiles would leave 106 CUs idle) and keeps its
// activations in LDS from the two input scalars to the 512 gate pre-activations; a wave owns a slice of the output columns, so
// each weight is needed by ONE wave of the workgroup and goes from L2 straight into its registers (16-byte loads along K).
// K order: lane (i, kq) of an MFMA step carries k = 16 c + 4 kq + step — A and B agree, which is all a contraction needs.
#pragma once
#include "adam.h"
#include "common.h"

namespace ivosw {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int FM = 48;            // rows per workgroup
constexpr int F_LD = 128 + 4;     // padded LDS row: a wave's 16-byte fragment reads of 16 rows fall on disjoint banks
constexpr int D_LD = 256 + 4;

struct EncJob {
    const float* prm;   // parameter arena of the net this job runs
    const float* x;     // rows [0, rows0): x[row*2 + c]
    const float* x2;    // rows [rows0, rows): x2[(row - rows0)*2 + c]
    float* gx;          // [rows, 512]
    float* a1;          // [rows, 128], written for rows >= keep_row (backward needs them for the `state` half only)
    float* e;           // [rows, 128], same
    int rows0, rows, keep_row;
};
// The minibatch DRAWN AND GATHERED by the encoder launch itself (ivosw_dqn_step_drawn: one link less in the step's launch chain
// than ivosw_replay_draw_gather in front of it).  Row r of a job is frame r % T of sample r / T; the sample's replay row is the
// draw of slot r / T (adam.h: draw_mix), its two input scalars come straight from the replay columns, and the policy job leaves
// the gathered minibatch behind for the head / tail kernels and the caller.  ds == nullptr: rows come from EncJob::x / x2.
struct EncDraw {
    const float *old_iou, *new_iou, *ann, *nann;       // [n, T] replay columns
    const int64_t* action;                              // [n]
    const float *rstep, *rdone;
    DrawState* ds;
    int n, B, T;
    int64_t* idx_out;                                   // [B] rows drawn
    float *state, *new_state;                           // [B, T, 2]
    int64_t* action_out;
    float *rstep_out, *rdone_out;
};
struct EncGroup {
    EncJob j[2];
    int first1;         // first workgroup of job 1 (grid size when there is one job)
    EncDraw dr;
};

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// phase stamps (tools/ubench/enc_probe.hip builds this header with IVOSW_FUSED_PROBE; the library never does)
#ifdef IVOSW_FUSED_PROBE
__device__ unsigned long long g_fused_probe[512][8];
#define FPROBE(i) do { if (threadIdx.x == 0) g_fused_probe[blockIdx.x][i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FPROBE(i) do { } while (0)
#endif

__global__ __launch_bounds__(256) void enc_fused_kernel(EncGroup grp, int o_w1, int o_b1, int o_w2, int o_b2, int o_wih) {
    __shared__ __attribute__((aligned(16))) float a_s[FM][F_LD];
    __shared__ __attribute__((aligned(16))) float e_s[FM][F_LD];
    __shared__ __attribute__((aligned(8))) float x_s[FM * 2];
    const int which = (int)blockIdx.x >= grp.first1;
    const EncJob& jb = grp.j[which];
    const int r0 = ((int)blockIdx.x - (which ? grp.first1 : 0)) * FM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m16 = lane & 15, kq = lane >> 4;
    const float* __restrict__ prm = jb.prm;
    FPROBE(0);
    // Every global operand the tile needs before its last phase is requested up front (one wave per SIMD: nothing else
    // hides an L2 round trip): the tile's 96 input scalars, this wave's 32 x 128 slice of W2 (16 float4 per lane) and the
    // first K-chunk of its 128 x 128 slice of W_ih.
    float xv = 0.f;
    unsigned draw_counter = 0;
    if (grp.dr.ds) draw_counter = grp.dr.ds->counter;       // read before this workgroup's ticket (below): the last ticket advances it
    if (tid < FM * 2) {
        const int row = min(r0 + (tid >> 1), jb.rows - 1), c = tid & 1;
        if (grp.dr.ds) {
            const EncDraw& d = grp.dr;
            const bool second = row >= jb.rows0;            // policy job: rows [0, B T) = s' (new state), [B T, 2 B T) = s
            const int rr = second ? row - jb.rows0 : row;
            const int b = rr / d.T, t = rr - b * d.T;
            const int64_t src = (int64_t)__umul64hi(draw_mix(d.ds->seed, draw_counter, (unsigned)b), (unsigned long long)d.n);
            const float* col = second ? (c ? d.ann : d.old_iou) : (c ? d.nann : d.new_iou);
            xv = col[(size_t)src * d.T + t];
            if (which == 0 && r0 + (tid >> 1) < jb.rows) {
                (second ? d.state : d.new_state)[(size_t)rr * 2 + c] = xv;
                if (second && t == 0 && c == 0) {
                    d.idx_out[b] = src;
                    d.action_out[b] = d.action[src];
                    d.rstep_out[b] = d.rstep[src];
                    d.rdone_out[b] = d.rdone[src];
                }
            }
        } else {
            xv = row < jb.rows0 ? jb.x[(size_t)row * 2 + c] : jb.x2[(size_t)(row - jb.rows0) * 2 + c];
        }
    }
    const int j1 = tid & 127;
    const float w10 = prm[o_w1 + 2 * j1], w11 = prm[o_w1 + 2 * j1 + 1], b1v = prm[o_b1 + j1];
    float4 w2r[8][2];
    {
        const float* wb = prm + o_w2 + (size_t)(32 * wave + m16) * 128 + 4 * kq;
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) w2r[c][nt] = *reinterpret_cast<const float4*>(wb + (size_t)nt * 16 * 128 + 16 * c);
    }
    const float* wih = prm + o_wih + (size_t)(128 * wave + m16) * 128 + 4 * kq;
    float4 bn[8];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) bn[nt] = *reinterpret_cast<const float4*>(wih + (size_t)nt * 16 * 128);
    const float b2v[2] = {prm[o_b2 + 32 * wave + m16], prm[o_b2 + 32 * wave + 16 + m16]};
    __builtin_amdgcn_sched_barrier(0);      // (hipcc otherwise sinks loads to their first use and waits out the round trip there)
    if (tid < FM * 2) x_s[tid] = xv;
    __syncthreads();
    FPROBE(1);
    // encoder_fc1 + relu: thread = (output unit j, row parity)
    {
        float2 xr[FM / 2];          // all 24 rows' inputs first: the loop was a chain of LDS round trips (7.6 k cycles)
#pragma unroll
        for (int i = 0; i < FM / 2; ++i) xr[i] = *reinterpret_cast<const float2*>(&x_s[2 * ((tid >> 7) + 2 * i)]);
#pragma unroll
        for (int i = 0; i < FM / 2; ++i) {
            const int r = (tid >> 7) + 2 * i;
            const float v = fmaxf(fmaf(xr[i].y, w11, fmaf(xr[i].x, w10, b1v)), 0.f);
            a_s[r][j1] = v;
            if (r0 + r < jb.rows && r0 + r >= jb.keep_row) jb.a1[(size_t)(r0 + r) * 128 + j1] = v;
        }
    }
    __syncthreads();
    FPROBE(2);
#define MFMA_STEP(comp, NT)                                                                  \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                         \
        _Pragma("unroll") for (int mt = 0; mt < 3; ++mt) acc[mt][nt] = MFMA16(a[mt].comp, b[nt].comp, acc[mt][nt]);
    {   // e = a1 W2^T + b2: wave w owns output columns [32 w, 32 w + 32)
        f32x4 acc[3][2];
#pragma unroll
        for (int mt = 0; mt < 3; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float4 a[3];
            const float4* b = w2r[c];
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) a[mt] = *reinterpret_cast<const float4*>(&a_s[mt * 16 + m16][16 * c + 4 * kq]);
            MFMA_STEP(x, 2) MFMA_STEP(y, 2) MFMA_STEP(z, 2) MFMA_STEP(w, 2)
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int col = 32 * wave + 16 * nt + m16;
            const float bias = b2v[nt];
#pragma unroll
            for (int mt = 0; mt < 3; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = mt * 16 + 4 * kq + r;
                    const float v = acc[mt][nt][r] + bias;
                    e_s[row][col] = v;
                    if (r0 + row < jb.rows && r0 + row >= jb.keep_row) jb.e[(size_t)(r0 + row) * 128 + col] = v;
                }
        }
    }
    __syncthreads();
    FPROBE(3);
    {   // gx = e W_ih^T: wave w owns gate columns [128 w, 128 w + 128) = gate w of every hidden unit
        f32x4 acc[3][8];
#pragma unroll
        for (int mt = 0; mt < 3; ++mt)
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < 8; ++c) {
            float4 a[3], b[8];
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) b[nt] = bn[nt];
            if (c + 1 < 8) {
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) bn[nt] = *reinterpret_cast<const float4*>(wih + (size_t)nt * 16 * 128 + 16 * (c + 1));
            }
            // hipcc otherwise sinks these loads to their first use, one chunk later, and every chunk waits out an L2 round trip
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) a[mt] = *reinterpret_cast<const float4*>(&e_s[mt * 16 + m16][16 * c + 4 * kq]);
            MFMA_STEP(x, 8) MFMA_STEP(y, 8) MFMA_STEP(z, 8) MFMA_STEP(w, 8)
        }
        FPROBE(4);
#pragma unroll
        for (int mt = 0; mt < 3; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = r0 + mt * 16 + 4 * kq + r;
                if (row < jb.rows) {
                    float* o = jb.gx + (size_t)row * 512 + 128 * wave + m16;
#pragma unroll
                    for (int nt = 0; nt < 8; ++nt) o[16 * nt] = acc[mt][nt][r];
                }
            }
        FPROBE(5);
    }
    if (grp.dr.ds && tid == 0) {                     // the LAST workgroup to finish advances the draw counter (every workgroup read it at its start)
        DrawState* ds = grp.dr.ds;
        if (atomicAdd(&ds->ticket, 1u) == gridDim.x - 1) {
            ds->counter = draw_counter + 1;
            atomicExch(&ds->ticket, 0u);
        }
    }
}

struct DecJob {
    const float* prm;
    const float* hs;    // [rows, 256] (fw | bw)
    float* d1;          // [rows, 128], written for rows >= keep_row
    float* q;           // [rows]
    int rows, keep_row;
};
struct DecGroup {
    DecJob j[2];
    int first1;
};

// 16-lane (one MFMA column group) sum: lanes l, l^1, l^2, l^4, l^8 stay inside a DPP row
__device__ __forceinline__ float row16_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}

__global__ __launch_bounds__(256) void dec_fused_kernel(DecGroup grp, int o_w3, int o_b3, int o_w4, int o_b4) {
    __shared__ __attribute__((aligned(16))) float h_s[FM][D_LD];
    __shared__ float q_s[4][FM];
    const int which = (int)blockIdx.x >= grp.first1;
    const DecJob& jb = grp.j[which];
    const int r0 = ((int)blockIdx.x - (which ? grp.first1 : 0)) * FM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* __restrict__ prm = jb.prm;
    const int m16 = lane & 15, kq = lane >> 4;
    float4 w3r[16][2];                                  // this wave's 32 x 256 slice of W3, requested before anything else
    {
        const float* wb = prm + o_w3 + (size_t)(32 * wave + m16) * 256 + 4 * kq;
#pragma unroll
        for (int c = 0; c < 16; ++c)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) w3r[c][nt] = *reinterpret_cast<const float4*>(wb + (size_t)nt * 16 * 256 + 16 * c);
    }
    const float b3v[2] = {prm[o_b3 + 32 * wave + m16], prm[o_b3 + 32 * wave + 16 + m16]};
    const float w4v[2] = {prm[o_w4 + 32 * wave + m16], prm[o_w4 + 32 * wave + 16 + m16]};
    const float b4v = prm[o_b4];
    {
        float4 hv[FM * 64 / 256];
#pragma unroll
        for (int i = 0; i < FM * 64 / 256; ++i) {       // relu(h_fw | h_bw) of the tile's rows -> LDS, 16 bytes per lane
            const int f = tid + 256 * i, r = f >> 6, c4 = f & 63;
            const int row = min(r0 + r, jb.rows - 1);
            hv[i] = *reinterpret_cast<const float4*>(jb.hs + (size_t)row * 256 + 4 * c4);
        }
#pragma unroll
        for (int i = 0; i < FM * 64 / 256; ++i) {
            const int f = tid + 256 * i, r = f >> 6, c4 = f & 63;
            const float4 v = hv[i];
            *reinterpret_cast<float4*>(&h_s[r][4 * c4]) = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    f32x4 acc[3][2];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        float4 a[3];
        const float4* b = w3r[c];
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) a[mt] = *reinterpret_cast<const float4*>(&h_s[mt * 16 + m16][16 * c + 4 * kq]);
        MFMA_STEP(x, 2) MFMA_STEP(y, 2) MFMA_STEP(z, 2) MFMA_STEP(w, 2)
    }
    // d1 = relu(. + b3); q = d1 . w4 + b4: a lane's partial over its two columns, the 16 column lanes summed by DPP-row
    // shuffles, the four waves' 32-column partials through LDS in a fixed order
    float part[3][4];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[mt][r] = 0.f;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int col = 32 * wave + 16 * nt + m16;
        const float bias = b3v[nt], w4 = w4v[nt];
#pragma unroll
        for (int mt = 0; mt < 3; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = r0 + mt * 16 + 4 * kq + r;
                const float v = fmaxf(acc[mt][nt][r] + bias, 0.f);
                if (row < jb.rows && row >= jb.keep_row) jb.d1[(size_t)row * 128 + col] = v;
                part[mt][r] = fmaf(v, w4, part[mt][r]);
            }
    }
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float s = row16_sum(part[mt][r]);
            if (m16 == 0) q_s[wave][mt * 16 + 4 * kq + r] = s;
        }
    __syncthreads();
    if (tid < FM && r0 + tid < jb.rows)
        jb.q[r0 + tid] = ((q_s[0][tid] + q_s[1][tid]) + (q_s[2][tid] + q_s[3][tid])) + b4v;
}

// One workgroup per sample b: Double-DQN target (first maximum of the policy's Q over s', the target net's Q there), the two
// MSE terms' gradient dL/dQ(s, a), then the decoder backward on row (b, a): dd1 = dq w4 . (d1 > 0), the dW4 term dq d1,
// relu(h) for dW3, and dL/dh = (dd1 W3) . (h > 0) as a 128-long dot product per thread.  Block 0 also forms the batch
// sums (loss, db4) with the reduction tree of dqn_head_kernel.
__global__ __launch_bounds__(256) void head_fused_kernel(const float* __restrict__ prm, int o_w3, int o_w4,
                                                         const float* __restrict__ q_np, const float* __restrict__ q_nt,
                                                         const float* __restrict__ q_s, const int64_t* __restrict__ action,
                                                         const float* __restrict__ r_step, const float* __restrict__ r_done,
                                                         int B, int T, float gamma, const float* __restrict__ d1_s,
                                                         const float* __restrict__ hs_s, float* __restrict__ dq,
                                                         float* __restrict__ dd1c, float* __restrict__ w4term,
                                                         float* __restrict__ hcc, float* __restrict__ dhc,
                                                         float* __restrict__ loss, float* __restrict__ db4) {
    __shared__ float dd_s[128];
    __shared__ float d_sh;
    __shared__ float red[2][256];
    const int b = blockIdx.x, tid = threadIdx.x;
    int a = (int)action[b];
    a = min(max(a, 0), T - 1);
    // Everything that does not depend on the argmax is requested up front (the kernel is a chain of L2 round trips
    // otherwise): this thread's column of W3 (128 registers), the loss row of d1 / h, the scalars of the targets.
    float w3c[128];
    {
        const float* w = prm + o_w3 + tid;
#pragma unroll
        for (int j = 0; j < 128; ++j) w3c[j] = w[(size_t)j * 256];
    }
    const size_t row = (size_t)b * T + a;
    const float hv = hs_s[row * 256 + tid];
    const float d1v = tid < 128 ? d1_s[row * 128 + tid] : 0.f;
    const float w4v = tid < 128 ? prm[o_w4 + tid] : 0.f;
    const float rs = r_step[b], rd = r_done[b], qsa = q_s[(size_t)b * T + a];
    __builtin_amdgcn_sched_barrier(0);
    if (tid < 64) {
        const float* qp = q_np + (size_t)b * T;
        float best = -INFINITY;
        int am = 0x7fffffff;
        for (int t = tid; t < T; t += 64) {
            const float v = qp[t];
            if (v > best || am == 0x7fffffff) { best = v; am = t; }     // first maximum of this lane's frames
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oa = __shfl_xor(am, o, 64);
            if (oa != 0x7fffffff && (am == 0x7fffffff || ob > best || (ob == best && oa < am))) { best = ob; am = oa; }
        }
        if (tid == 0) {
            const float qn = q_nt[(size_t)b * T + am];
            const float y1 = qn * gamma + rs * 0.1f;
            const float y2 = rd * 0.1f;
            const float e1 = qsa - y1, e2 = qsa - y2;
            const float d = (2.0f / (float)B) * (e1 + e2);
            d_sh = d;
            dq[b] = d;
        }
    }
    __syncthreads();
    const float d = d_sh;
    if (tid < 128) {
        w4term[(size_t)b * 128 + tid] = d * d1v;
        const float dd = (d1v > 0.f) ? d * w4v : 0.f;
        dd1c[(size_t)b * 128 + tid] = dd;
        dd_s[tid] = dd;
    }
    const float h = fmaxf(hv, 0.f);
    hcc[(size_t)b * 256 + tid] = h;
    __syncthreads();
    {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int j = 0; j < 128; j += 4) {
            a0 = fmaf(dd_s[j], w3c[j], a0);
            a1 = fmaf(dd_s[j + 1], w3c[j + 1], a1);
            a2 = fmaf(dd_s[j + 2], w3c[j + 2], a2);
            a3 = fmaf(dd_s[j + 3], w3c[j + 3], a3);
        }
        dhc[(size_t)b * 256 + tid] = (h > 0.f) ? (a0 + a1) + (a2 + a3) : 0.f;
    }
    if (b != 0) return;
    // batch sums, the arithmetic and order of dqn_head_kernel
    float l = 0.f, sdq = 0.f;
    for (int s = tid; s < B; s += 256) {
        const float* qp = q_np + (size_t)s * T;
        int am = 0;
        float best = qp[0];
        for (int t = 1; t < T; ++t) {
            const float v = qp[t];
            if (v > best) { best = v; am = t; }
        }
        const float qn = q_nt[(size_t)s * T + am];
        const float y1 = qn * gamma + r_step[s] * 0.1f;
        const float y2 = r_done[s] * 0.1f;
        int as = (int)action[s];
        as = min(max(as, 0), T - 1);
        const float qsa = q_s[(size_t)s * T + as];
        const float e1 = qsa - y1, e2 = qsa - y2;
        l += e1 * e1 + e2 * e2;
        sdq += (2.0f / (float)B) * (e1 + e2);
    }
    red[0][tid] = l;
    red[1][tid] = sdq;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            red[0][tid] += red[0][tid + o];
            red[1][tid] += red[1][tid + o];
        }
        __syncthreads();
    }
    if (tid == 0) {
        *loss = red[0][0] / (float)B;
        *db4 = red[1][0];
    }
}

}  // namespace ivosw
