// Convolution tower kernels for the quality-assessment CNN (K4/K5/K6): NHWC implicit-GEMM convolution on the
// matrix cores with BN folded into the weights, bias/residual/ReLU fused in the epilogue; stem 7x7 (RGB+mask
// concatenated along Cin); 3x3/2 max-pool; 8x8 average pool + fc.
//
// Reference arithmetic: Encoder.forward (models/assessment.py:46-63) over torchvision's ResNet-50 v1.5
// bottlenecks, avg_pool2d(.,8) + fc1 (:179-180).
//
// GEMM view: M = B*Ho*Wo output pixels, N = Cout, K = KH*KW*Cin.  A K-tile is 128 bytes of one filter tap
// (64 bf16 / 32 fp32 channels), so an A row is one contiguous 128-B NHWC segment (or zeros at the border)
// and a B row is 128 B of the K-major packed weights.  Tiles are staged global -> registers -> LDS
// (double-buffered, one barrier per K-tile, next tile's loads in flight under the MFMAs); the LDS image is
// XOR-swizzled at 16-B granularity (chunk ^= (row>>1)&7) so both the 8-lane ds_write_b128 groups and the
// 16-lane ds_read_b128 groups are bank-conflict free.  bf16 mode: v_mfma_f32_32x32x16_bf16; fp32 parity
// mode: v_mfma_f32_32x32x2_f32 (exact fp32 fma chain).  The accumulator tile goes through LDS once so that
// global stores (and residual loads) are full 16-B-per-lane row segments.
#include <stdlib.h>

#include <type_traits>
#include <utility>
#include <vector>

#include "common.h"
#include "conv.h"
#include "mfma_tile.h"

namespace ivosw {


// eight fp32 values -> four packed bf16 hi pairs (truncation: the top halves) and four packed lo pairs (RNE(x - hi), an exact
// difference): the producer-side split of the IVOSW_F32X3 activation format, ~3 VALU per value ONCE instead of 24 per 8-float
// fragment in every wave that consumes it
__device__ __forceinline__ void split8_store_x3(const float (&v)[8], uint32_t (&hi)[4], uint32_t (&lo)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned a = __float_as_uint(v[2 * q]), b = __float_as_uint(v[2 * q + 1]);
        hi[q] = __builtin_amdgcn_perm(b, a, 0x07060302u);
        lo[q] = pack2_bf16(v[2 * q] - __uint_as_float(a & 0xffff0000u), v[2 * q + 1] - __uint_as_float(b & 0xffff0000u));
    }
}

// epilogue phase 2: 8 consecutive channels per thread: + bias (+ residual) -> ReLU -> 16/32-B stores
template <typename T, int BM, int BN, int NT = 256, bool PRE = false>
__device__ __forceinline__ void epilogue_store(const ConvArgs& p, const float* Cs, int m0, int n0, int M, int tid,
                                               const uint4* pre = nullptr) {
    constexpr int ES = (int)sizeof(T);
    constexpr int CPR = BN / 8;  // 8-channel groups per row
    static_assert(NT % CPR == 0, "a thread keeps one channel group across its items");
    T* Y = static_cast<T*>(p.y);
    const T* R = static_cast<const T*>(p.res);
    // the thread's 8 output channels are the same for all of its items: one bias fetch
    const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n0 + (tid % CPR) * 8);
    const float4 b1 = *reinterpret_cast<const float4*>(p.bias + n0 + (tid % CPR) * 8 + 4);
    // IVOSW_F32X3 residual: ALL items' (hi, lo) chunks are requested before the first item is processed (round 6).  Inside the item loop a
    // load sits behind the previous item's stores, which the compiler must assume may alias: a chain of ITEMS dependent round trips per thread
    constexpr int XIT = (ES == 4) ? (BM * CPR) / NT : 1;
    u32x4 xrh[XIT], xrl[XIT];
    const bool xres = ES == 4 && R != nullptr && !ABL(p.debug, 1);              // (plain fp32 tensors too: the item's two float4)
    if constexpr (ES == 4) {
        if (xres) {
#pragma unroll
            for (int it = 0; it < XIT; ++it) {
                const int item = it * NT + tid;
                const int row = item / CPR, cg = item - row * CPR;
                const int m = min(m0 + row, M - 1);
                const int n = n0 + cg * 8;
                if (p.x3 != 2) {
                    const u32x4* rf = reinterpret_cast<const u32x4*>(R + (long)m * p.Cout + n);
                    xrh[it] = rf[0]; xrl[it] = rf[1];
                    continue;
                }
                const u32x4* rs = reinterpret_cast<const u32x4*>(R + (long)m * p.Cout + (n & ~31));
                if (p.nt & 8) { xrh[it] = __builtin_nontemporal_load(rs + ((n & 31) >> 3)); xrl[it] = __builtin_nontemporal_load(rs + 4 + ((n & 31) >> 3)); }
                else { xrh[it] = rs[(n & 31) >> 3]; xrl[it] = rs[4 + ((n & 31) >> 3)]; }
            }
        }
    }
#pragma unroll
    for (int it = 0; it < (BM * CPR) / NT; ++it) {
        const int item = it * NT + tid;
        const int row = item / CPR, cg = item - row * CPR;
        const int m = m0 + row;
        if (m >= M) continue;
        const int n = n0 + cg * 8;
        const float4 v0 = *reinterpret_cast<const float4*>(Cs + row * BN + cg * 8);
        const float4 v1 = *reinterpret_cast<const float4*>(Cs + row * BN + cg * 8 + 4);
        float v[8] = {v0.x + b0.x, v0.y + b0.y, v0.z + b0.z, v0.w + b0.w, v1.x + b1.x, v1.y + b1.y, v1.z + b1.z, v1.w + b1.w};
        const long o = (long)m * p.Cout + n;
        if ABL(p.debug, 1) { if (v0.x == 1.2345e30f) Y[o] = T(0); continue; }
        if constexpr (ES == 2) {
            if (R) {
                uint4 rr;
                if constexpr (PRE) rr = pre[it];
                else if (p.nt & 2) { const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(R + o)); rr = make_uint4(t[0], t[1], t[2], t[3]); }
                else rr = *reinterpret_cast<const uint4*>(R + o);
                const uint32_t w4[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[2 * q] += bf16_to_f32((bf16_t)(w4[q] & 0xffff));
                    v[2 * q + 1] += bf16_to_f32((bf16_t)(w4[q] >> 16));
                }
            }
            uint32_t pk[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                pk[q] = act2_bf16(v[2 * q], v[2 * q + 1], p.relu != 0);
            }
            if (p.nt & 1) {
                u32x4 ov = {pk[0], pk[1], pk[2], pk[3]};
                __builtin_nontemporal_store(ov, reinterpret_cast<u32x4*>(Y + o));
            } else {
                *reinterpret_cast<uint4*>(Y + o) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            }
        } else if (p.x3 == 2) {
            // IVOSW_F32X3 activations are stored SPLIT (round 5): a 32-channel group is 128 bytes, [32 x bf16 hi | 32 x bf16 lo] - the
            // bytes of 32 floats, so every offset, K-tile and DMA piece of the fp32 layout stays what it is.  This thread's 8 channels
            // are 16-byte chunk c of the group: hi at chunk c, lo at chunk 4 + c
            const long og = (long)m * p.Cout + (n & ~31);                      // in floats
            const int c = (n & 31) >> 3;
            const uint4* rs = reinterpret_cast<const uint4*>(R + og);
            if (R) {
                const u32x4 rh = xrh[ES == 4 ? it : 0], rl = xrl[ES == 4 ? it : 0];
                const uint32_t h4[4] = {rh[0], rh[1], rh[2], rh[3]}, l4[4] = {rl[0], rl[1], rl[2], rl[3]};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[2 * q] += __uint_as_float(h4[q] << 16) + __uint_as_float(l4[q] << 16);
                    v[2 * q + 1] += __uint_as_float(h4[q] & 0xffff0000u) + __uint_as_float(l4[q] & 0xffff0000u);
                }
            }
            if (p.relu) {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
            }
            uint32_t hi[4], lo[4];
            split8_store_x3(v, hi, lo);
            uint4* ys = reinterpret_cast<uint4*>(Y + og);
            if (p.nt & 4) {
                const u32x4 vh = {hi[0], hi[1], hi[2], hi[3]}, vl = {lo[0], lo[1], lo[2], lo[3]};
                __builtin_nontemporal_store(vh, reinterpret_cast<u32x4*>(ys + c));
                __builtin_nontemporal_store(vl, reinterpret_cast<u32x4*>(ys + 4 + c));
            } else {
                ys[c] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
                ys[4 + c] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
            }
        } else {
            if (R) {
                const u32x4 r0v = xrh[ES == 4 ? it : 0], r1v = xrl[ES == 4 ? it : 0];
                v[0] += __uint_as_float(r0v[0]); v[1] += __uint_as_float(r0v[1]); v[2] += __uint_as_float(r0v[2]); v[3] += __uint_as_float(r0v[3]);
                v[4] += __uint_as_float(r1v[0]); v[5] += __uint_as_float(r1v[1]); v[6] += __uint_as_float(r1v[2]); v[7] += __uint_as_float(r1v[3]);
            }
            if (p.relu) {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
            }
            *reinterpret_cast<float4*>(Y + o) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(Y + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
    }
}

// ---------------------------------------------------------------- split-bf16 contraction of one fp32 K-tile (IVOSW_F32X3)
// A K-tile is 32 fp32 activations per row (128 B) and, for the weights, the same 32 k as [32 x bf16 hi | 32 x bf16 lo] (128 B, split
// at pack time: launch_split_weights_x3).  Two 32x32x16 MFMA steps per K-tile; in step s lane half h supplies k = 16 s + 8 h + [0, 8):
// activations from 16-byte chunks 4 s + 2 h and 4 s + 2 h + 1 (eight floats, split here: hi = truncation — one v_perm per pair —,
// lo = RNE(x - hi), exact difference), weights from chunk 2 s + h (hi) and 4 + 2 s + h (lo).  Three products per operand pair:
// ah bh + ah bl + al bh; the dropped al bl is ~ 2^-17 of the product, like the rounding of the lo parts.
// Special values: x = +-Inf splits into hi = +-Inf, lo = Inf - Inf = NaN, so an activation that OVERFLOWED to
// infinity comes out of this mode as NaN where the fp32 mode keeps Inf; NaN stays NaN in both.  Finite inputs are unaffected; the
// mode promises no Inf-propagation parity (three more VALU instructions per pair in a VALU-bound body would buy it).
__device__ __forceinline__ void split8_x3(const u32x4& c0, const u32x4& c1, u32x4& hi, u32x4& lo) {
    const unsigned x[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        hi[q] = __builtin_amdgcn_perm(x[2 * q + 1], x[2 * q], 0x07060302u);      // high halves of x[2q+1] | x[2q]
        const float l0 = __uint_as_float(x[2 * q]) - __uint_as_float(x[2 * q] & 0xffff0000u);
        const float l1 = __uint_as_float(x[2 * q + 1]) - __uint_as_float(x[2 * q + 1] & 0xffff0000u);
        lo[q] = pack2_bf16(l0, l1);
    }
}
// PRE: the activation K-tile is already in the split layout (every layer behind the stem: epilogue_store writes it) - hi and lo
// fragments are read like the weights', no VALU; !PRE (the stem, whose operand is gathered from the fp32 ROI tile): split here.
template <int TM, int TN, bool PRE>
__device__ __forceinline__ void ktile_mma_x3(f32x16 (&acc)[TM][TN], unsigned a_base, unsigned b_base, int wm, int wn, int lrow, int lhalf) {
    u32x4 a0[2][TM], a1[2][TM], bh[2][TN], bl[2][TN];
    auto frag_read = [&](int s, int buf) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            a0[buf][i] = lds_read_b128(a_base + swz((wm * TM + i) * 32 + lrow, PRE ? 2 * s + lhalf : 4 * s + 2 * lhalf));
            a1[buf][i] = lds_read_b128(a_base + swz((wm * TM + i) * 32 + lrow, PRE ? 4 + 2 * s + lhalf : 4 * s + 2 * lhalf + 1));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            bh[buf][j] = lds_read_b128(b_base + swz((wn * TN + j) * 32 + lrow, 2 * s + lhalf));
            bl[buf][j] = lds_read_b128(b_base + swz((wn * TN + j) * 32 + lrow, 4 + 2 * s + lhalf));
        }
    };
    frag_read(0, 0);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        lds_wait();
        if (s < 1) frag_read(1, 1);
        u32x4 ah[TM], al[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if constexpr (PRE) { ah[i] = a0[s][i]; al[i] = a1[s][i]; }
            else split8_x3(a0[s][i], a1[s][i], ah[i], al[i]);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[i][j] = mfma_bf16(ah[i], bh[s][j], acc[i][j]);
                acc[i][j] = mfma_bf16(ah[i], bl[s][j], acc[i][j]);
                acc[i][j] = mfma_bf16(al[i], bh[s][j], acc[i][j]);
            }
    }
}

// WAVES_M x WAVES_N waves (4 total), each owning TM x TN 32x32 MFMA tiles.
template <typename T, int WAVES_M, int WAVES_N, int TM, int TN, bool STEM, bool X3 = false>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs p) {
    static_assert(!X3 || sizeof(T) == 4, "the three-pass split is a mode of the fp32 tensors");
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    constexpr int ES = (int)sizeof(T);
    constexpr int KE = ROWB / ES;   // elements per K-tile (64 bf16 / 32 fp32)
    constexpr int CE = 16 / ES;     // elements per 16-B chunk
    constexpr int AI = BM / 32, BI = BN / 32;  // rows per thread for the A / B staging passes
    constexpr int STAGE_BYTES = (BM + BN) * ROWB;
    constexpr int EPI_BYTES = BM * BN * 4;
    constexpr int LDS_BYTES = (2 * STAGE_BYTES > EPI_BYTES) ? 2 * STAGE_BYTES : EPI_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int nbn = p.Cout / BN;
    const int nwg = gridDim.x;
    const int L = xcd_remap(blockIdx.x, nwg);
    const int tile_n = L % nbn, tile_m = L / nbn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int M = p.B * p.Ho * p.Wo;

    // ---- per-thread staging coordinates: chunk c of rows (tid>>3) + 32*i
    const int c = tid & 7;
    const int r0 = tid >> 3;
    long abase[AI];
    int aiy[AI], aix[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int m = m0 + r0 + 32 * i;
        if (m < M) {
            const int b = m / (p.Ho * p.Wo);
            const int rem = m - b * (p.Ho * p.Wo);
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            aiy[i] = oy * p.stride - p.pad;
            aix[i] = ox * p.stride - p.pad;
            abase[i] = (long)b * p.H * p.W * p.Cin;
        } else {
            aiy[i] = -100000; aix[i] = -100000; abase[i] = 0;
        }
    }
    const T* X = static_cast<const T*>(p.x);
    const T* Wt = static_cast<const T*>(p.w);
    const int K = STEM ? (ES == 2 ? 256 : 224) : p.KH * p.KW * p.Cin;
    const int nk = K / KE;
    const int ctiles = STEM ? 1 : p.Cin / KE;  // K-tiles per filter tap

    uint4 ra[AI], rb[BI];
    auto load_tiles = [&](int kt) {
        if constexpr (!STEM) {
            const int tap = kt / ctiles, c0 = (kt - tap * ctiles) * KE;
            const int ky = tap / p.KW, kx = tap - ky * p.KW;
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                const int iy = aiy[i] + ky, ix = aix[i] + kx;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
                    v = *reinterpret_cast<const uint4*>(X + abase[i] + ((long)iy * p.W + ix) * p.Cin + c0 + c * CE);
                ra[i] = v;
            }
        } else {
            // stem: Cin = 4 (R,G,B,P); one K-tile = filter row(s) of 8 pixels x 4 channels (8th pixel and, in
            // bf16, the 8th filter row are zero-weight padding)
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if constexpr (ES == 2) {
                    const int ky = 2 * kt + (c >> 2), px = (c & 3) * 2;
                    const int iy = aiy[i] + ky, ix = aix[i] + px;
                    if (ky < 7 && iy >= 0 && iy < p.H) {
                        const T* row = X + abase[i] + (long)iy * p.W * 4;
                        if (ix >= 0 && ix < p.W) { const uint2 t = *reinterpret_cast<const uint2*>(row + (long)ix * 4); v.x = t.x; v.y = t.y; }
                        if (ix + 1 >= 0 && ix + 1 < p.W && px + 1 < 7) { const uint2 t = *reinterpret_cast<const uint2*>(row + (long)(ix + 1) * 4); v.z = t.x; v.w = t.y; }
                    }
                } else {
                    const int iy = aiy[i] + kt, ix = aix[i] + c;
                    if (c < 7 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
                        v = *reinterpret_cast<const uint4*>(X + abase[i] + ((long)iy * p.W + ix) * 4);
                }
                ra[i] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < BI; ++i)
            rb[i] = *reinterpret_cast<const uint4*>(Wt + (long)(n0 + r0 + 32 * i) * K + (long)kt * KE + c * CE);
    };
    auto store_tiles = [&](int buf) {
        unsigned char* As = lds + buf * STAGE_BYTES;
        unsigned char* Bs = As + BM * ROWB;
#pragma unroll
        for (int i = 0; i < AI; ++i) *reinterpret_cast<uint4*>(As + swz(r0 + 32 * i, c)) = ra[i];
#pragma unroll
        for (int i = 0; i < BI; ++i) *reinterpret_cast<uint4*>(Bs + swz(r0 + 32 * i, c)) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    const int lrow = lane & 31, lhalf = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tiles(kt + 1);
        const unsigned char* As = lds + buf * STAGE_BYTES;
        const unsigned char* Bs = As + BM * ROWB;
        if constexpr (X3) {
            const unsigned lb = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds + buf * STAGE_BYTES;
            ktile_mma_x3<TM, TN, !STEM>(acc, lb, lb + BM * ROWB, wm, wn, lrow, lhalf);
        } else
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            // 16-B chunk (2*ks + lhalf): bf16 -> k = 16*ks + 8*lhalf + [0,8) (one 32x32x16 MFMA);
            // fp32 -> 4 floats feeding four 32x32x2 MFMAs (k-permutation identical for A and B)
            const int ch = 2 * ks + lhalf;
            uint4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const uint4*>(As + swz((wm * TM + i) * 32 + lrow, ch));
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const uint4*>(Bs + swz((wn * TN + j) * 32 + lrow, ch));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (ES == 2) {
                        union { uint4 u; bf16x8 v; } ua, ub;
                        ua.u = fa[i]; ub.u = fb[j];
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.v, ub.v, acc[i][j], 0, 0, 0);
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fa[i].x), __uint_as_float(fb[j].x), acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fa[i].y), __uint_as_float(fb[j].y), acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fa[i].z), __uint_as_float(fb[j].z), acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fa[i].w), __uint_as_float(fb[j].w), acc[i][j], 0, 0, 0);
                    }
                }
        }
        if (kt + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue phase 1: accumulators -> LDS fp32 tile [BM][BN] (C/D map: col = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5))
    float* Cs = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                const int col = (wn * TN + j) * 32 + lrow;
                Cs[row * BN + col] = acc[i][j][r];
            }
    __syncthreads();
    epilogue_store<T, BM, BN>(p, Cs, m0, n0, M, tid);
}

// ---------------------------------------------------------------- LDS-DMA pipelined variant (tower convs)
// Same GEMM view and LDS image as above, but tiles go HBM/L2 -> LDS directly (global_load_lds_dwordx4, 1 KiB per
// wave-instruction, no VGPR round trip) into an S-deep ring, with counted s_waitcnt vmcnt(N) and ONE raw
// s_barrier per K-tile, so S-1 tiles (up to 96 KB per CU) stay in flight under the MFMAs.  The DMA writes LDS
// lane-linearly (wave-uniform base + lane*16), so the XOR swizzle is applied on the SOURCE side: the lane that
// lands on chunk position p of row r fetches chunk p ^ ((r>>1)&7); reads use the same involution.  Zero padding:
// out-of-image taps fetch from a 16-B page of zeros.
template <typename T, int WAVES_M, int WAVES_N, int TM, int TN, int S, bool X3 = false>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64) void conv_igemm_dma_kernel(ConvArgs p) {
    static_assert(!X3 || sizeof(T) == 4, "the three-pass split is a mode of the fp32 tensors");
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    constexpr int ES = (int)sizeof(T);
    constexpr int KE = ROWB / ES, CE = 16 / ES;
    constexpr int AG = BM / 8 / NW, BG = BN / 8 / NW;  // 8-row groups (= DMA instructions) per wave per K-tile
    static_assert(AG * 8 * NW == BM && BG * 8 * NW == BN, "tile rows must split evenly over the waves");
    constexpr int LPT = AG + BG;
    constexpr int STAGE_BYTES = (BM + BN) * ROWB;
    constexpr int EPI_BYTES = BM * BN * 4;
    constexpr int LDS_BYTES = (S * STAGE_BYTES > EPI_BYTES) ? S * STAGE_BYTES : EPI_BYTES;
    static_assert(S >= 2 && S <= 4, "ring depth");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];  // the ONLY LDS object (keeps hipcc from draining vmcnt)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int nbn = p.Cout / BN;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = L % nbn, tile_m = L / nbn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int M = p.B * p.Ho * p.Wo;

    const T* X = static_cast<const T*>(p.x);
    const T* X2 = static_cast<const T*>(p.x2);
    const T* Wt = static_cast<const T*>(p.w);
    const int K = p.KH * p.KW * p.Cin + (X2 ? p.Cin2 : 0);
    const int nk = K / KE;
    const int nk1 = p.KH * p.KW * p.Cin / KE;        // K-tiles of the first source
    const int ctiles = p.Cin / KE;

    // ---- this lane's DMA duty: row (l>>3) of each of the wave's 8-row groups, chunk position (l&7)
    const int rsub = lane >> 3, cpos = lane & 7;
    long abase[AG], abase2[AG];
    int aiy[AG], aix[AG], achunk[AG];
#pragma unroll
    for (int i = 0; i < AG; ++i) {
        const int row = (wave * AG + i) * 8 + rsub;
        achunk[i] = (cpos ^ ((row >> 1) & 7)) * CE;
        const int m = m0 + row;
        abase2[i] = -1;
        if (m < M) {
            const int b = m / (p.Ho * p.Wo);
            const int rem = m - b * (p.Ho * p.Wo);
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            aiy[i] = oy * p.stride - p.pad;
            aix[i] = ox * p.stride - p.pad;
            abase[i] = (long)b * p.H * p.W * p.Cin;
            if (X2) abase2[i] = (((long)b * p.H2 + oy * p.stride2) * p.W2 + ox * p.stride2) * p.Cin2 + achunk[i];
        } else {
            aiy[i] = -100000; aix[i] = -100000; abase[i] = 0;
        }
    }
    const T* bsrc[BG];
#pragma unroll
    for (int i = 0; i < BG; ++i) {
        const int row = (wave * BG + i) * 8 + rsub;
        bsrc[i] = Wt + (long)(n0 + row) * K + (cpos ^ ((row >> 1) & 7)) * CE;
    }
    const T* zeros = static_cast<const T*>(p.zeros);

    auto issue = [&](int kt, int stage) {
        const int tap = kt / ctiles, c0 = (kt - tap * ctiles) * KE;
        const int ky = tap / p.KW, kx = tap - ky * p.KW;
        unsigned char* sb = lds + stage * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < AG; ++i) {
            const int iy = aiy[i] + ky, ix = aix[i] + kx;
            const bool ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            const T* src = ok ? X + abase[i] + ((long)iy * p.W + ix) * p.Cin + c0 + achunk[i] : zeros;
            if (kt >= nk1) src = abase2[i] >= 0 ? X2 + abase2[i] + (long)(kt - nk1) * KE : zeros;   // K-extension: second source
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sb + (wave * AG + i) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < BG; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(bsrc[i] + (long)kt * KE), (lptr_t)(sb + BM * ROWB + (wave * BG + i) * 1024), 16, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // residual tile -> registers before anything else (bf16: 16 B per item): its HBM/L2 latency hides under the
    // whole K loop instead of sitting in the epilogue.  These loads are older than every DMA, and loads return
    // in order, so the counted vmcnt waits below stay valid.
    constexpr bool PRE = (ES == 2);
    constexpr int ITEMS = (BM * (BN / 8)) / (NW * 64);
    uint4 rres[PRE ? ITEMS : 1];
    if constexpr (PRE) {
        if (p.res) {
            const T* R = static_cast<const T*>(p.res);
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const int item = it * (NW * 64) + tid;
                const int row = item / (BN / 8), cg = item - row * (BN / 8);
                const int m = min(m0 + row, M - 1);   // clamp instead of branching: a predicated load would be waited for at once
                if (p.nt & 2) { const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(R + (long)m * p.Cout + n0 + cg * 8)); rres[it] = make_uint4(t[0], t[1], t[2], t[3]); }
                else rres[it] = *reinterpret_cast<const uint4*>(R + (long)m * p.Cout + n0 + cg * 8);
            }
        }
    }

#pragma unroll
    for (int s = 0; s < S - 1; ++s)
        if (s < nk) issue(s, s);

    const int lrow = lane & 31, lhalf = lane >> 5;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    int stage = 0;            // ring slot of tile kt
    int fill = S - 1;         // ring slot the next issued tile goes to
    for (int kt = 0; kt < nk; ++kt) {
        // tiles kt+1 .. kt+ahead were issued after tile kt and may stay in flight
        const int ahead = min(S - 2, nk - 1 - kt);
        if (ahead >= 2) wait_vmcnt<2 * LPT>();
        else if (ahead == 1) wait_vmcnt<LPT>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();   // every wave's share of tile kt has landed; every wave is done with tile kt-1
        asm volatile("" ::: "memory");
        if (kt + S - 1 < nk && !ABL(p.debug, 4)) issue(kt + S - 1, fill);   // refill the slot tile kt-1 just vacated
        const unsigned a_base = lds_base + stage * STAGE_BYTES, b_base = a_base + BM * ROWB;
        if constexpr (X3) {
            ktile_mma_x3<TM, TN, true>(acc, a_base, b_base, wm, wn, lrow, lhalf);
            stage = (stage + 1 == S) ? 0 : stage + 1;
            fill = (fill + 1 == S) ? 0 : fill + 1;
            continue;
        }
        u32x4 fa[2][TM], fb[2][TN];
        auto frag_read = [&](int ks, int buf) {
            const int ch = 2 * ks + lhalf;
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[buf][i] = lds_read_b128(a_base + swz((wm * TM + i) * 32 + lrow, ch));
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[buf][j] = lds_read_b128(b_base + swz((wn * TN + j) * 32 + lrow, ch));
        };
        frag_read(0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            lds_wait();                                  // fragments of K-step ks are in registers
            if (ks < 3) frag_read(ks + 1, (ks + 1) & 1);  // next K-step's reads fly under this step's MFMAs
            if ABL(p.debug, 2) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const u32x4 av = fa[ks & 1][i], bv = fb[ks & 1][j];
                    if constexpr (ES == 2) {
                        union { u32x4 u; bf16x8 v; } ua, ub;
                        ua.u = av; ub.u = bv;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.v, ub.v, acc[i][j], 0, 0, 0);
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(av.x), __uint_as_float(bv.x), acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(av.y), __uint_as_float(bv.y), acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(av.z), __uint_as_float(bv.z), acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(av.w), __uint_as_float(bv.w), acc[i][j], 0, 0, 0);
                    }
                }
        }
        stage = (stage + 1 == S) ? 0 : stage + 1;
        fill = (fill + 1 == S) ? 0 : fill + 1;
    }
    __syncthreads();  // nothing in flight (last wait was vmcnt(0)); all waves done reading the ring

    // ---- epilogue: identical to the register-staged kernel
    float* Cs = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                const int col = (wn * TN + j) * 32 + lrow;
                Cs[row * BN + col] = acc[i][j][r];
            }
    __syncthreads();
    epilogue_store<T, BM, BN, NW * 64, PRE>(p, Cs, m0, n0, M, tid, rres);
}

// ---------------------------------------------------------------- wave-specialised variant (K-heavy layers)
// Same ring / swizzle / epilogue, but LW extra waves (one per SIMD) do nothing except issue the LDS-DMA and count
// it: an LDS-DMA wave-instruction occupies its wave's issue slot for ~60-180 cycles, which the in-order compute
// waves above pay in front of their MFMAs.  Here the NW compute waves only run {barrier, ds_read, MFMA}.
template <typename T, int WAVES_M, int WAVES_N, int TM, int TN, int S, int LW, bool X3 = false>
__global__ __launch_bounds__((WAVES_M * WAVES_N + LW) * 64) void conv_igemm_ws_kernel(ConvArgs p) {
    static_assert(!X3 || sizeof(T) == 4, "the three-pass split is a mode of the fp32 tensors");
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    constexpr int ES = (int)sizeof(T);
    constexpr int KE = ROWB / ES, CE = 16 / ES;
    constexpr int AG = BM / 8 / LW, BG = BN / 8 / LW;  // 8-row groups per loader wave per K-tile
    static_assert(AG * 8 * LW == BM && BG * 8 * LW == BN, "tile rows must split evenly over the loader waves");
    constexpr int LPT = AG + BG;
    constexpr int STAGE_BYTES = (BM + BN) * ROWB;
    constexpr int EPI_BYTES = BM * BN * 4;
    constexpr int LDS_BYTES = (S * STAGE_BYTES > EPI_BYTES) ? S * STAGE_BYTES : EPI_BYTES;
    static_assert(S >= 2 && S <= 4, "ring depth");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nbn = p.Cout / BN;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    // tile order: m-major (the n-tiles of one m-tile adjacent on one XCD: the A tile is fetched from HBM once) or
    // n-major (an XCD keeps ONE weight slice in its L2 and walks the m-tiles: for slices too big to share an L2)
    const int nbm = gridDim.x / nbn;
    const int tile_n = p.nmajor ? L / nbm : L % nbn, tile_m0 = p.nmajor ? L % nbm : L / nbn;
    const int tile_m = p.rev ? nbm - 1 - tile_m0 : tile_m0;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int M = p.B * p.Ho * p.Wo;
    const int K = p.KH * p.KW * p.Cin + (p.x2 ? p.Cin2 : 0);
    const int nk = K / KE;

    if (wave >= NW) {
        // ================= loader wave =================
        const int lw = wave - NW;
        const T* X = static_cast<const T*>(p.x);
        const T* X2 = static_cast<const T*>(p.x2);
        const T* Wt = static_cast<const T*>(p.w);
        const T* zeros = static_cast<const T*>(p.zeros);
        const int ctiles = p.Cin / KE;
        const int nk1 = p.KH * p.KW * p.Cin / KE;
        const int rsub = lane >> 3, cpos = lane & 7;
        long abase[AG], abase2[AG];
        int aiy[AG], aix[AG], achunk[AG];
#pragma unroll
        for (int i = 0; i < AG; ++i) {
            const int row = (lw * AG + i) * 8 + rsub;
            achunk[i] = (cpos ^ ((row >> 1) & 7)) * CE;
            const int m = m0 + row;
            abase2[i] = -1;
            if (m < M) {
                const int b = m / (p.Ho * p.Wo);
                const int rem = m - b * (p.Ho * p.Wo);
                const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                aiy[i] = oy * p.stride - p.pad;
                aix[i] = ox * p.stride - p.pad;
                abase[i] = (long)b * p.H * p.W * p.Cin;
                if (X2) abase2[i] = (((long)b * p.H2 + oy * p.stride2) * p.W2 + ox * p.stride2) * p.Cin2 + achunk[i];
            } else {
                aiy[i] = -100000; aix[i] = -100000; abase[i] = 0;
            }
        }
        const T* bsrc[BG];
#pragma unroll
        for (int i = 0; i < BG; ++i) {
            const int row = (lw * BG + i) * 8 + rsub;
            bsrc[i] = Wt + (long)(n0 + row) * K + (cpos ^ ((row >> 1) & 7)) * CE;
        }
        auto issue = [&](int kt, int stage) {
            const int tap = kt / ctiles, c0 = (kt - tap * ctiles) * KE;
            const int ky = tap / p.KW, kx = tap - ky * p.KW;
            unsigned char* sb = lds + stage * STAGE_BYTES;
#pragma unroll
            for (int i = 0; i < AG; ++i) {
                const int iy = aiy[i] + ky, ix = aix[i] + kx;
                const bool ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                const T* src = ok ? X + abase[i] + ((long)iy * p.W + ix) * p.Cin + c0 + achunk[i] : zeros;
                if (kt >= nk1) src = abase2[i] >= 0 ? X2 + abase2[i] + (long)(kt - nk1) * KE : zeros;   // K-extension: second source
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sb + (lw * AG + i) * 1024), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < BG; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t)(bsrc[i] + (long)kt * KE), (lptr_t)(sb + BM * ROWB + (lw * BG + i) * 1024), 16, 0, 0);
        };
#pragma unroll
        for (int s = 0; s < S - 1; ++s)
            if (s < nk) issue(s, s);
        int fill = S - 1;
        for (int kt = 0; kt < nk; ++kt) {
            const int ahead = min(S - 2, nk - 1 - kt);
            if (ahead >= 2) wait_vmcnt<2 * LPT>();
            else if (ahead == 1) wait_vmcnt<LPT>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();   // tile kt landed (all loaders); compute waves are done with tile kt-1
            if (kt + S - 1 < nk) issue(kt + S - 1, fill);
            fill = (fill + 1 == S) ? 0 : fill + 1;
        }
        __builtin_amdgcn_s_barrier();       // matches the two epilogue barriers of the compute waves
        __builtin_amdgcn_s_barrier();
        return;
    }

    // ================= compute wave =================
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    constexpr bool PRE = (ES == 2) && (NW + LW) <= 12;   // 16-wave variant: 128 VGPRs per wave, no room for the residual tile
    constexpr int ITEMS = (BM * (BN / 8)) / (NW * 64);
    uint4 rres[PRE ? ITEMS : 1];
    if constexpr (PRE) {
        if (p.res) {
            const T* R = static_cast<const T*>(p.res);
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const int item = it * (NW * 64) + tid;
                const int row = item / (BN / 8), cg = item - row * (BN / 8);
                const int m = min(m0 + row, M - 1);
                if (p.nt & 2) { const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(R + (long)m * p.Cout + n0 + cg * 8)); rres[it] = make_uint4(t[0], t[1], t[2], t[3]); }
                else rres[it] = *reinterpret_cast<const uint4*>(R + (long)m * p.Cout + n0 + cg * 8);
            }
        }
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int lrow = lane & 31, lhalf = lane >> 5;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if ABL(p.debug, 8) continue;                   // ablation: loaders + barriers only (pure fill rate of the real access pattern)
        const unsigned a_base = lds_base + stage * STAGE_BYTES, b_base = a_base + BM * ROWB;
        if constexpr (X3) {
            ktile_mma_x3<TM, TN, true>(acc, a_base, b_base, wm, wn, lrow, lhalf);
            stage = (stage + 1 == S) ? 0 : stage + 1;
            continue;
        }
        u32x4 fa[2][TM], fb[2][TN];
        auto frag_read = [&](int ks, int buf) {
            const int ch = 2 * ks + lhalf;
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[buf][i] = lds_read_b128(a_base + swz((wm * TM + i) * 32 + lrow, ch));
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[buf][j] = lds_read_b128(b_base + swz((wn * TN + j) * 32 + lrow, ch));
        };
        frag_read(0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            lds_wait();
            if (ks < 3) frag_read(ks + 1, (ks + 1) & 1);
            if ABL(p.debug, 2) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const u32x4 av = fa[ks & 1][i], bv = fb[ks & 1][j];
                    if constexpr (ES == 2) {
                        union { u32x4 u; bf16x8 v; } ua, ub;
                        ua.u = av; ub.u = bv;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.v, ub.v, acc[i][j], 0, 0, 0);
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(av.x), __uint_as_float(bv.x), acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(av.y), __uint_as_float(bv.y), acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(av.z), __uint_as_float(bv.z), acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(av.w), __uint_as_float(bv.w), acc[i][j], 0, 0, 0);
                    }
                }
        }
        stage = (stage + 1 == S) ? 0 : stage + 1;
    }
    __builtin_amdgcn_s_barrier();  // every compute wave is done reading the ring (loaders have nothing in flight)
    asm volatile("" ::: "memory");
    float* Cs = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                const int col = (wn * TN + j) * 32 + lrow;
                Cs[row * BN + col] = acc[i][j][r];
            }
    __syncthreads();               // second epilogue barrier (with the LDS-write drain)
    epilogue_store<T, BM, BN, NW * 64, PRE>(p, Cs, m0, n0, M, tid, rres);
}

// ---------------------------------------------------------------- 3x3 stride-1 convolution with an LDS-resident patch (bf16)
// The implicit-GEMM kernels above fetch the A tile once per filter tap: 9 x 32 KB per 64-channel slice of a 256-pixel
// tile, and the ~20 B/clk/CU fill path (L2 hits + DRAM misses retiring in order) is what bounds them.  Here the
// halo patch of the tile (<= 400 pixels x 64 channels, 41-50 KB) is fetched ONCE per channel slice and all nine taps
// read it through shifted fragment rows (output pixel m, tap (ky,kx) -> patch row hb(m) + ky*(TW+2) + kx); only the
// 16-KB weight tiles stream per tap.  Fill bytes per tap drop from 48 KB to ~21 KB.
//   tile = F frames x TH x TW output pixels (256), BN = 128 output channels, 8 compute + 8 loader waves
//   LDS: two patch buffers (channel slices alternate) | 3-slot weight ring | epilogue tile aliases everything
// Swizzle key of a patch row from its PATCH COORDINATES (round 5; it was (hr >> 1) & 7 of the linear row index: 49 % of this kernel's LDS
// cycles were bank conflicts).  A 16-lane ds_read_b128 group reads, at one tap, TW = 16: columns {x .. x+3, x+12 .. x+15} of one patch line and
// {x+4 .. x+11} of the next - 16 consecutive columns -> key = (hx >> 1) & 7; TW = 8: one 4-pixel half of each of four consecutive lines
// -> key = (hy & 3, (hx >> 1) & 1).  The line pitch TW + 2 (and the frame pitch) is even, so the row's 128-byte half is the column's parity
// and (hx & 1, key) names 16 different 16-byte slots either way.
template <int TW>
__device__ __forceinline__ int patch_key(int hy, int hx) {
    static_assert(TW == 16 || TW == 8, "tile widths of the patch kernel");
    return TW == 16 ? ((hx >> 1) & 7) : (((hy & 3) << 1) | ((hx >> 1) & 1));
}

template <int F, int TH, bool KEYXY = true>
__global__ __launch_bounds__(1024) void conv3x3_patch_kernel(ConvArgs p) {
    typedef bf16_t T;
    constexpr int TW = TH, HW2 = TW + 2, HP = (TH + 2) * HW2, HRT = F * HP;   // halo rows of the tile
    constexpr int NG = (HRT + 7) / 8;                                          // 1-KB DMA row groups per patch
    constexpr int PB = NG * 1024;
    constexpr int BM = F * TH * TW, TM = BM / 128;           // 256 pixels per tile (TM = 2), or 128 (F = 2 frames of 8 x 8: small launches, round 5)
    constexpr int NW = 8, LW = 8, TN = 2, BN = 128, KE = 64, CE = 8;
    static_assert(BM == 256 || BM == 128, "a tile is 8 or 4 pixel tiles of 32");
    constexpr int WSLOT = BN * ROWB, WR = 3;
    constexpr int W_OFF = 2 * PB;
    constexpr int EPI_BYTES = BM * BN * 4;
    constexpr int USED = W_OFF + WR * WSLOT;
    constexpr int LDS_BYTES = USED > EPI_BYTES ? USED : EPI_BYTES;
    static_assert(F * TH * TW == BM && LDS_BYTES <= 163840, "tile geometry / LDS budget");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nbn = p.Cout / BN;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    // n-major (p.nmajor): the workgroups of one XCD share ONE 128-channel weight slice (its L2 keeps it: the weight
    // tiles are ~85 % of the fill bytes here); m-major: the n-tiles of one pixel tile share its patch instead
    const int nbm = gridDim.x / nbn;
    const int tile_n = p.nmajor ? L / nbm : L % nbn, tile_m0 = p.nmajor ? L % nbm : L / nbn;
    const int tile_m = p.rev ? nbm - 1 - tile_m0 : tile_m0;
    const int n0 = tile_n * BN;
    const int tpf = (p.H / TH) * (p.W / TW);                 // tiles per frame (1 when a tile spans F whole frames)
    const int b0 = (F > 1) ? tile_m * F : tile_m / tpf;
    const int tl = (F > 1) ? 0 : tile_m % tpf;
    const int y0 = (tl / (p.W / TW)) * TH, x0 = (tl % (p.W / TW)) * TW;
    const int K = 9 * p.Cin;
    const int nc = p.Cin / KE;                               // channel slices
    const int nj = nc * 9;                                   // (slice, tap) steps

    if (wave >= NW) {
        // ================= loader wave =================
        const int lw = wave - NW;
        const T* X = static_cast<const T*>(p.x);
        const T* Wt = static_cast<const T*>(p.w);
        const T* zeros = static_cast<const T*>(p.zeros);
        const int rsub = lane >> 3, cpos = lane & 7;
        constexpr int MAXG = (NG + LW - 1) / LW;
        const int np = (NG - lw + LW - 1) / LW;              // patch row groups of this wave: lw, lw+8, ...
        const T* abase[MAXG];
        unsigned okmask = 0;
#pragma unroll
        for (int i = 0; i < MAXG; ++i) {
            const int g = lw + LW * i;
            const int hr = g * 8 + rsub;
            const int f = hr / HP, rem = hr - f * HP;
            const int hy = rem / HW2, hx = rem - hy * HW2;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const bool ok = g < NG && hr < HRT && y >= 0 && y < p.H && x >= 0 && x < p.W && b0 + f < p.B;   // (frames past the batch in the last 4-frame tile read zeros)
            abase[i] = ok ? X + (((long)(b0 + f) * p.H + y) * p.W + x) * p.Cin + (cpos ^ (KEYXY ? patch_key<TW>(hy, hx) : ((hr >> 1) & 7))) * CE : zeros;
            okmask |= ok ? (1u << i) : 0u;
        }
        const T* bsrc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (lw * 2 + i) * 8 + rsub;
            bsrc[i] = Wt + (long)(n0 + row) * K + (cpos ^ ((row >> 1) & 7)) * CE;
        }
        auto issue_patch = [&](int c) {
            unsigned char* pbuf = lds + (c & 1) * PB;
#pragma unroll
            for (int i = 0; i < MAXG; ++i) {
                const int g = lw + LW * i;
                if (g < NG) dma16(abase[i] + (((okmask >> i) & 1u) ? c * KE : 0), pbuf + g * 1024);
            }
        };
        auto issue_w = [&](int j) {                          // weight tile of step j = slice * 9 + tap
            const int c = j / 9, t = j - c * 9;
            unsigned char* sb = lds + W_OFF + (j % WR) * WSLOT;
#pragma unroll
            for (int i = 0; i < 2; ++i) dma16(bsrc[i] + (long)t * p.Cin + c * KE, sb + (lw * 2 + i) * 1024);
        };
        // issue order: P0 W0 W1 | after barrier j: W(j+2); after barrier (c,3): P(c+1)
        issue_patch(0);
        issue_w(0);
        if (nj > 1) issue_w(1);
        for (int j = 0; j < nj; ++j) {
            const int c = j / 9, t = j - c * 9;
            // younger than W(j) at this point: W(j+1), and P(c+1) during taps 4 and 5 (issued behind W(c*9+5))
            int younger = (j + 1 < nj) ? 2 : 0;
            if ((t == 4 || t == 5) && c + 1 < nc) younger += np;
            wait_vmcnt_n(younger);
            __builtin_amdgcn_s_barrier();                    // step j: its weights (and patch) landed; step j-1 is done
            if (j + 2 < nj) issue_w(j + 2);
            if (t == 3 && c + 1 < nc) issue_patch(c + 1);    // patch buffer (c+1)&1 was last read in slice c-1
        }
        __builtin_amdgcn_s_barrier();                        // matches the two epilogue barriers of the compute waves
        __builtin_amdgcn_s_barrier();
        return;
    }

    // ================= compute wave =================
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int lrow = lane & 31, lhalf = lane >> 5;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    int hb[TM], hy0[TM], hx0[TM];                            // patch row (and line / column) of this lane's output pixel at tap (0,0)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = (wm * TM + i) * 32 + lrow;
        const int f = m / (TH * TW), rem = m - f * (TH * TW);
        const int y = rem / TW, x = rem - y * TW;
        hb[i] = f * HP + y * HW2 + x;
        hy0[i] = y; hx0[i] = x;
    }
    int c = 0, t = 0;
    for (int j = 0; j < nj; ++j) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if ABL(p.debug, 8) continue;                           // ablation: loaders + barriers only
        const unsigned a_base = lds_base + (c & 1) * PB, b_base = lds_base + W_OFF + (j % WR) * WSLOT;
        const int ky = t / 3, kx = t - ky * 3;
        unsigned arow[TM], asw[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int hr = hb[i] + ky * HW2 + kx;
            arow[i] = a_base + hr * ROWB;
            asw[i] = KEYXY ? patch_key<TW>(hy0[i] + ky, hx0[i] + kx) : ((hr >> 1) & 7);
        }
        u32x4 fa[2][TM], fb[2][TN];
        auto frag_read = [&](int ks, int buf) {
            const int ch = 2 * ks + lhalf;
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[buf][i] = lds_read_b128(arow[i] + ((ch ^ asw[i]) << 4));
#pragma unroll
            for (int jj = 0; jj < TN; ++jj) fb[buf][jj] = lds_read_b128(b_base + swz((wn * TN + jj) * 32 + lrow, ch));
        };
        frag_read(0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            lds_wait();
            if (ks < 3) frag_read(ks + 1, (ks + 1) & 1);
            if ABL(p.debug, 2) {                               // ablation: fragment reads without the MFMAs
                asm volatile("" ::"v"(fa[ks & 1][0]), "v"(fa[ks & 1][TM - 1]), "v"(fb[ks & 1][0]), "v"(fb[ks & 1][TN - 1]));      // (TM = 1 in the 2-frame tiling)
                continue;
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jj = 0; jj < TN; ++jj) acc[i][jj] = mfma_bf16(fa[ks & 1][i], fb[ks & 1][jj], acc[i][jj]);
        }
        if (++t == 9) { t = 0; ++c; }
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    float* Cs = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                const int col = (wn * TN + j) * 32 + lrow;
                Cs[row * BN + col] = acc[i][j][r];
            }
    __syncthreads();
    // rows of the tile are (frame, y, x) inside the tile: map them back to NHWC pixel indices for the store pass
    ConvArgs q = p;
    q.res = nullptr;
    constexpr int CPR = BN / 8;
    T* Y = static_cast<T*>(p.y);
    const float4 bb0 = *reinterpret_cast<const float4*>(p.bias + n0 + (tid % CPR) * 8);
    const float4 bb1 = *reinterpret_cast<const float4*>(p.bias + n0 + (tid % CPR) * 8 + 4);
#pragma unroll
    for (int it = 0; it < (BM * CPR) / (NW * 64); ++it) {
        const int item = it * (NW * 64) + tid;
        const int row = item / CPR, cg = item - row * CPR;
        const int f = row / (TH * TW), rem = row - f * (TH * TW);
        const int y = rem / TW, x = rem - y * TW;
        if (F > 1 && b0 + f >= p.B) continue;                  // partial last tile (B % F != 0)
        const long o = ((((long)(b0 + f) * p.H + y0 + y) * p.W + x0 + x)) * p.Cout + n0 + cg * 8;
        const float4 v0 = *reinterpret_cast<const float4*>(Cs + row * BN + cg * 8);
        const float4 v1 = *reinterpret_cast<const float4*>(Cs + row * BN + cg * 8 + 4);
        float v[8] = {v0.x + bb0.x, v0.y + bb0.y, v0.z + bb0.z, v0.w + bb0.w, v1.x + bb1.x, v1.y + bb1.y, v1.z + bb1.z, v1.w + bb1.w};
        uint32_t pk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            pk[k] = act2_bf16(v[2 * k], v[2 * k + 1], p.relu != 0);
        }
        if (p.nt & 4) {                                       // bit 2 (off by default: measured slightly worse for this kernel)
            u32x4 ov = {pk[0], pk[1], pk[2], pk[3]};
            __builtin_nontemporal_store(ov, reinterpret_cast<u32x4*>(Y + o));
        } else {
            *reinterpret_cast<uint4*>(Y + o) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        }
    }
}

// ---------------------------------------------------------------- the same for the three-pass mode's 64-channel 3x3 (res2, IVOSW_F32X3; round 6)
// res2's 3x3 (64 -> 64 channels on 64 x 64 frames) was the largest row of that mode's layer table: 256 x 64 tiles of the per-tap kernel gather
// 9 taps x 2 slices x 32 KB of pixels per tile - 734 KB with the weights - and sit on the CU's fill rate at 185 "fp32-equivalent" TFLOP/s where the
// 128-channel 3x3 layers reach 314 - 335.  Here a workgroup owns a 16 x 16-pixel tile: its 18 x 18 halo patch is fetched once per 32-channel slice
// (128-byte rows in the split layout [32 hi | 32 lo], 41 KB per slice), the nine taps read it through shifted rows, only the 8-KB weight tiles
// stream per (slice, tap): 227 KB per tile.  Four compute waves own 2 pixel tiles x BOTH 32-channel tiles (16 KB of fragment reads per 24 MFMAs:
// 83 B/clk of LDS reads per CU; eight waves with one channel tile each would need 125), four loader waves issue the LDS-DMA.
// K order: slice-major (all nine taps of channels 0 .. 31, then 32 .. 63) - the per-tap kernel runs tap-major, so the two differ in the last
// bits of the fp32 sums (not bit-identical; the test compares them at 1e-6).  LDS: two patch buffers | 3-slot weight ring; the epilogue tile aliases.
__global__ __launch_bounds__(512) void conv3x3_patch_x3_kernel(ConvArgs p) {
    constexpr int TH = 16, TW = 16, HW2 = TW + 2, HRT = (TH + 2) * HW2;        // 324 halo rows
    constexpr int NG = (HRT + 7) / 8, PB = NG * 1024;                          // 41 DMA row groups, 41 984 B per slice
    constexpr int BM = 256, BN = 64, NW = 4, LW = 4, TM = 2, TN = 2, KE = 32, CE = 4;   // floats per K-tile / per 16-byte chunk
    constexpr int WSLOT = BN * ROWB, WR = 3, W_OFF = 2 * PB;
    constexpr int EPI_BYTES = BM * BN * 4, USED = W_OFF + WR * WSLOT;
    constexpr int LDS_BYTES = USED > EPI_BYTES ? USED : EPI_BYTES;
    static_assert(LDS_BYTES <= 163840, "LDS budget");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = p.rev ? (int)gridDim.x - 1 - L : L;
    const int tpx = p.W / TW, tpf = (p.H / TH) * tpx;
    const int b0 = tile_m / tpf, tl = tile_m - b0 * tpf;
    const int y0 = (tl / tpx) * TH, x0 = (tl % tpx) * TW;
    const int K = 9 * p.Cin;
    const int nc = p.Cin / KE, nj = nc * 9;

    if (wave >= NW) {
        // ================= loader wave =================
        const int lw = wave - NW;
        const float* X = static_cast<const float*>(p.x);
        const float* Wt = static_cast<const float*>(p.w);
        const float* zeros = static_cast<const float*>(p.zeros);
        const int rsub = lane >> 3, cpos = lane & 7;
        constexpr int MAXG = (NG + LW - 1) / LW;             // 11
        const int np = (NG - lw + LW - 1) / LW;
        const float* abase[MAXG];
        unsigned okmask = 0;
#pragma unroll
        for (int i = 0; i < MAXG; ++i) {
            const int g = lw + LW * i;
            const int hr = g * 8 + rsub;
            const int hy = hr / HW2, hx = hr - hy * HW2;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const bool ok = g < NG && hr < HRT && y >= 0 && y < p.H && x >= 0 && x < p.W;
            abase[i] = ok ? X + (((long)b0 * p.H + y) * p.W + x) * p.Cin + (cpos ^ patch_key<TW>(hy, hx)) * CE : zeros;
            okmask |= ok ? (1u << i) : 0u;
        }
        const float* bsrc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (lw * 2 + i) * 8 + rsub;
            bsrc[i] = Wt + (long)row * K + (cpos ^ ((row >> 1) & 7)) * CE;
        }
        auto issue_patch = [&](int c) {
            unsigned char* pbuf = lds + (c & 1) * PB;
#pragma unroll
            for (int i = 0; i < MAXG; ++i) {
                const int g = lw + LW * i;
                if (g < NG) dma16(abase[i] + (((okmask >> i) & 1u) ? c * KE : 0), pbuf + g * 1024);
            }
        };
        auto issue_w = [&](int j) {                          // weight tile of step j = slice * 9 + tap
            const int c = j / 9, t = j - c * 9;
            unsigned char* sb = lds + W_OFF + (j % WR) * WSLOT;
#pragma unroll
            for (int i = 0; i < 2; ++i) dma16(bsrc[i] + (long)t * p.Cin + c * KE, sb + (lw * 2 + i) * 1024);
        };
        // issue order: P0 W0 W1 | after barrier j: W(j+2); after barrier (c,3): P(c+1)        [as conv3x3_patch_kernel]
        issue_patch(0);
        issue_w(0);
        issue_w(1);
        for (int j = 0; j < nj; ++j) {
            const int c = j / 9, t = j - c * 9;
            int younger = (j + 1 < nj) ? 2 : 0;
            if ((t == 4 || t == 5) && c + 1 < nc) younger += np;
            wait_vmcnt_n(younger);
            __builtin_amdgcn_s_barrier();
            if (j + 2 < nj) issue_w(j + 2);
            if (t == 3 && c + 1 < nc) issue_patch(c + 1);
        }
        __builtin_amdgcn_s_barrier();                        // matches the two epilogue barriers of the compute waves
        __builtin_amdgcn_s_barrier();
        return;
    }

    // ================= compute wave: pixel tiles 2 wave, 2 wave + 1 (rows 4 wave .. 4 wave + 3 of the tile) x both channel tiles =================
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int lrow = lane & 31, lhalf = lane >> 5;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    int hb[TM], hy0[TM], hx0[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = (wave * TM + i) * 32 + lrow;
        const int y = m / TW, x = m - y * TW;
        hb[i] = y * HW2 + x;
        hy0[i] = y; hx0[i] = x;
    }
    int c = 0, t = 0;
    for (int j = 0; j < nj; ++j) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const unsigned a_base = lds_base + (c & 1) * PB, b_base = lds_base + W_OFF + (j % WR) * WSLOT;
        const int ky = t / 3, kx = t - ky * 3;
        unsigned arow[TM], asw[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            arow[i] = a_base + (hb[i] + ky * HW2 + kx) * ROWB;
            asw[i] = patch_key<TW>(hy0[i] + ky, hx0[i] + kx);
        }
        u32x4 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
        auto frag_read = [&](int s2, int buf) {              // ktile_mma_x3<PRE = true>'s chunks: hi at 2 s + h, lo at 4 + 2 s + h
            const int ch = 2 * s2 + lhalf;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[buf][i] = lds_read_b128(arow[i] + ((ch ^ asw[i]) << 4));
                al[buf][i] = lds_read_b128(arow[i] + (((4 + ch) ^ asw[i]) << 4));
            }
#pragma unroll
            for (int jj = 0; jj < TN; ++jj) {
                bh[buf][jj] = lds_read_b128(b_base + swz(jj * 32 + lrow, ch));
                bl[buf][jj] = lds_read_b128(b_base + swz(jj * 32 + lrow, 4 + ch));
            }
        };
        frag_read(0, 0);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            lds_wait();
            if (s2 < 1) frag_read(1, 1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jj = 0; jj < TN; ++jj) {
                    acc[i][jj] = mfma_bf16(ah[s2][i], bh[s2][jj], acc[i][jj]);
                    acc[i][jj] = mfma_bf16(ah[s2][i], bl[s2][jj], acc[i][jj]);
                    acc[i][jj] = mfma_bf16(al[s2][i], bh[s2][jj], acc[i][jj]);
                }
        }
        if (++t == 9) { t = 0; ++c; }
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    float* Cs = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (wave * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                const int col = j * 32 + lrow;
                Cs[row * BN + col] = acc[i][j][r];
            }
    __syncthreads();                                         // the LDS-write drain + the second epilogue barrier (the loader waves count it too)
    // store pass: thread = (tile pixel, 8 channels); + bias, ReLU, split (epilogue_store's IVOSW_F32X3 branch)
    constexpr int CPR = BN / 8;
    float* Y = static_cast<float*>(p.y);
    const float4 bb0 = *reinterpret_cast<const float4*>(p.bias + (tid % CPR) * 8);
    const float4 bb1 = *reinterpret_cast<const float4*>(p.bias + (tid % CPR) * 8 + 4);
#pragma unroll
    for (int it = 0; it < (BM * CPR) / (NW * 64); ++it) {
        const int item = it * (NW * 64) + tid;
        const int row = item / CPR, cg = item - row * CPR;
        const int y = row / TW, x = row - y * TW;
        const float4 v0 = *reinterpret_cast<const float4*>(Cs + row * BN + cg * 8);
        const float4 v1 = *reinterpret_cast<const float4*>(Cs + row * BN + cg * 8 + 4);
        float v[8] = {v0.x + bb0.x, v0.y + bb0.y, v0.z + bb0.z, v0.w + bb0.w, v1.x + bb1.x, v1.y + bb1.y, v1.z + bb1.z, v1.w + bb1.w};
        if (p.relu) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
        }
        uint32_t hi[4], lo[4];
        split8_store_x3(v, hi, lo);
        const int n = cg * 8;
        uint4* ys = reinterpret_cast<uint4*>(Y + ((((long)b0 * p.H + y0 + y) * p.W + x0 + x)) * p.Cout + (n & ~31));
        const int cq = (n & 31) >> 3;
        if (p.nt & 4) {
            const u32x4 vh = {hi[0], hi[1], hi[2], hi[3]}, vl = {lo[0], lo[1], lo[2], lo[3]};
            __builtin_nontemporal_store(vh, reinterpret_cast<u32x4*>(ys + cq));
            __builtin_nontemporal_store(vl, reinterpret_cast<u32x4*>(ys + 4 + cq));
        } else {
            ys[cq] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            ys[4 + cq] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        }
    }
}
// ---------------------------------------------------------------- res2's identity blocks of the three-pass mode behind their conv1: 3x3 -> conv3 + residual in ONE launch
// (round 6).  Layer by layer the 64-channel t2 (1 KB per pixel in the split format, written and read back) and a second launch's prologue sit
// between the patch kernel above and the 64 -> 256 expand layer, which is bound by its 4-byte tensors (t2 in, residual in, y out).  Here the
// patch kernel's accumulators become the expand layer's pixel operand without leaving the CU: t2 = relu(acc + b2) goes to LDS, is split IN
// PLACE (a 32-channel group of the split format is the 128 bytes of its 32 floats; the eight lanes that own a row read before any of them
// writes), and four 64-channel chunks of y = relu(W3 t2 + b3 + x) follow, their weight tiles (8 KB per chunk and 32-channel K-tile) through
// the same ring behind the eighteen 3x3 tiles.  Every row of t2 / of a y chunk belongs to ONE compute wave (pixel tiles 2 w, 2 w + 1) from
// the accumulators to the store pass, so phase 2 needs no barrier beyond the ring's.  Per pixel the products and their order are those of the
// two kernels it replaces (conv3x3_patch_x3_kernel, then the 128 x 128-tile expand layer: K-tile order, hi/lo order, bias after the sum,
// residual after the bias): bit-identical to them.
// LDS (152 KB): ring [0, 24 K) | patch buffers [24 K, 106 K) -> t2 [24 K, 88 K) | y-chunk staging [88 K, 152 K).
struct Res2TailX3Args {
    const float *t1, *x, *w2, *b2, *w3, *b3, *zeros;
    const float* p2;             // DS (res2's first block): the block input, second K half of [conv3 | downsample]; x (the residual) is unused there
    float* y;
    int B, H, W, nt, rev;
};
// DS: res2's FIRST block - no residual; conv3 is [conv3 | downsample] along K (K = 128: t2, then the block input p2 at the tile's pixels, whose
// fragments every compute wave takes straight from global memory into 64 registers once - the residual's registers are free there).
template <bool DS>
__global__ __launch_bounds__(512) void res2_tail_x3_kernel(Res2TailX3Args p) {
    constexpr int TH = 16, TW = 16, HW2 = TW + 2, HRT = (TH + 2) * HW2, NG = (HRT + 7) / 8, PB = NG * 1024;
    constexpr int NW = 4, LW = 4, TM = 2, TN = 2, KE = 32, CE = 4, C = 64, C4 = 256;
    constexpr int WSLOT = 64 * ROWB, WR = 3, P_OFF = WR * WSLOT, T2_OFF = P_OFF, YC_OFF = T2_OFF + 256 * 256;
    constexpr int LDS_BYTES = YC_OFF + 256 * 256;
    constexpr int NK2 = DS ? 4 : 2, K3 = NK2 * KE;           // K-tiles / K of the expand layer
    constexpr int NJ1 = 18, NJ = NJ1 + 4 * NK2;              // ring tiles: (slice, tap) of the 3x3, then (chunk, K-tile) of the expand layer
    static_assert(P_OFF + 2 * PB <= LDS_BYTES && LDS_BYTES <= 163840, "LDS map");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = p.rev ? (int)gridDim.x - 1 - L : L;
    const int tpx = p.W / TW, tpf = (p.H / TH) * tpx;
    const int b0 = tile_m / tpf, tl = tile_m - b0 * tpf;
    const int y0 = (tl / tpx) * TH, x0 = (tl % tpx) * TW;

    if (wave >= NW) {
        // ================= loader wave =================
        const int lw = wave - NW;
        const int rsub = lane >> 3, cpos = lane & 7;
        constexpr int MAXG = (NG + LW - 1) / LW;
        const int np = (NG - lw + LW - 1) / LW;
        const float* abase[MAXG];
        unsigned okmask = 0;
#pragma unroll
        for (int i = 0; i < MAXG; ++i) {
            const int g = lw + LW * i;
            const int hr = g * 8 + rsub;
            const int hy = hr / HW2, hx = hr - hy * HW2;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const bool ok = g < NG && hr < HRT && y >= 0 && y < p.H && x >= 0 && x < p.W;
            abase[i] = ok ? p.t1 + (((long)b0 * p.H + y) * p.W + x) * C + (cpos ^ patch_key<TW>(hy, hx)) * CE : p.zeros;
            okmask |= ok ? (1u << i) : 0u;
        }
        const float *bsrc2[2], *bsrc3[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (lw * 2 + i) * 8 + rsub;
            bsrc2[i] = p.w2 + (long)row * (9 * C) + (cpos ^ ((row >> 1) & 7)) * CE;
            bsrc3[i] = p.w3 + (long)row * K3 + (cpos ^ ((row >> 1) & 7)) * CE;
        }
        auto issue_patch = [&](int c) {
            unsigned char* pbuf = lds + P_OFF + (c & 1) * PB;
#pragma unroll
            for (int i = 0; i < MAXG; ++i) {
                const int g = lw + LW * i;
                if (g < NG) dma16(abase[i] + (((okmask >> i) & 1u) ? c * KE : 0), pbuf + g * 1024);
            }
        };
        auto issue_w = [&](int j) {
            unsigned char* sb = lds + (j % WR) * WSLOT;
            if (j < NJ1) {
                const int c = j / 9, t = j - c * 9;
#pragma unroll
                for (int i = 0; i < 2; ++i) dma16(bsrc2[i] + (long)t * C + c * KE, sb + (lw * 2 + i) * 1024);
            } else {
                const int q = j - NJ1, n = q / NK2, kt = q - n * NK2;
#pragma unroll
                for (int i = 0; i < 2; ++i) dma16(bsrc3[i] + (long)n * 64 * K3 + kt * KE, sb + (lw * 2 + i) * 1024);
            }
        };
        issue_patch(0);
        issue_w(0);
        issue_w(1);
        for (int j = 0; j < NJ; ++j) {
            const int c = j / 9, t = j - c * 9;
            int younger = (j + 1 < NJ) ? 2 : 0;
            if (j < NJ1 && (t == 4 || t == 5) && c == 0) younger += np;
            wait_vmcnt_n(younger);
            __builtin_amdgcn_s_barrier();
            if (j + 2 < NJ) issue_w(j + 2);
            if (j == 3) issue_patch(1);
            if (j == NJ1 - 1) __builtin_amdgcn_s_barrier();          // the compute waves' "patches are dead" barrier
        }
        return;
    }

    // ================= compute wave: pixel tiles 2 wave, 2 wave + 1 (64 rows) through both phases =================
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int lrow = lane & 31, lhalf = lane >> 5;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    int hb[TM], hy0[TM], hx0[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = (wave * TM + i) * 32 + lrow;
        const int y = m / TW, x = m - y * TW;
        hb[i] = y * HW2 + x;
        hy0[i] = y; hx0[i] = x;
    }
    auto mma3 = [&](const u32x4 (&ah)[TM], const u32x4 (&al)[TM], const u32x4 (&bh)[TN], const u32x4 (&bl)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jj = 0; jj < TN; ++jj) {
                acc[i][jj] = mfma_bf16(ah[i], bh[jj], acc[i][jj]);
                acc[i][jj] = mfma_bf16(ah[i], bl[jj], acc[i][jj]);
                acc[i][jj] = mfma_bf16(al[i], bh[jj], acc[i][jj]);
            }
    };
    // ---------------- phase 1: the 3x3 (conv3x3_patch_x3_kernel's loop)
    {
        int c = 0, t = 0;
        for (int j = 0; j < NJ1; ++j) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const unsigned a_base = lds_base + P_OFF + (c & 1) * PB, b_base = lds_base + (j % WR) * WSLOT;
            const int ky = t / 3, kx = t - ky * 3;
            unsigned arow[TM], asw[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                arow[i] = a_base + (hb[i] + ky * HW2 + kx) * ROWB;
                asw[i] = patch_key<TW>(hy0[i] + ky, hx0[i] + kx);
            }
            u32x4 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
            auto frag_read = [&](int s2, int buf) {
                const int ch = 2 * s2 + lhalf;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[buf][i] = lds_read_b128(arow[i] + ((ch ^ asw[i]) << 4));
                    al[buf][i] = lds_read_b128(arow[i] + (((4 + ch) ^ asw[i]) << 4));
                }
#pragma unroll
                for (int jj = 0; jj < TN; ++jj) {
                    bh[buf][jj] = lds_read_b128(b_base + swz(jj * 32 + lrow, ch));
                    bl[buf][jj] = lds_read_b128(b_base + swz(jj * 32 + lrow, 4 + ch));
                }
            };
            frag_read(0, 0);
            lds_wait();
            frag_read(1, 1);
            mma3(ah[0], al[0], bh[0], bl[0]);
            lds_wait();
            mma3(ah[1], al[1], bh[1], bl[1]);
            if (++t == 9) { t = 0; ++c; }
        }
    }
    __builtin_amdgcn_s_barrier();                            // every wave is done with the patches: t2 and the y-chunk staging take their place
    asm volatile("" ::: "memory");
    u32x4 pfh[DS ? TM : 1][2][2], pfl[DS ? TM : 1][2][2];    // DS: the block input's fragments [pixel tile][K-tile][step], requested under the t2 epilogue
    if constexpr (DS) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = (wave * TM + i) * 32 + lrow;
            const int yy = row / TW, xx = row - yy * TW;
            const u32x4* ps = reinterpret_cast<const u32x4*>(p.p2 + ((((long)b0 * p.H + y0 + yy) * p.W + x0 + xx)) * C);
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    pfh[i][k2][s2] = ps[k2 * 8 + 2 * s2 + lhalf];
                    pfl[i][k2][s2] = ps[k2 * 8 + 4 + 2 * s2 + lhalf];
                }
        }
    }
    // ---------------- t2 = relu(acc + b2) -> LDS (fp32, this wave's 64 rows), then split in place: row = 256 B = two [32 hi | 32 lo] groups,
    // 16-byte slot (8 g + chunk) ^ (row & 15)
    float* T2f = reinterpret_cast<float*>(lds + T2_OFF);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (wave * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                T2f[row * 64 + j * 32 + lrow] = acc[i][j][r];
            }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    {
        const int cg = lane & 7;
        const float4 bb0 = *reinterpret_cast<const float4*>(p.b2 + cg * 8), bb1 = *reinterpret_cast<const float4*>(p.b2 + cg * 8 + 4);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = wave * 64 + it * 8 + (lane >> 3);
            const float4 v0 = *reinterpret_cast<const float4*>(T2f + row * 64 + cg * 8), v1 = *reinterpret_cast<const float4*>(T2f + row * 64 + cg * 8 + 4);
            float v[8] = {v0.x + bb0.x, v0.y + bb0.y, v0.z + bb0.z, v0.w + bb0.w, v1.x + bb1.x, v1.y + bb1.y, v1.z + bb1.z, v1.w + bb1.w};
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
            uint32_t hi[4], lo[4];
            split8_store_x3(v, hi, lo);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the row's eight lanes have their 32 bytes before any of them writes
            const int g8 = (cg >> 2) * 8, cq = cg & 3, key = row & 15;
            unsigned char* rb = lds + T2_OFF + row * 256;
            *reinterpret_cast<uint4*>(rb + (((g8 + cq) ^ key) << 4)) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            *reinterpret_cast<uint4*>(rb + (((g8 + 4 + cq) ^ key) << 4)) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // ---------------- phase 2: y chunk n = relu(W3[64 n .. 64 n + 63] t2 + b3 + x), K = 64 = two K-tiles
    float* YCf = reinterpret_cast<float*>(lds + YC_OFF);
    unsigned trow[TM], tkey[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = (wave * TM + i) * 32 + lrow;
        trow[i] = lds_base + T2_OFF + row * 256;
        tkey[i] = row & 15;
    }
    const int cg = lane & 7;
    auto item_ofs = [&](int it, int nch, int ln = -1) {      // float offset of (this lane's row of item `it`, the 32-channel group of channel nch) in x / y
        const int row = wave * 64 + it * 8 + ((ln < 0 ? lane : ln) >> 3);
        const int yy = row / TW, xx = row - yy * TW;
        return ((((long)b0 * p.H + y0 + yy) * p.W + x0 + xx)) * C4 + (nch & ~31);
    };
#pragma unroll 1
    for (int n = 0; n < 4; ++n) {
        int lr2 = lrow, ln2 = lane;                         // opaque per chunk: keeps the chunk loop's addresses from being precomputed (and spilled) outside it
        asm volatile("" : "+v"(lr2), "+v"(ln2));
        // the chunk's residual: every item's (hi, lo) chunk pair is requested HERE, in front of the chunk's MFMAs - the store pass below is the only
        // consumer, and with one compute wave per SIMD nothing else would hide the round trip
        const int nch = n * 64 + cg * 8, cq = (nch & 31) >> 3;
        u32x4 rh[DS ? 1 : 8], rl[DS ? 1 : 8];
        if constexpr (!DS) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const u32x4* xs = reinterpret_cast<const u32x4*>(p.x + item_ofs(it, nch, ln2));
                rh[it] = xs[cq]; rl[it] = xs[4 + cq];
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < NK2; ++kt) {
            const int j = NJ1 + NK2 * n + kt;
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const unsigned b_base = lds_base + (j % WR) * WSLOT;
            if (DS && kt >= 2) {
                // the block input's K-tiles: the pixel fragments are in registers, only the weight fragments come from the ring
                u32x4 bh[2][TN], bl[2][TN];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int jj = 0; jj < TN; ++jj) {
                        bh[s2][jj] = lds_read_b128(b_base + swz(jj * 32 + lr2, 2 * s2 + lhalf));
                        bl[s2][jj] = lds_read_b128(b_base + swz(jj * 32 + lr2, 4 + 2 * s2 + lhalf));
                    }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    if (s2 == 0) { asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } else lds_wait();
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int jj = 0; jj < TN; ++jj) {
                            acc[i][jj] = mfma_bf16(pfh[DS ? i : 0][kt & 1][s2], bh[s2][jj], acc[i][jj]);
                            acc[i][jj] = mfma_bf16(pfh[DS ? i : 0][kt & 1][s2], bl[s2][jj], acc[i][jj]);
                            acc[i][jj] = mfma_bf16(pfl[DS ? i : 0][kt & 1][s2], bh[s2][jj], acc[i][jj]);
                        }
                }
                continue;
            }
            u32x4 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
            unsigned tk[TM];                             // (opaque per K-tile: sixteen precomputed fragment addresses per wave, hoisted out of the chunk loop, were
#pragma unroll
            for (int i = 0; i < TM; ++i) { tk[i] = tkey[i]; asm volatile("" : "+v"(tk[i])); }     //  what the DS form spilled - and reloaded in every chunk)
            auto frag_read = [&](int s2, int buf) {
                const int ch = 2 * s2 + lhalf;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[buf][i] = lds_read_b128(trow[i] + (((kt * 8 + ch) ^ tk[i]) << 4));
                    al[buf][i] = lds_read_b128(trow[i] + (((kt * 8 + 4 + ch) ^ tk[i]) << 4));
                }
#pragma unroll
                for (int jj = 0; jj < TN; ++jj) {
                    bh[buf][jj] = lds_read_b128(b_base + swz(jj * 32 + lr2, ch));
                    bl[buf][jj] = lds_read_b128(b_base + swz(jj * 32 + lr2, 4 + ch));
                }
            };
            frag_read(0, 0);
            lds_wait();
            frag_read(1, 1);
            mma3(ah[0], al[0], bh[0], bl[0]);
            lds_wait();
            mma3(ah[1], al[1], bh[1], bl[1]);
        }
        // the chunk's store pass, this wave's 64 rows: every item's residual chunks requested first
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (wave * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                    YCf[row * 64 + j * 32 + lrow] = acc[i][j][r];
                }
        const float4 bb0 = *reinterpret_cast<const float4*>(p.b3 + nch), bb1 = *reinterpret_cast<const float4*>(p.b3 + nch + 4);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = wave * 64 + it * 8 + (lane >> 3);
            const float4 v0 = *reinterpret_cast<const float4*>(YCf + row * 64 + cg * 8), v1 = *reinterpret_cast<const float4*>(YCf + row * 64 + cg * 8 + 4);
            float v[8] = {v0.x + bb0.x, v0.y + bb0.y, v0.z + bb0.z, v0.w + bb0.w, v1.x + bb1.x, v1.y + bb1.y, v1.z + bb1.z, v1.w + bb1.w};
            if constexpr (!DS) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[2 * q] += __uint_as_float(rh[it][q] << 16) + __uint_as_float(rl[it][q] << 16);
                    v[2 * q + 1] += __uint_as_float(rh[it][q] & 0xffff0000u) + __uint_as_float(rl[it][q] & 0xffff0000u);
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
            uint32_t hi[4], lo[4];
            split8_store_x3(v, hi, lo);
            u32x4* ys = reinterpret_cast<u32x4*>(p.y + item_ofs(it, nch, ln2));
            const u32x4 vh = {hi[0], hi[1], hi[2], hi[3]}, vl = {lo[0], lo[1], lo[2], lo[3]};
            if (p.nt & 4) { __builtin_nontemporal_store(vh, ys + cq); __builtin_nontemporal_store(vl, ys + 4 + cq); }
            else { ys[cq] = vh; ys[4 + cq] = vl; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // (the staging rows are free for the next chunk's accumulators)
    }
}

void launch_res2_tail_x3(const void* t1, const void* x, const void* w2, const float* b2, const void* w3, const float* b3, const void* zeros, void* y,
                         int B, int H, int W, int rev, hipStream_t st, const void* p2) {
    Res2TailX3Args a{};
    a.t1 = static_cast<const float*>(t1); a.x = static_cast<const float*>(x); a.w2 = static_cast<const float*>(w2); a.b2 = b2;
    a.w3 = static_cast<const float*>(w3); a.b3 = b3; a.zeros = static_cast<const float*>(zeros); a.y = static_cast<float*>(y);
    a.p2 = static_cast<const float*>(p2);
    a.B = B; a.H = H; a.W = W; a.rev = rev;
    a.nt = tune_get("NT", 3) | (tune_get("NT_X3", 1) << 2);
    ConvArgs d{};
    d.B = B; d.H = H; d.W = W; d.Ho = H; d.Wo = W; d.Cin = 64; d.Cout = 256; d.KH = -4; d.KW = -4; d.stride = 1; d.res = p2 ? nullptr : x;     // KH = -4: "3x3 + expand" row of the layer report
    void* tok = prof_begin(d, 4, st);
    if (p2) hipLaunchKernelGGL(res2_tail_x3_kernel<true>, dim3(B * (H / 16) * (W / 16)), dim3(512), 0, st, a);
    else hipLaunchKernelGGL(res2_tail_x3_kernel<false>, dim3(B * (H / 16) * (W / 16)), dim3(512), 0, st, a);
    prof_end(tok, st);
}

// ---------------------------------------------------------------- the same for res3's identity blocks (128 -> 128 3x3 on 32 x 32 frames, 128 -> 512 expand + residual)
// Tile = 8 x 16 pixels (t2 of a 256-pixel tile would be 128 KB): 10 x 18 halo patch per 32-channel slice (four slices, two buffers), 16-KB weight
// tiles (128 output channels x one K-tile) through a 3-slot ring - 36 (slice, tap) tiles of the 3x3, then 16 (128-channel double chunk, K-tile) tiles
// of conv3.  Compute wave (wm, wn) owns pixel tiles 2 wm, 2 wm + 1 x channel tiles 2 wn, 2 wn + 1 of whatever 128 channels are being produced, so a
// row of t2 is written by two waves: one more barrier between the in-place split and phase 2.  The store pass stages a wave's 64 rows x 32 channels at
// a time in a private 8-KB tile.  3x3 in slice-major K order (the per-tap kernel it replaces runs tap-major): equal to it up to fp32 summation noise
// and single flips of the split format's last bit, like conv3x3_patch_x3_kernel.  LDS (144 KB): ring [0, 48 K) | patches -> t2 [48 K, 112 K) | staging.
__global__ __launch_bounds__(512) void res3_tail_x3_kernel(Res2TailX3Args p) {
    constexpr int TH = 8, TW = 16, HW2 = TW + 2, HRT = (TH + 2) * HW2, NG = (HRT + 7) / 8, PB = NG * 1024;     // 180 halo rows, 23 groups
    constexpr int NW = 4, LW = 4, TM = 2, TN = 2, KE = 32, CE = 4, C = 128, C4 = 512, NC = C / KE;
    constexpr int WSLOT = 128 * ROWB, WR = 3, P_OFF = WR * WSLOT, T2_OFF = P_OFF, T2_ROW = C * 4, YC_OFF = T2_OFF + 128 * T2_ROW;
    constexpr int LDS_BYTES = YC_OFF + NW * 8192;
    constexpr int NJ1 = NC * 9, NJ = NJ1 + 4 * NC;
    static_assert(P_OFF + 2 * PB <= YC_OFF && LDS_BYTES <= 163840, "LDS map");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = p.rev ? (int)gridDim.x - 1 - L : L;
    const int tpx = p.W / TW, tpf = (p.H / TH) * tpx;
    const int b0 = tile_m / tpf, tl = tile_m - b0 * tpf;
    const int y0 = (tl / tpx) * TH, x0 = (tl % tpx) * TW;

    if (wave >= NW) {
        // ================= loader wave =================
        const int lw = wave - NW;
        const int rsub = lane >> 3, cpos = lane & 7;
        constexpr int MAXG = (NG + LW - 1) / LW;             // 6
        const int np = (NG - lw + LW - 1) / LW;
        const float* abase[MAXG];
        unsigned okmask = 0;
#pragma unroll
        for (int i = 0; i < MAXG; ++i) {
            const int g = lw + LW * i;
            const int hr = g * 8 + rsub;
            const int hy = hr / HW2, hx = hr - hy * HW2;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const bool ok = g < NG && hr < HRT && y >= 0 && y < p.H && x >= 0 && x < p.W;
            abase[i] = ok ? p.t1 + (((long)b0 * p.H + y) * p.W + x) * C + (cpos ^ patch_key<TW>(hy, hx)) * CE : p.zeros;
            okmask |= ok ? (1u << i) : 0u;
        }
        const float *bsrc2[4], *bsrc3[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (lw * 4 + i) * 8 + rsub;         // 0 .. 127
            bsrc2[i] = p.w2 + (long)row * (9 * C) + (cpos ^ ((row >> 1) & 7)) * CE;
            bsrc3[i] = p.w3 + (long)row * C + (cpos ^ ((row >> 1) & 7)) * CE;
        }
        auto issue_patch = [&](int c) {
            unsigned char* pbuf = lds + P_OFF + (c & 1) * PB;
#pragma unroll
            for (int i = 0; i < MAXG; ++i) {
                const int g = lw + LW * i;
                if (g < NG) dma16(abase[i] + (((okmask >> i) & 1u) ? c * KE : 0), pbuf + g * 1024);
            }
        };
        auto issue_w = [&](int j) {
            unsigned char* sb = lds + (j % WR) * WSLOT;
            if (j < NJ1) {
                const int c = j / 9, t = j - c * 9;
#pragma unroll
                for (int i = 0; i < 4; ++i) dma16(bsrc2[i] + (long)t * C + c * KE, sb + (lw * 4 + i) * 1024);
            } else {
                const int q = j - NJ1, n2 = q >> 2, kt = q & 3;
#pragma unroll
                for (int i = 0; i < 4; ++i) dma16(bsrc3[i] + (long)n2 * 128 * C + kt * KE, sb + (lw * 4 + i) * 1024);
            }
        };
        issue_patch(0);
        issue_w(0);
        issue_w(1);
        for (int j = 0; j < NJ; ++j) {
            const int c = j / 9, t = j - c * 9;
            int younger = (j + 1 < NJ) ? 4 : 0;
            if (j < NJ1 && (t == 4 || t == 5) && c + 1 < NC) younger += np;
            wait_vmcnt_n(younger);
            __builtin_amdgcn_s_barrier();
            if (j + 2 < NJ) issue_w(j + 2);
            if (j < NJ1 && t == 3 && c + 1 < NC) issue_patch(c + 1);
            if (j == NJ1 - 1) {                              // the compute waves' "patches are dead" and "t2 is complete" barriers
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_s_barrier();
            }
        }
        return;
    }

    // ================= compute wave (wm, wn): pixel tiles 2 wm, 2 wm + 1 x channel tiles 2 wn, 2 wn + 1 =================
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int lrow = lane & 31, lhalf = lane >> 5;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    int hb[TM], hy0[TM], hx0[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = (wm * TM + i) * 32 + lrow;
        const int y = m / TW, x = m - y * TW;
        hb[i] = y * HW2 + x;
        hy0[i] = y; hx0[i] = x;
    }
    auto mma3 = [&](const u32x4 (&ah)[TM], const u32x4 (&al)[TM], const u32x4 (&bh)[TN], const u32x4 (&bl)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jj = 0; jj < TN; ++jj) {
                acc[i][jj] = mfma_bf16(ah[i], bh[jj], acc[i][jj]);
                acc[i][jj] = mfma_bf16(ah[i], bl[jj], acc[i][jj]);
                acc[i][jj] = mfma_bf16(al[i], bh[jj], acc[i][jj]);
            }
    };
    {
        int c = 0, t = 0;
        for (int j = 0; j < NJ1; ++j) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const unsigned a_base = lds_base + P_OFF + (c & 1) * PB, b_base = lds_base + (j % WR) * WSLOT;
            const int ky = t / 3, kx = t - ky * 3;
            unsigned arow[TM], asw[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                arow[i] = a_base + (hb[i] + ky * HW2 + kx) * ROWB;
                asw[i] = patch_key<TW>(hy0[i] + ky, hx0[i] + kx);
            }
            u32x4 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
            auto frag_read = [&](int s2, int buf) {
                const int ch = 2 * s2 + lhalf;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[buf][i] = lds_read_b128(arow[i] + ((ch ^ asw[i]) << 4));
                    al[buf][i] = lds_read_b128(arow[i] + (((4 + ch) ^ asw[i]) << 4));
                }
#pragma unroll
                for (int jj = 0; jj < TN; ++jj) {
                    bh[buf][jj] = lds_read_b128(b_base + swz((wn * TN + jj) * 32 + lrow, ch));
                    bl[buf][jj] = lds_read_b128(b_base + swz((wn * TN + jj) * 32 + lrow, 4 + ch));
                }
            };
            frag_read(0, 0);
            lds_wait();
            frag_read(1, 1);
            mma3(ah[0], al[0], bh[0], bl[0]);
            lds_wait();
            mma3(ah[1], al[1], bh[1], bl[1]);
            if (++t == 9) { t = 0; ++c; }
        }
    }
    __builtin_amdgcn_s_barrier();                            // every wave is done with the patches
    asm volatile("" ::: "memory");
    // t2 = relu(acc + b2): this wave's 64 rows x 64 channels as fp32, then split in place; row = 512 B = four [32 hi | 32 lo] groups, slot (8 g + chunk) ^ (row & 15)
    float* T2f = reinterpret_cast<float*>(lds + T2_OFF);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                T2f[row * C + wn * 64 + j * 32 + lrow] = acc[i][j][r];
            }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    {
        const int cg = lane & 7;
        const float4 bb0 = *reinterpret_cast<const float4*>(p.b2 + wn * 64 + cg * 8), bb1 = *reinterpret_cast<const float4*>(p.b2 + wn * 64 + cg * 8 + 4);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = wm * 64 + it * 8 + (lane >> 3);
            const float4 v0 = *reinterpret_cast<const float4*>(T2f + row * C + wn * 64 + cg * 8), v1 = *reinterpret_cast<const float4*>(T2f + row * C + wn * 64 + cg * 8 + 4);
            float v[8] = {v0.x + bb0.x, v0.y + bb0.y, v0.z + bb0.z, v0.w + bb0.w, v1.x + bb1.x, v1.y + bb1.y, v1.z + bb1.z, v1.w + bb1.w};
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
            uint32_t hi[4], lo[4];
            split8_store_x3(v, hi, lo);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the eight lanes of a (row, 64-channel half) have their 32 bytes before any of them writes
            const int g8 = (wn * 2 + (cg >> 2)) * 8, cq = cg & 3, key = row & 15;
            unsigned char* rb = lds + T2_OFF + row * T2_ROW;
            *reinterpret_cast<uint4*>(rb + (((g8 + cq) ^ key) << 4)) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            *reinterpret_cast<uint4*>(rb + (((g8 + 4 + cq) ^ key) << 4)) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                            // t2 complete: a row's two 64-channel halves come from two waves
    asm volatile("" ::: "memory");
    // ---------------- phase 2: y[128 n2 + 64 wn .. + 63] = relu(W3 t2 + b3 + x), K = 128 = four K-tiles per 128-channel double chunk
    float* YCf = reinterpret_cast<float*>(lds + YC_OFF + wave * 8192);       // private staging: 64 rows x 32 channels
    unsigned trow[TM], tkey[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = (wm * TM + i) * 32 + lrow;
        trow[i] = lds_base + T2_OFF + row * T2_ROW;
        tkey[i] = row & 15;
    }
    const int cgl = lane & 3;
    auto item_ofs = [&](int it, int nch) {
        const int row = wm * 64 + it * 16 + (lane >> 2);
        const int yy = row / TW, xx = row - yy * TW;
        return ((((long)b0 * p.H + y0 + yy) * p.W + x0 + xx)) * C4 + (nch & ~31);
    };
    for (int n2 = 0; n2 < 4; ++n2) {
        u32x4 rh[1][4], rl[1][4];                            // the first 32-channel half's residual, requested in front of the double chunk's MFMAs (the second
        {                                                    // half's goes out under the first half's store pass: both halves up front spilled 24 registers)
            const int nch = n2 * 128 + wn * 64 + cgl * 8;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const u32x4* xs = reinterpret_cast<const u32x4*>(p.x + item_ofs(it, nch));
                rh[0][it] = xs[cgl]; rl[0][it] = xs[4 + cgl];
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < NC; ++kt) {
            const int j = NJ1 + NC * n2 + kt;
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const unsigned b_base = lds_base + (j % WR) * WSLOT;
            u32x4 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
            auto frag_read = [&](int s2, int buf) {
                const int ch = 2 * s2 + lhalf;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[buf][i] = lds_read_b128(trow[i] + (((kt * 8 + ch) ^ tkey[i]) << 4));
                    al[buf][i] = lds_read_b128(trow[i] + (((kt * 8 + 4 + ch) ^ tkey[i]) << 4));
                }
#pragma unroll
                for (int jj = 0; jj < TN; ++jj) {
                    bh[buf][jj] = lds_read_b128(b_base + swz((wn * TN + jj) * 32 + lrow, ch));
                    bl[buf][jj] = lds_read_b128(b_base + swz((wn * TN + jj) * 32 + lrow, 4 + ch));
                }
            };
            frag_read(0, 0);
            lds_wait();
            frag_read(1, 1);
            mma3(ah[0], al[0], bh[0], bl[0]);
            lds_wait();
            mma3(ah[1], al[1], bh[1], bl[1]);
        }
#pragma unroll
        for (int jj = 0; jj < TN; ++jj) {                    // store pass, 32 channels at a time through the private staging tile
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) YCf[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf) * 32 + lrow] = acc[i][jj][r];
            const int nch = n2 * 128 + wn * 64 + jj * 32 + cgl * 8;
            u32x4 qh[4], ql[4];                              // this half's residual; the next half's is requested before this half is processed
#pragma unroll
            for (int it = 0; it < 4; ++it) { qh[it] = rh[0][it]; ql[it] = rl[0][it]; }
            if (jj + 1 < TN) {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const u32x4* xs = reinterpret_cast<const u32x4*>(p.x + item_ofs(it, nch + 32));
                    rh[0][it] = xs[cgl]; rl[0][it] = xs[4 + cgl];
                }
            }
            const float4 bb0 = *reinterpret_cast<const float4*>(p.b3 + nch), bb1 = *reinterpret_cast<const float4*>(p.b3 + nch + 4);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int rl_ = it * 16 + (lane >> 2);
                const float4 v0 = *reinterpret_cast<const float4*>(YCf + rl_ * 32 + cgl * 8), v1 = *reinterpret_cast<const float4*>(YCf + rl_ * 32 + cgl * 8 + 4);
                float v[8] = {v0.x + bb0.x, v0.y + bb0.y, v0.z + bb0.z, v0.w + bb0.w, v1.x + bb1.x, v1.y + bb1.y, v1.z + bb1.z, v1.w + bb1.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[2 * q] += __uint_as_float(qh[it][q] << 16) + __uint_as_float(ql[it][q] << 16);
                    v[2 * q + 1] += __uint_as_float(qh[it][q] & 0xffff0000u) + __uint_as_float(ql[it][q] & 0xffff0000u);
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
                uint32_t hi[4], lo[4];
                split8_store_x3(v, hi, lo);
                u32x4* ys = reinterpret_cast<u32x4*>(p.y + item_ofs(it, nch));
                const u32x4 vh = {hi[0], hi[1], hi[2], hi[3]}, vl = {lo[0], lo[1], lo[2], lo[3]};
                if (p.nt & 4) { __builtin_nontemporal_store(vh, ys + cgl); __builtin_nontemporal_store(vl, ys + 4 + cgl); }
                else { ys[cgl] = vh; ys[4 + cgl] = vl; }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
}

void launch_res3_tail_x3(const void* t1, const void* x, const void* w2, const float* b2, const void* w3, const float* b3, const void* zeros, void* y,
                         int B, int H, int W, int rev, hipStream_t st) {
    Res2TailX3Args a{};
    a.t1 = static_cast<const float*>(t1); a.x = static_cast<const float*>(x); a.w2 = static_cast<const float*>(w2); a.b2 = b2;
    a.w3 = static_cast<const float*>(w3); a.b3 = b3; a.zeros = static_cast<const float*>(zeros); a.y = static_cast<float*>(y);
    a.B = B; a.H = H; a.W = W; a.rev = rev;
    a.nt = tune_get("NT", 3) | (tune_get("NT_X3", 1) << 2);
    ConvArgs d{};
    d.B = B; d.H = H; d.W = W; d.Ho = H; d.Wo = W; d.Cin = 128; d.Cout = 512; d.KH = -4; d.KW = -4; d.stride = 1; d.res = x;
    void* tok = prof_begin(d, 4, st);
    hipLaunchKernelGGL(res3_tail_x3_kernel, dim3(B * (H / 8) * (W / 16)), dim3(512), 0, st, a);
    prof_end(tok, st);
}

static bool patch3x3_x3_ok(const ConvArgs& a) {              // res2's 3x3 in the split activation format; the shape only, never the batch
    return a.x3 == 2 && a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && !a.res && !a.x2 && a.Cin == 64 && a.Cout == 64 &&
           a.H == a.W && a.H % 16 == 0 && a.Ho == a.H && a.Wo == a.W && a.zeros;
}

// shapes covered by conv3x3_patch_kernel
static bool patch3x3_ok(const ConvArgs& a) {
    if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.res || a.H != a.W || a.Cin % 64 || a.Cout % 128) return false;
    // ELIGIBILITY looks at the layer shape only, never at the batch (a frame's result must not depend on the batch it travels in:
    // B % 4 != 0 at H == 8 used to fall back to the per-tap kernel for the WHOLE launch, i.e. another summation order).  The TILING inside
    // the patch kernel may depend on B (round 5: 2-frame tiles for small launches) because the 2- and 4-frame tilings share the K order.
    return a.H == 32 || a.H == 16 || a.H == 8;
}

template <typename T, bool STEM, bool X3 = false>
static void launch_conv_t(const ConvArgs& a, hipStream_t st) {
    const int M = a.B * a.Ho * a.Wo;
    if constexpr (STEM) {
        const int grid = ((M + 127) / 128) * (a.Cout / 64);
        hipLaunchKernelGGL((conv_igemm_kernel<T, 2, 2, 2, 1, true, X3>), dim3(grid), dim3(256), 0, st, a);
    } else {
        // Tile / ring selection.  8 waves (2 per SIMD) so one wave's LDS-DMA issue overlaps the other's MFMAs;
        // 256-row tiles halve the DMA instructions per MFMA.  Short K loops (1x1 convs on 64/128 channels) use a
        // 2-deep ring.
        const int nk = (a.KH * a.KW * a.Cin + (a.x2 ? a.Cin2 : 0)) / (128 / (int)sizeof(T));
        if (a.Cout % 128 == 0) {
            const int grid = ((M + 255) / 256) * (a.Cout / 128);
            static const int tune_nk = getenv("IVOSW_TUNE_NK") ? atoi(getenv("IVOSW_TUNE_NK")) : 8;
            static const int use_ws = getenv("IVOSW_TUNE_WS") ? atoi(getenv("IVOSW_TUNE_WS")) : 1;
            if constexpr (sizeof(T) == 2) {
                if (patch3x3_ok(a) && tune_get("PATCH3", 1)) {
                    const int g3 = ((M + 255) / 256) * (a.Cout / 128);      // H == 8: four frames per tile, the last one may be partial
                    // PATCH_KEYXY = 1: swizzle key from the patch coordinates - PMC: bank-conflict share of this kernel 0.49 -> 0.013, LDS cycles
                    // per launch halved - and yet 74.7 against 72.4 us per res5 launch on alternating runs of one box (profiles/r05_lds_conflicts.txt):
                    // the kernel is bound by its MFMA issue at a power-limited clock, not by LDS time, so the row-index key stays the default
                    const bool kxy = tune_get("PATCH_KEYXY", 0) != 0;
                    // evaluation-size launches (round 5): fewer 4-frame tiles than PATCH_SMALL workgroups (res5 at < 128 frames per launch:
                    // 13 x 4 = 52 workgroups for 50 frames) -> 2-frame tiles, twice the workgroups; same (slice, tap) K order per output
                    // element, so a frame's result does not depend on which of the two ran (kernel trace at 100 units: this layer took
                    // 4.4 x its per-frame time at batch 256)
                    if (a.H == 8 && !kxy && g3 < tune_get("PATCH_SMALL", 128)) {
                        const int g2 = ((M + 127) / 128) * (a.Cout / 128);
                        hipLaunchKernelGGL((conv3x3_patch_kernel<2, 8, false>), dim3(g2), dim3(1024), 0, st, a);
                        return;
                    }
                    if (a.H == 8 && kxy) hipLaunchKernelGGL((conv3x3_patch_kernel<4, 8>), dim3(g3), dim3(1024), 0, st, a);
                    else if (a.H == 8) hipLaunchKernelGGL((conv3x3_patch_kernel<4, 8, false>), dim3(g3), dim3(1024), 0, st, a);
                    else if (kxy) hipLaunchKernelGGL((conv3x3_patch_kernel<1, 16>), dim3(g3), dim3(1024), 0, st, a);
                    else hipLaunchKernelGGL((conv3x3_patch_kernel<1, 16, false>), dim3(g3), dim3(1024), 0, st, a);
                    return;
                }
            }
            const int lw = tune_get("LW", 8);
            // small launches (an evaluation-size batch: fewer 256 x 128 tiles than SMALL_GRID workgroups): 128 x 128 tiles, two
            // workgroups per CU - twice the workgroups; same K order per output element, so a frame's result does not depend on it
            if (nk > tune_nk && grid < tune_get("SMALL_GRID", 128)) {
                const int g2 = ((M + 127) / 128) * (a.Cout / 128);
                hipLaunchKernelGGL((conv_igemm_dma_kernel<T, 4, 2, 1, 2, 3, X3>), dim3(g2), dim3(512), 0, st, a);
                return;
            }

            if (nk > tune_nk && use_ws && lw == 8) hipLaunchKernelGGL((conv_igemm_ws_kernel<T, 4, 2, 2, 2, 3, 8, X3>), dim3(grid), dim3(1024), 0, st, a);
            else if (nk > tune_nk && use_ws) hipLaunchKernelGGL((conv_igemm_ws_kernel<T, 4, 2, 2, 2, 3, 4, X3>), dim3(grid), dim3(768), 0, st, a);
            else if (nk > tune_nk) hipLaunchKernelGGL((conv_igemm_dma_kernel<T, 4, 2, 2, 2, 3, X3>), dim3(grid), dim3(512), 0, st, a);
            else {  // K <= 128: bound by the output/residual stream -> 128x128 tiles, 64 KB LDS, 2 workgroups per CU
                const int g2 = ((M + 127) / 128) * (a.Cout / 128);
                hipLaunchKernelGGL((conv_igemm_dma_kernel<T, 4, 2, 1, 2, 2, X3>), dim3(g2), dim3(512), 0, st, a);
            }
        } else {  // Cout == 64 layers: 256 x 64 tile
            if constexpr (X3) {
                if (patch3x3_x3_ok(a) && tune_get("PATCH3_X3", 1)) {
                    hipLaunchKernelGGL(conv3x3_patch_x3_kernel, dim3(a.B * (a.H / 16) * (a.W / 16)), dim3(512), 0, st, a);
                    return;
                }
            }
            const int grid = ((M + 255) / 256) * (a.Cout / 64);
            static const int use_ws64 = getenv("IVOSW_TUNE_WS") ? atoi(getenv("IVOSW_TUNE_WS")) : 1;
            if (nk >= 4 && use_ws64) hipLaunchKernelGGL((conv_igemm_ws_kernel<T, 4, 2, 2, 1, 3, 4, X3>), dim3(grid), dim3(768), 0, st, a);
            else if (nk >= 2) hipLaunchKernelGGL((conv_igemm_dma_kernel<T, 4, 2, 2, 1, 3, X3>), dim3(grid), dim3(512), 0, st, a);
            else hipLaunchKernelGGL((conv_igemm_dma_kernel<T, 4, 2, 2, 1, 2, X3>), dim3(grid), dim3(512), 0, st, a);
        }
    }
}

// measurement hook (include/ivosw.h: ivosw_profile_start/stop): hipEvent pairs around every conv launch
struct ConvProfiler {
    bool on = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    std::vector<ConvArgs> args;
    std::vector<int> es;
    size_t used = 0;
    // span mode (ivosw_profile_span_*): ONE event pair around each uninterrupted run of tower launches (stem .. last res5
    // kernel of a pass), so the family time contains its own launch gaps but no per-launch event overhead, and
    // family time <= wall time of the step holds by construction
    // With the two-stream split of a batch (assess.hip) two spans run concurrently, one per stream: spans of one forward
    // pass form a GROUP whose time is latest end - earliest start (the wall interval during which tower kernels ran).
    struct SpanRec { hipEvent_t a, b; int group; };
    bool span_on = false, span_open_[2] = {false, false};
    size_t span_idx[2] = {0, 0};
    std::vector<SpanRec> sev;
    size_t sused = 0;
    int group_next = 0, group_forced = -1;
    long span_launches = 0;
};
static ConvProfiler g_prof;

void span_group_begin() {
    if (g_prof.span_on) g_prof.group_forced = g_prof.group_next++;
}
void span_group_end() { g_prof.group_forced = -1; }
void span_open(hipStream_t st, int slot) {
    if (!g_prof.span_on || g_prof.span_open_[slot]) return;
    if (g_prof.sused == g_prof.sev.size()) {
        hipEvent_t x, y;
        (void)hipEventCreate(&x);
        (void)hipEventCreate(&y);
        g_prof.sev.push_back({x, y, 0});
    }
    g_prof.sev[g_prof.sused].group = g_prof.group_forced >= 0 ? g_prof.group_forced : g_prof.group_next++;
    (void)hipEventRecord(g_prof.sev[g_prof.sused].a, st);
    g_prof.span_idx[slot] = g_prof.sused++;
    g_prof.span_open_[slot] = true;
}
void span_close(hipStream_t st, int slot) {
    if (!g_prof.span_on || !g_prof.span_open_[slot]) return;
    (void)hipEventRecord(g_prof.sev[g_prof.span_idx[slot]].b, st);
    g_prof.span_open_[slot] = false;
}

// begin/end of one profiled launch: `a` describes the layer for the report (KH == 0 marks a fused bottleneck:
// Cin -> Cout/4 -> Cout/4 (3x3) -> Cout + residual)
void* prof_begin(const ConvArgs& a, int es, hipStream_t st) {
    if (g_prof.span_on && (g_prof.span_open_[0] || g_prof.span_open_[1])) ++g_prof.span_launches;
    if (!g_prof.on) return nullptr;
    if (g_prof.used == g_prof.ev.size()) {
        hipEvent_t x, y;
        (void)hipEventCreate(&x);
        (void)hipEventCreate(&y);
        g_prof.ev.emplace_back(x, y);
    }
    hipEvent_t e0 = g_prof.ev[g_prof.used].first, e1 = g_prof.ev[g_prof.used].second;
    if (g_prof.args.size() <= g_prof.used) { g_prof.args.resize(g_prof.used + 1); g_prof.es.resize(g_prof.used + 1); }
    g_prof.args[g_prof.used] = a;
    g_prof.es[g_prof.used] = es;
    ++g_prof.used;
    (void)hipEventRecord(e0, st);
    return e1;
}
void prof_end(void* tok, hipStream_t st) {
    if (tok) (void)hipEventRecord(static_cast<hipEvent_t>(tok), st);
}

void launch_conv(const ConvArgs& a_in, int dtype, bool stem, hipStream_t st) {
    static const int dbg = getenv("IVOSW_DEBUG_CONV") ? atoi(getenv("IVOSW_DEBUG_CONV")) : 0;
    ConvArgs a = a_in;
    a.debug = dbg;
    {
        const int nm = tune_get("NMAJOR", 0);       // 0 off, 1 all ws / patch launches, 2 only 3x3
        a.nmajor = (nm == 1) || (nm == 2 && a.KH == 3);
        a.nt = tune_get("NT", 3);   // streaming tensors are far larger than L2: keep them from evicting the A / weight lines that ARE reused
        // three-pass mode: bit 2 / 3 = non-temporal stores / residual loads of the split activations (NT_X3 = 1 / 2 / 3).  Two alternating rounds
        // of the whole mode on one box: stores + 1.5 % (20.97 -> 21.28 k frames/s), residual loads - 0.3 %, both + 1.1 %: stores only
        if (dtype == IVOSW_F32X3) a.nt |= tune_get("NT_X3", 1) << 2;
    }
    a.x3 = dtype == IVOSW_F32X3 ? 2 : 0;         // 2: activations in the split layout (every output; every input except the stem's ROI tile)
    void* tok = prof_begin(a, (dtype == IVOSW_BF16) ? 2 : 4, st);
    if (dtype == IVOSW_BF16) {
        if (stem) launch_conv_t<bf16_t, true>(a, st); else launch_conv_t<bf16_t, false>(a, st);
    } else if (a.x3) {
        if (stem) launch_conv_t<float, true, true>(a, st); else launch_conv_t<float, false, true>(a, st);
    } else {
        if (stem) launch_conv_t<float, true>(a, st); else launch_conv_t<float, false>(a, st);
    }
    prof_end(tok, st);
}

// [rows][K] fp32, in place: K-tile t of a row (floats 32 t .. 32 t + 31, 128 bytes) -> 32 x bf16 hi | 32 x bf16 lo
__global__ void split_weights_x3_kernel(float* __restrict__ w, long ntiles) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    float* base = w + t * 32;
    float x[32];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float4 v = reinterpret_cast<const float4*>(base)[q];
        x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
    }
    uint32_t hi[16], lo[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        hi[q] = pack2_bf16(x[2 * q], x[2 * q + 1]);
        const float h0 = __uint_as_float(hi[q] << 16), h1 = __uint_as_float(hi[q] & 0xffff0000u);
        lo[q] = pack2_bf16(x[2 * q] - h0, x[2 * q + 1] - h1);
    }
    uint4* o = reinterpret_cast<uint4*>(base);       // every input of the tile has been read: in place is safe
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = make_uint4(hi[4 * q], hi[4 * q + 1], hi[4 * q + 2], hi[4 * q + 3]);
#pragma unroll
    for (int q = 0; q < 4; ++q) o[4 + q] = make_uint4(lo[4 * q], lo[4 * q + 1], lo[4 * q + 2], lo[4 * q + 3]);
}
void launch_split_weights_x3(void* w, long rows, int K, hipStream_t st) {
    const long ntiles = rows * (K / 32);
    hipLaunchKernelGGL(split_weights_x3_kernel, dim3((unsigned)((ntiles + 255) / 256)), dim3(256), 0, st, static_cast<float*>(w), ntiles);
}

// ---------------------------------------------------------------- weight packing (BN fold + K-major repack)
// w [Cout,Cin,KH,KW] fp32 -> out[Cout][(ky*KW+kx)*Cin + ci] * gamma/sqrt(var+eps); bias = beta - mean*scale
template <typename T>
__global__ void pack_conv_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ beta,
                                 const float* __restrict__ rmean, const float* __restrict__ rvar, float eps, int Cout, int Cin,
                                 int KH, int KW, T* __restrict__ ow, float* __restrict__ ob) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long K = (long)KH * KW * Cin;
    if (i >= (long)Cout * K) return;
    const int co = (int)(i / K);
    const int k = (int)(i - (long)co * K);
    const int tap = k / Cin, ci = k - tap * Cin;
    const int ky = tap / KW, kx = tap - ky * KW;
    const float scale = gamma[co] / sqrtf(rvar[co] + eps);
    ow[i] = Elem<T>::from_f32(w[(((long)co * Cin + ci) * KH + ky) * KW + kx] * scale);
    if (k == 0) ob[co] = beta[co] - rmean[co] * scale;
}

// stem: conv1 [64,3,7,7] | conv1_p [64,1,7,7] -> out[64][KY][8 px][4 ch], zero for px == 7 / ky == 7
template <typename T>
__global__ void pack_stem_kernel(const float* __restrict__ w3, const float* __restrict__ w1, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, const float* __restrict__ rmean, const float* __restrict__ rvar,
                                 float eps, int KY, T* __restrict__ ow, float* __restrict__ ob) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int K = KY * 32;
    if (i >= 64 * K) return;
    const int co = i / K, k = i - co * K;
    const int ky = k >> 5, px = (k >> 2) & 7, ch = k & 3;
    const float scale = gamma[co] / sqrtf(rvar[co] + eps);
    float v = 0.f;
    if (ky < 7 && px < 7) v = (ch < 3) ? w3[((co * 3 + ch) * 7 + ky) * 7 + px] : w1[(co * 7 + ky) * 7 + px];
    ow[i] = Elem<T>::from_f32(v * scale);
    if (k == 0) ob[co] = beta[co] - rmean[co] * scale;
}

void launch_pack_conv(const float* w, const float* g, const float* b, const float* rm, const float* rv, int Cout, int Cin,
                      int KH, int KW, int dtype, void* ow, float* ob, hipStream_t st) {
    const long n = (long)Cout * Cin * KH * KW;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (dtype == IVOSW_BF16)
        hipLaunchKernelGGL(pack_conv_kernel<bf16_t>, grid, dim3(256), 0, st, w, g, b, rm, rv, 1e-5f, Cout, Cin, KH, KW, static_cast<bf16_t*>(ow), ob);
    else
        hipLaunchKernelGGL(pack_conv_kernel<float>, grid, dim3(256), 0, st, w, g, b, rm, rv, 1e-5f, Cout, Cin, KH, KW, static_cast<float*>(ow), ob);
}

template <typename T>
__global__ void concat_k_kernel(const T* __restrict__ w1, const float* __restrict__ b1, int K1, const T* __restrict__ w2,
                                const float* __restrict__ b2, int K2, int Cout, T* __restrict__ ow, float* __restrict__ ob) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int K = K1 + K2;
    if (i >= (long)Cout * K) return;
    const int co = (int)(i / K), k = (int)(i - (long)co * K);
    ow[i] = k < K1 ? w1[(long)co * K1 + k] : w2[(long)co * K2 + (k - K1)];
    if (k == 0) ob[co] = b1[co] + b2[co];
}

void launch_concat_k(const void* w1, const float* b1, int K1, const void* w2, const float* b2, int K2, int Cout, int dtype,
                     void* ow, float* ob, hipStream_t st) {
    const long n = (long)Cout * (K1 + K2);
    const dim3 grid((unsigned)((n + 255) / 256));
    if (dtype == IVOSW_BF16)
        hipLaunchKernelGGL(concat_k_kernel<bf16_t>, grid, dim3(256), 0, st, static_cast<const bf16_t*>(w1), b1, K1, static_cast<const bf16_t*>(w2), b2, K2, Cout, static_cast<bf16_t*>(ow), ob);
    else
        hipLaunchKernelGGL(concat_k_kernel<float>, grid, dim3(256), 0, st, static_cast<const float*>(w1), b1, K1, static_cast<const float*>(w2), b2, K2, Cout, static_cast<float*>(ow), ob);
}

void launch_pack_stem(const float* w3, const float* w1, const float* g, const float* b, const float* rm, const float* rv,
                      int dtype, void* ow, float* ob, hipStream_t st) {
    const int KY = (dtype == IVOSW_BF16) ? 8 : 7;
    const dim3 grid((64 * KY * 32 + 255) / 256);
    if (dtype == IVOSW_BF16)
        hipLaunchKernelGGL(pack_stem_kernel<bf16_t>, grid, dim3(256), 0, st, w3, w1, g, b, rm, rv, 1e-5f, KY, static_cast<bf16_t*>(ow), ob);
    else
        hipLaunchKernelGGL(pack_stem_kernel<float>, grid, dim3(256), 0, st, w3, w1, g, b, rm, rv, 1e-5f, KY, static_cast<float*>(ow), ob);
}

// ---------------------------------------------------------------- 3x3 stride-2 pad-1 max pool, NHWC (C = 64)
template <typename T>
__global__ void maxpool_kernel(const T* __restrict__ x, int B, int H, int W, int C, T* __restrict__ y) {
    constexpr int CE = 16 / (int)sizeof(T);
    const int Ho = H / 2, Wo = W / 2, CG = C / CE;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * Ho * Wo * CG) return;
    const int cg = (int)(i % CG);
    long t = i / CG;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    float m[CE];
#pragma unroll
    for (int q = 0; q < CE; ++q) m[q] = -INFINITY;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int iy = oy * 2 - 1 + dy;
        if (iy < 0 || iy >= H) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int ix = ox * 2 - 1 + dx;
            if (ix < 0 || ix >= W) continue;
            const uint4 v = *reinterpret_cast<const uint4*>(x + (((long)b * H + iy) * W + ix) * C + cg * CE);
            const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    m[2 * q] = fmaxf(m[2 * q], bf16_to_f32((bf16_t)(w4[q] & 0xffff)));
                    m[2 * q + 1] = fmaxf(m[2 * q + 1], bf16_to_f32((bf16_t)(w4[q] >> 16)));
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) m[q] = fmaxf(m[q], __uint_as_float(w4[q]));
            }
        }
    }
    T* o = y + (((long)b * Ho + oy) * Wo + ox) * C + cg * CE;
    if constexpr (sizeof(T) == 2) {
        uint32_t pk[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) pk[q] = pack2_bf16(m[2 * q], m[2 * q + 1]);
        *reinterpret_cast<uint4*>(o) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    } else {
        *reinterpret_cast<float4*>(o) = make_float4(m[0], m[1], m[2], m[3]);
    }
}

// the same on the split activation layout of IVOSW_F32X3 ([32 x bf16 hi | 32 x bf16 lo] per 32-channel group): a thread takes the 8
// channels of one 16-byte chunk, compares the reconstructed values (hi + lo is exact in fp32) and re-splits the maxima
__global__ void maxpool_x3_kernel(const uint4* __restrict__ x, int B, int H, int W, int C, uint4* __restrict__ y) {
    const int Ho = H / 2, Wo = W / 2, CG = C / 8, G8 = C / 32 * 8;      // uint4 chunks per pixel: 8 per 32-channel group
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * Ho * Wo * CG) return;
    const int cg = (int)(i % CG);
    long t = i / CG;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const int ch = (cg >> 2) * 8 + (cg & 3);                            // hi chunk of this thread's channels; lo = + 4
    float m[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) m[q] = -INFINITY;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int iy = oy * 2 - 1 + dy;
        if (iy < 0 || iy >= H) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int ix = ox * 2 - 1 + dx;
            if (ix < 0 || ix >= W) continue;
            const uint4* px = x + (((long)b * H + iy) * W + ix) * G8;
            const uint4 vh = px[ch], vl = px[ch + 4];
            const uint32_t h4[4] = {vh.x, vh.y, vh.z, vh.w}, l4[4] = {vl.x, vl.y, vl.z, vl.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                m[2 * q] = fmaxf(m[2 * q], __uint_as_float(h4[q] << 16) + __uint_as_float(l4[q] << 16));
                m[2 * q + 1] = fmaxf(m[2 * q + 1], __uint_as_float(h4[q] & 0xffff0000u) + __uint_as_float(l4[q] & 0xffff0000u));
            }
        }
    }
    uint32_t hi[4], lo[4];
    split8_store_x3(m, hi, lo);
    uint4* o = y + (((long)b * Ho + oy) * Wo + ox) * G8;
    o[ch] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    o[ch + 4] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}

// split layout -> plain fp32, in place (the test taps of the IVOSW_F32X3 mode): one thread per 32-channel group
__global__ void unsplit_x3_kernel(uint4* __restrict__ buf, long ngroups) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    uint4* p = buf + g * 8;
    uint4 v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = p[c];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint32_t h4[4] = {v[c].x, v[c].y, v[c].z, v[c].w}, l4[4] = {v[4 + c].x, v[4 + c].y, v[4 + c].z, v[4 + c].w};
        float f[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f[2 * q] = __uint_as_float(h4[q] << 16) + __uint_as_float(l4[q] << 16);
            f[2 * q + 1] = __uint_as_float(h4[q] & 0xffff0000u) + __uint_as_float(l4[q] & 0xffff0000u);
        }
        float4* o = reinterpret_cast<float4*>(p) + 2 * c;
        o[0] = make_float4(f[0], f[1], f[2], f[3]);
        o[1] = make_float4(f[4], f[5], f[6], f[7]);
    }
}

void launch_unsplit_x3(void* buf, size_t nfloats, hipStream_t st) {
    const long ng = (long)(nfloats / 32);
    hipLaunchKernelGGL(unsplit_x3_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, st, static_cast<uint4*>(buf), ng);
}

void launch_maxpool(const void* x, int B, int H, int W, int C, int dtype, void* y, hipStream_t st) {
    if (dtype == IVOSW_F32X3) {
        const long n = (long)B * (H / 2) * (W / 2) * (C / 8);
        hipLaunchKernelGGL(maxpool_x3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, static_cast<const uint4*>(x), B, H, W, C, static_cast<uint4*>(y));
    } else if (dtype == IVOSW_BF16) {
        const long n = (long)B * (H / 2) * (W / 2) * (C / 8);
        hipLaunchKernelGGL(maxpool_kernel<bf16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, static_cast<const bf16_t*>(x), B, H, W, C, static_cast<bf16_t*>(y));
    } else {
        const long n = (long)B * (H / 2) * (W / 2) * (C / 4);
        hipLaunchKernelGGL(maxpool_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, static_cast<const float*>(x), B, H, W, C, static_cast<float*>(y));
    }
}

// ---------------------------------------------------------------- 8x8 average pool + fc (2048 -> 1); one block per frame
template <typename T, bool SPLIT = false>
__global__ __launch_bounds__(256) void pool_fc_kernel(const T* __restrict__ x, const float* __restrict__ fcw, const float* __restrict__ fcb,
                                                      float* __restrict__ score, float* __restrict__ pooled_out) {
    constexpr int C = 2048, P = 64;
    const int b = blockIdx.x, tid = threadIdx.x;
    float s[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) s[q] = 0.f;
    const T* base = x + (long)b * P * C + tid * 8;
    for (int pidx = 0; pidx < P; ++pidx) {
        if constexpr (SPLIT) {
            // channels tid * 8 .. + 7 = chunk tid & 3 of 32-channel group tid >> 2: hi at chunk c, lo at chunk 4 + c of the group's 8
            const uint4* g = reinterpret_cast<const uint4*>(x + (long)b * P * C + (long)pidx * C) + (tid >> 2) * 8;
            const uint4 vh = g[tid & 3], vl = g[4 + (tid & 3)];
            const uint32_t h4[4] = {vh.x, vh.y, vh.z, vh.w}, l4[4] = {vl.x, vl.y, vl.z, vl.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                s[2 * q] += __uint_as_float(h4[q] << 16) + __uint_as_float(l4[q] << 16);
                s[2 * q + 1] += __uint_as_float(h4[q] & 0xffff0000u) + __uint_as_float(l4[q] & 0xffff0000u);
            }
        } else if constexpr (sizeof(T) == 2) {
            const uint4 v = *reinterpret_cast<const uint4*>(base + (long)pidx * C);
            const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                s[2 * q] += bf16_to_f32((bf16_t)(w4[q] & 0xffff));
                s[2 * q + 1] += bf16_to_f32((bf16_t)(w4[q] >> 16));
            }
        } else {
            const float4 v0 = *reinterpret_cast<const float4*>(base + (long)pidx * C);
            const float4 v1 = *reinterpret_cast<const float4*>(base + (long)pidx * C + 4);
            s[0] += v0.x; s[1] += v0.y; s[2] += v0.z; s[3] += v0.w;
            s[4] += v1.x; s[5] += v1.y; s[6] += v1.z; s[7] += v1.w;
        }
    }
    float dot = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float pv = s[q] * (1.0f / 64.0f);
        if (pooled_out) pooled_out[(long)b * C + tid * 8 + q] = pv;
        dot = fmaf(pv, fcw[tid * 8 + q], dot);
    }
    dot = wave_sum(dot);
    __shared__ float red[4];
    if ((tid & 63) == 0) red[tid >> 6] = dot;
    __syncthreads();
    if (tid == 0) score[b] = (red[0] + red[1]) + (red[2] + red[3]) + fcb[0];
}

void launch_pool_fc(const void* x, int B, int dtype, const float* fcw, const float* fcb, float* score, float* pooled,
                    hipStream_t st) {
    if (dtype == IVOSW_BF16)
        hipLaunchKernelGGL(pool_fc_kernel<bf16_t>, dim3(B), dim3(256), 0, st, static_cast<const bf16_t*>(x), fcw, fcb, score, pooled);
    else if (dtype == IVOSW_F32X3)
        hipLaunchKernelGGL((pool_fc_kernel<float, true>), dim3(B), dim3(256), 0, st, static_cast<const float*>(x), fcw, fcb, score, pooled);
    else
        hipLaunchKernelGGL(pool_fc_kernel<float>, dim3(B), dim3(256), 0, st, static_cast<const float*>(x), fcw, fcb, score, pooled);
}

}  // namespace ivosw

extern "C" int ivosw_profile_span_start(void) {
    using namespace ivosw;
    g_prof.span_on = true;
    g_prof.span_open_[0] = g_prof.span_open_[1] = false;
    g_prof.sused = 0;
    g_prof.group_next = 0;
    g_prof.group_forced = -1;
    g_prof.span_launches = 0;
    return IVOSW_OK;
}

extern "C" int ivosw_profile_span_stop(double* total_ms, int* spans, int* launches) {
    using namespace ivosw;
    IVOSW_REQUIRE(total_ms && spans && launches, "null pointer");
    double tot = 0.0;
    for (size_t i = 0; i < g_prof.sused; ++i) (void)hipEventSynchronize(g_prof.sev[i].b);
    for (size_t i = 0; i < g_prof.sused;) {      // the spans of a group are consecutive
        size_t j = i;
        while (j < g_prof.sused && g_prof.sev[j].group == g_prof.sev[i].group) ++j;
        // latest end - earliest start of the group's spans = the largest (end_y - start_x)
        float best = 0.f;
        for (size_t x = i; x < j; ++x)
            for (size_t y = i; y < j; ++y) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, g_prof.sev[x].a, g_prof.sev[y].b) == hipSuccess) best = std::max(best, ms);
            }
        tot += best;
        i = j;
    }
    (void)hipGetLastError();
    *total_ms = tot;
    *spans = g_prof.group_next;
    *launches = (int)g_prof.span_launches;
    g_prof.span_on = false;
    g_prof.span_open_[0] = g_prof.span_open_[1] = false;
    g_prof.sused = 0;
    return IVOSW_OK;
}

extern "C" int ivosw_profile_start(void) {
    ivosw::g_prof.on = true;
    ivosw::g_prof.used = 0;
    return IVOSW_OK;
}

// per-layer-shape table of the launches recorded since ivosw_profile_start (call before ivosw_profile_stop)
extern "C" int ivosw_profile_report(char* buf, size_t cap) {
    using namespace ivosw;
    IVOSW_REQUIRE(buf && cap > 0, "null buffer");
    struct Row { ConvArgs a; int es; int n; double ms; };
    std::vector<Row> rows;
    for (size_t i = 0; i < g_prof.used; ++i) {
        (void)hipEventSynchronize(g_prof.ev[i].second);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, g_prof.ev[i].first, g_prof.ev[i].second);
        const ConvArgs& a = g_prof.args[i];
        bool found = false;
        for (Row& r : rows)
            if (r.a.B == a.B && r.a.H == a.H && r.a.Cin == a.Cin && r.a.Cout == a.Cout && r.a.KH == a.KH && r.a.stride == a.stride &&
                (r.a.res != nullptr) == (a.res != nullptr)) { r.n++; r.ms += ms; found = true; break; }
        if (!found) rows.push_back(Row{a, g_prof.es[i], 1, ms});
    }
    size_t off = 0;
    off += snprintf(buf + off, cap - off, "%5s %4s %5s %5s %2s %2s %3s %5s %9s %9s %8s %8s\n", "B", "H", "Cin", "Cout", "K", "s", "res", "calls",
                    "avg_us", "total_ms", "TFLOP/s", "GB/s");
    for (const Row& r : rows) {
        if (off + 160 > cap) break;
        const ConvArgs& a = r.a;
        const double M = (double)a.B * a.Ho * a.Wo;
        const double cm = a.Cout / 4.0;
        // K = -2: the whole of res2 + res3's forwarded conv1 in one launch (res2_stage.hip): 1 006 632 960 algorithmic MACs per frame
        // (b0 301 989 888, b1 / b2 285 212 672 each, forwarded 1x1 134 217 728); bytes = pooled stem output in, even-pixel y2 + t1' out
        // K = -3: res3's first block behind its conv1 in one launch (stage_first.hip): 3x3 stride 2 + [conv3 | downsample]
        const double flops = a.KH == -2 ? 2.0 * a.B * 1006632960.0
                           : a.KH == -3 ? 2.0 * M * (9.0 * a.Cin * a.Cin + (double)a.Cout * (a.Cin + a.Cin2))
                           : a.KH ? 2.0 * M * a.Cout * (a.KH * a.KW * a.Cin + (a.x2 ? a.Cin2 : 0)) : 2.0 * M * (a.Cin * cm + 9.0 * cm * cm + cm * a.Cout);
        const double bytes = a.KH == -2 ? (double)a.B * (64.0 * 64 * 64 + 32.0 * 32 * 256 + 64.0 * 64 * 128) * r.es
                           : a.KH == -3 ? ((double)a.B * a.H * a.W * a.Cin + M * (a.Cin2 + a.Cout)) * r.es
                           : a.KH ? ((double)a.B * a.H * a.W * a.Cin + M * a.Cout * (a.res ? 2 : 1) + (double)a.Cout * a.KH * a.KW * a.Cin) * r.es
                                  : (M * (a.Cin + a.Cout) + a.Cin * cm + 13.0 * cm * cm) * r.es;
        const double t = r.ms / r.n * 1e-3;
        off += snprintf(buf + off, cap - off, "%5d %4d %5d %5d %2d %2d %3d %5d %9.2f %9.3f %8.1f %8.1f\n", a.B, a.H, a.Cin, a.Cout, a.KH, a.stride,
                        a.res ? 1 : 0, r.n, r.ms / r.n * 1e3, r.ms, flops / t / 1e12, bytes / t / 1e9);
    }
    return IVOSW_OK;
}

extern "C" int ivosw_profile_stop(double* total_ms, int* launches) {
    using namespace ivosw;
    IVOSW_REQUIRE(total_ms && launches, "null pointer");
    double tot = 0.0;
    for (size_t i = 0; i < g_prof.used; ++i) {
        (void)hipEventSynchronize(g_prof.ev[i].second);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_prof.ev[i].first, g_prof.ev[i].second) == hipSuccess) tot += ms;
    }
    *total_ms = tot;
    *launches = (int)g_prof.used;
    g_prof.on = false;
    g_prof.used = 0;
    return IVOSW_OK;
}
