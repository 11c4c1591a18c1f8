// res2 (ResNet-50 layer1: three bottlenecks at 64 x 64, 64 -> 256 channels) as ONE launch, bf16.
//
// Reference arithmetic: torchvision Bottleneck x 3 as Encoder.forward runs it (models/assessment.py:58, `self.res2`), BN folded:
//     b0: y0 = relu(W3 relu(W2 * relu(W1 p)) + Wd p)     b1, b2: y = relu(W3 relu(W2 * relu(W1 x)) + x)
// plus res3's first 1x1 (models/assessment.py:59) applied to y2 while it is on chip ("conv1 forwarding", as the per-block
// kernels of bottleneck_wide.hip do).
//
// Why: block by block, res2 is bound by the bytes a block cannot avoid — per 8 x 16-pixel tile 103 + 167 + 135 KB cross the
// fabric for ~13 k cycles of MFMA issue, and the three launches sit at ~11.5 B/clk/CU (0.18 of the bf16 peak,
// 12.8 MB of HBM traffic per frame against 2.5 MB for a stage-fused res2).  Here a workgroup owns an 8 x 16 OUTPUT tile through
// all three blocks: it reads the pooled stem output on a 14 x 22 halo (39 KB), recomputes the shrinking halo rings
// (b0 on 12 x 20, b1 on 10 x 18, b2 on 8 x 16: 1.51 x the MACs) and writes only what leaves the stage — y2 (at the even pixels:
// its only reader is res3's stride-2 downsample) and res3's t1: 48 KB.  y0 / y1 never exist in HBM: the next block's conv1 reads
// them from an LDS image, and the residual of the next block stays in the registers of the wave that produced it.
//
// Pixel order.  1x1 convolutions do not care in which order pixels are laid out, so every block enumerates its pixels in ONE
// slot order: [ 8x16 core (128) | ring of the 10x18 region (52) | 12 unused | ring of the 12x20 region (60) | 4 unused ].
// b0 works on slots [0, 256), b1 on [0, 192), b2 on [0, 128): the wave that owns (channel tile, pixel tile) in block k owns the
// same pair in block k + 1, which is what keeps the residual in registers.  Only the 3x3 inputs (t1) are stored as rasters
// (padded 144-byte rows: tap addresses become immediates, consecutive rows 36 banks apart).
//
// Dataflow per block (8 waves; weights in MFMA-fragment order stream from L2 straight into the registers of the one or two
// waves that need them; pixel operands come from LDS):
//   A  (b0 only)  t1 = relu(W1 p) on the 14x22 raster                         wave = (channel tile, pixel tiles g, g+4, g+8)
//   B             t2 = relu(W2 * t1): 36 k-steps split in two K halves         wave = (channel tile, pixel-tile parity, K half);
//                 the halves exchange partial sums through LDS, each finishes half of the pair's tiles
//   C             y = relu(W3 t2 [+ Wd p] [+ residual from registers])         wave = (channel tiles c, c+4; pixel-tile parity)
//                 -> bf16 -> LDS image (operand of D) + registers (residual of the next block)
//   D             t1' = relu(W1' y), zero outside the frame (the 3x3's padding) -> raster image / (b2) HBM
// b0's C / D run as two K halves of y0 (128 channels x 256 slots = 64 KB image each) with D's accumulators carried across.
// Summation orders equal those of the per-block kernels (bottleneck_wide.hip), so the stage output is bit-identical to theirs.
//
// LDS (163 840 B, one workgroup per CU):
//   A0/B0: P [0, 39936)  T1A [39936, 84288)  partial sums [84288, 149824)  -> T2A [30720, 63488), PD [0, 30720) (p again, slot order, by LDS-DMA)
//   C0/D0: PD, T2A, Y [63488, 129024), T1B [129024, 163584)
//   B1:    T1B, partial sums [24576, 73728) -> T2B [0, 24576)        C1/D1: T2B, Y [24576, 122880), T1C [122880, 148800)
//   B2:    T1C, partial sums [39936, 72704) -> T2C [72704, 89088)    C2/D2: T2C, Y [89088, 154624); [0, 39936): the next tile's P
#include <algorithm>
#include <type_traits>

#include "conv.h"
#include "mfma_tile.h"

namespace ivosw {

namespace {
#ifndef R2_DEPTH0                                    // k-steps of lead of the 3x3's pixel-fragment reads per block: 2 / 1 / 2 measured the same as
#define R2_DEPTH0 1                                  // 1 / 1 / 1 (B0 11 437 vs 11 387 cycles, B2 5 398 vs 5 788): the loops do not wait for LDS latency
#define R2_DEPTH1 1
#define R2_DEPTH2 1
#endif
constexpr int R2_LDS = 163840;
constexpr int T1R = 144;                             // bytes per padded t1 raster row
constexpr int P_OFF = 0, T1A_OFF = 39936, SB0_OFF = 84288, PD_OFF = 0, T2A_OFF = 30720, Y0_OFF = 63488, T1B_OFF = 129024;
constexpr int T2B_OFF = 0, SB1_OFF = 24576, Y1_OFF = 24576, T1C_OFF = 122880;
// (round 4's R2_GROUPS experiment - block 2 as two 4-wave groups behind LDS-counter group barriers: correct, 0.7 - 2.1 % slower, LAB_NOTES
// round 4 item 1 - was removed from this file in round 6: history, commit "two 4-wave groups behind LDS-counter group barriers")
constexpr int SB2_OFF = 39936, T2C_OFF = 72704, Y2_OFF = 89088;      // [0, 39936) stays free from B2 on: the NEXT tile's p halo lands there
static_assert(T1A_OFF + 308 * T1R <= SB0_OFF && SB0_OFF + 65536 <= R2_LDS, "A0/B0 map");
static_assert(PD_OFF + 240 * ROWB <= T2A_OFF && T2A_OFF + 256 * ROWB <= Y0_OFF && Y0_OFF + 65536 <= T1B_OFF && T1B_OFF + 240 * T1R <= R2_LDS, "C0/D0 map");
static_assert(T2B_OFF + 192 * ROWB <= SB1_OFF && SB1_OFF + 49152 <= T1B_OFF && Y1_OFF + 98304 <= T1C_OFF && T1C_OFF + 180 * T1R <= R2_LDS, "b1 map");
static_assert( SB2_OFF + 32768 <= T2C_OFF && T2C_OFF + 128 * ROWB <= Y2_OFF && Y2_OFF + 65536 <= R2_LDS && T2C_OFF + 128 * ROWB <= T1C_OFF, "b2 map");

__device__ __forceinline__ const uint4* wfr(const void* base, int ct, int KS, int ks, int lane) {
    return reinterpret_cast<const uint4*>(static_cast<const char*>(base) + ((size_t)(ct * KS + ks) * 64 + lane) * 16);
}
__device__ __forceinline__ u32x4 u4(uint4 v) {
    u32x4 r = {v.x, v.y, v.z, v.w};
    return r;
}
template <int N>
__device__ __forceinline__ void lgkm(void) {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void wg_barrier() {
    lds_wait();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// slot -> pixel offset from the tile origin (see "Pixel order"); unused slots map to the origin (their results are dropped)
__device__ __forceinline__ void slot_pos(int s, int& dy, int& dx) {
    dy = 0; dx = 0;
    if (s < 128) { dy = s >> 4; dx = s & 15; }
    else if (s < 180) {
        const int r = s - 128;
        if (r < 18) { dy = -1; dx = r - 1; } else if (r < 36) { dy = 8; dx = r - 19; } else if (r < 44) { dy = r - 36; dx = -1; } else { dy = r - 44; dx = 16; }
    } else if (s >= 192 && s < 252) {
        const int r = s - 192;
        if (r < 20) { dy = -2; dx = r - 2; } else if (r < 40) { dy = 9; dx = r - 22; } else if (r < 50) { dy = r - 41; dx = -2; } else { dy = r - 51; dx = 17; }
    }
}
__device__ __forceinline__ bool slot_used(int s) { return s < 180 || (s >= 192 && s < 252); }

// The lane's pixel position in pixel tile t (slot t * 32 + lrow).  Core tiles (t < 4) have a closed form; the ring tiles' branchy
// decoding (slot_pos) is done ONCE per kernel and kept packed in two registers — it used to run ~25 times per wave and tile (every
// phase that turns slots into raster addresses), ~15 % of the non-MFMA instructions of a kernel that is VALU-issue-bound.
struct TilePos {
    unsigned ring[2];                                // tiles 4 | 5 and 6 | 7: (dy + 2) | (dx + 2) << 8 in each half
    int lrow;
    __device__ __forceinline__ void init(int lrow_) {
        lrow = lrow_;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            int dy0, dx0, dy1, dx1;
            slot_pos((4 + 2 * k) * 32 + lrow, dy0, dx0);
            slot_pos((5 + 2 * k) * 32 + lrow, dy1, dx1);
            ring[k] = (unsigned)((dy0 + 2) | ((dx0 + 2) << 8)) | ((unsigned)((dy1 + 2) | ((dx1 + 2) << 8)) << 16);
        }
    }
    __device__ __forceinline__ void get(int t, int& dy, int& dx) const {      // t is wave-uniform
        if (t < 4) {
            dy = 2 * t + (lrow >> 4); dx = lrow & 15;
        } else {
            const unsigned r = (t < 6 ? ring[0] : ring[1]) >> ((t & 1) * 16);
            dy = (int)(r & 0xff) - 2; dx = (int)((r >> 8) & 0xff) - 2;
        }
    }
};

// Weights and biases of a phase are REQUESTED ONE PHASE AHEAD (plain loads pinned by a scheduling fence: hipcc otherwise sinks
// them to their first use) — the first timeline of this kernel showed every phase opening with an exposed L2 round trip
// (~1-2 k cycles of a 3-6 k cycle phase) and the 3x3's three-k-step weight ring stalling on every refill.
// Register budget (256 per lane, residuals of the next block live across phases): a phase keeps a RING of weight fragments — 9
// for the 3x3 (18 k-steps per K half), 8 for the forwarded 1x1 (16 k-steps): fragment j sits in slot j % ring; the first ring is
// requested during the previous phase's MFMA loop, slot k is refilled right after step k consumed it (a ring's worth of k-steps
// ahead of its use).
struct WB { uint4 w[9]; float4 b[4]; };              // phase B: ring of one K half (18 k-steps) of one channel tile + bias
struct WC { uint4 w[8]; float4 b[4]; };              // phase C: one channel tile (4 k-steps, 8 with the downsample) + bias
struct WD { uint4 w[8]; float4 b[4]; };              // phase D: ring of one output-channel tile (K = 256: 16 k-steps) + bias
__device__ __forceinline__ void pin() { __builtin_amdgcn_sched_barrier(0); }

template <int DBG>
__device__ __forceinline__ f32x16 mm(u32x4 a, u32x4 b, f32x16 c) {
    if constexpr (DBG & 4) {
        asm volatile("" ::"v"(a), "v"(b));
        return c;
    } else return mfma_bf16(a, b, c);
}

// ---------------------------------------------------------------- phase B: 3x3 on a padded raster, NPT pixel tiles of slots
// t1 raster of width SRCW at T1_OFF -> t2 [slot][128 B swizzled] at T2_OFF; partial sums through SB_OFF.  `ahead` runs between
// the k-loop and the exchange: the caller requests the next phase's weights there.
template <int NPT, int SRCW, int DBG, int PER, int DEPTH, typename FO, typename F>
__device__ __forceinline__ void phase_b(unsigned char* lds, unsigned lds_base, int T1_OFF, int T2_OFF, int SB_OFF, WB& wb, int wave,
                                        int lane, const TilePos& tp, FO&& own, F&& ahead) {
    constexpr int NT = NPT / 2, OFS = (SRCW - 16) / 2 - 1;
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int ct = wave & 1, q = (wave >> 1) & 1, kh = wave >> 2;
    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    unsigned rb[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        int dy, dx;
        tp.get(q + 2 * i, dy, dx);
        rb[i] = lds_base + T1_OFF + ((dy + OFS) * SRCW + dx + OFS) * T1R + lhalf * 16;
    }
    auto run_half = [&](auto khc) {
        constexpr int K0 = decltype(khc)::value * 18;
        // pixel fragments DEPTH k-steps ahead of the MFMAs that use them (a ring of DEPTH + 1 sets): with two or three MFMAs per
        // k-step and two waves per SIMD, one step of lead (~128 - 256 cycles) is about the LDS round trip under eight waves' load
        u32x4 pf[DEPTH + 1][NT];
        auto rd = [&](auto kc) {
            constexpr int kstep = K0 + decltype(kc)::value, buf = decltype(kc)::value % (DEPTH + 1);
            constexpr int tap = kstep >> 2, ks = kstep & 3, toff = (tap / 3) * SRCW + (tap % 3);
#pragma unroll
            for (int i = 0; i < NT; ++i) pf[buf][i] = lds_read_b128_o<toff * T1R + ks * 32>(rb[i]);
        };
        lgkm<0>();                                   // nothing of the compiler's (LDS or scalar loads) may be outstanding when the counted waits start
        static_for<0, DEPTH>([&](auto kc) { rd(kc); });
        static_for<0, 18>([&](auto kc) {
            constexpr int KS = decltype(kc)::value;
            if constexpr (KS + DEPTH < 18 && !(DBG & 2)) rd(std::integral_constant<int, KS + DEPTH>{});
            if constexpr (KS >= 9) static_for<0, PER>([&](auto jc) { ahead(std::integral_constant<int, (KS - 9) * PER + decltype(jc)::value>{}); });
            constexpr int INFLIGHT = (17 - KS < DEPTH ? 17 - KS : DEPTH) * NT;       // reads younger than this step's
            if (DBG & 2) { if (KS == 0) lgkm<0>(); } else lgkm<INFLIGHT>();
            const u32x4 w = u4(wb.w[KS % 9]);
#pragma unroll
            for (int i = 0; i < NT; ++i) acc[i] = mm<DBG>(w, pf[(DBG & 2) ? 0 : (KS % (DEPTH + 1))][i], acc[i]);
            if constexpr (KS < 9) {                  // slot KS is free: fragment 9 + KS of this K half
                __builtin_amdgcn_sched_barrier(0);
                own(kc);
            }
        });
    };
    if (kh) run_half(std::integral_constant<int, 1>{}); else run_half(std::integral_constant<int, 0>{});
    // exchange: the lower K half finishes the even tiles of the pair, the upper half the odd ones
    float* scr = reinterpret_cast<float*>(lds + SB_OFF + (ct + 2 * q) * NT * 4096);
#pragma unroll
    for (int i = 0; i < NT; ++i)
        if ((i & 1) != kh) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(scr + ((i * 4 + g) * 64 + lane) * 4) = make_float4(acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]);
        }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NT; ++i)
        if ((i & 1) == kh) {
            const int row = (q + 2 * i) * 32 + lrow;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 o = *reinterpret_cast<const float4*>(scr + ((i * 4 + g) * 64 + lane) * 4);
                u32x2 pk;
                pk.x = relu2_bf16(acc[i][4 * g] + o.x + wb.b[g].x, acc[i][4 * g + 1] + o.y + wb.b[g].y);
                pk.y = relu2_bf16(acc[i][4 * g + 2] + o.z + wb.b[g].z, acc[i][4 * g + 3] + o.w + wb.b[g].w);
                lds_write_b64(lds_base + T2_OFF + row * ROWB + (((ct * 4 + g) ^ ((row >> 1) & 7)) << 4) + 8 * lhalf, pk);
            }
        }
}

// ---------------------------------------------------------------- phase C for one channel tile x NTC pixel tiles (parity h)
// acc = bias + W3[ct] t2 (4 k-steps) [+ Wd[ct] p (4 k-steps, DS)]; the caller finishes (residual, ReLU, stores)
template <int NTC, bool DS, int DBG, int PER, typename F>
__device__ __forceinline__ void phase_c_mma(f32x16 (&acc)[NTC], unsigned lds_base, int T2_OFF, const WC& wc, int h, int lane, F&& ahead) {
    // h = first pixel tile of the wave's set (tiles h, h + 2, ...)
    const int lrow = lane & 31, lhalf = lane >> 5;
#pragma unroll
    for (int i = 0; i < NTC; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            acc[i][4 * g] = wc.b[g].x; acc[i][4 * g + 1] = wc.b[g].y; acc[i][4 * g + 2] = wc.b[g].z; acc[i][4 * g + 3] = wc.b[g].w;
        }
    const int row0 = h * 32 + lrow;                  // tiles h, h+2, ...: rows 64 apart share the swizzle key
    const unsigned trow = lds_base + T2_OFF + row0 * ROWB;
    const int key = (row0 >> 1) & 7;
    u32x4 pf[2][NTC];
    auto rd = [&](int ks, int buf) {
        const unsigned a = trow + (((2 * ks + lhalf) ^ key) << 4);
        static_for<0, NTC>([&](auto ic) { pf[buf][decltype(ic)::value] = lds_read_b128_o<decltype(ic)::value * 64 * ROWB>(a); });
    };
    unsigned prow[NTC];
    int pkey[NTC];
    if constexpr (DS) {
        // p in slot order (PD: 240 compact rows — slots [0,180) and [192,252)); per-tile row and key
#pragma unroll
        for (int i = 0; i < NTC; ++i) {
            const int s = (h + 2 * i) * 32 + lrow;
            const int rc = s < 180 ? s : ((s >= 192 && s < 252) ? s - 12 : 0);
            prow[i] = lds_base + PD_OFF + rc * ROWB;
            pkey[i] = (rc >> 1) & 7;
        }
    }
    auto rdp = [&](int ks, int buf) {
#pragma unroll
        for (int i = 0; i < NTC; ++i) pf[buf][i] = lds_read_b128(prow[i] + (((2 * ks + lhalf) ^ pkey[i]) << 4));
    };
    lgkm<0>();
    rd(0, 0);
    static_for<0, 4>([&](auto kc) {
        constexpr int ks = decltype(kc)::value;
        if (!(DBG & 2)) { if (ks < 3) rd(ks + 1, (ks + 1) & 1); else if (DS) rdp(0, 0); }
        static_for<0, PER>([&](auto jc) { ahead(std::integral_constant<int, ks * PER + decltype(jc)::value>{}); });
        if (DBG & 2) lgkm<0>(); else if (ks < 3 || DS) lgkm<NTC>(); else lgkm<0>();
        const u32x4 ww = u4(wc.w[ks]);
#pragma unroll
        for (int i = 0; i < NTC; ++i) acc[i] = mm<DBG>(ww, pf[(DBG & 2) ? 0 : (ks & 1)][i], acc[i]);
    });
    if constexpr (DS) {
        static_for<0, 4>([&](auto kc) {
            constexpr int ks = decltype(kc)::value;
            if (ks < 3 && !(DBG & 2)) rdp(ks + 1, (ks + 1) & 1);
            static_for<0, PER>([&](auto jc) { ahead(std::integral_constant<int, (4 + ks) * PER + decltype(jc)::value>{}); });
            if (DBG & 2) lgkm<0>(); else if (ks < 3) lgkm<NTC>(); else lgkm<0>();
            const u32x4 ww = u4(wc.w[4 + ks]);
#pragma unroll
            for (int i = 0; i < NTC; ++i) acc[i] = mm<DBG>(ww, pf[(DBG & 2) ? 0 : (ks & 1)][i], acc[i]);
        });
    }
}

// finish one (channel tile, pixel tile): + residual (packed bf16 pairs, or none), ReLU, bf16 -> `out` (8 packed registers) and the
// y image: slice (64 channels) `sl`, 16-byte chunk base `cb` (0 / 4) of row `row`, image of NROWS rows per slice at Y_OFF
template <bool RES>
__device__ __forceinline__ void finish_c(const f32x16& a, const unsigned (&res)[8], unsigned (&out)[8], unsigned lds_base, int Y_OFF,
                                         int slice_bytes, int sl, int cb, int row, int lhalf) {
    const unsigned ya = lds_base + Y_OFF + sl * slice_bytes + row * ROWB + 8 * lhalf;
    const int key = (row >> 1) & 7;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float v[4] = {a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
        if (RES) {
            v[0] += __uint_as_float(res[2 * g] << 16); v[1] += __uint_as_float(res[2 * g] & 0xffff0000u);
            v[2] += __uint_as_float(res[2 * g + 1] << 16); v[3] += __uint_as_float(res[2 * g + 1] & 0xffff0000u);
        }
        u32x2 pk;
        pk.x = relu2_bf16(v[0], v[1]);
        pk.y = relu2_bf16(v[2], v[3]);
        out[2 * g] = pk.x; out[2 * g + 1] = pk.y;
        lds_write_b64(ya + (((cb + g) ^ key) << 4), pk);
    }
}

// ---------------------------------------------------------------- phase D k-steps: dacc[i] += w[k] y[tile i], k < NKS
// y image: slices of 64 channels, NROWS rows each; tiles of a wave are PSTRIDE tiles apart (rows PSTRIDE * 32 apart: same key)
// TWO (second pixel tile or not) is a TEMPLATE parameter: as a run-time flag it put wave-uniform branches inside the counted-wait loop,
// which neither hipcc's scheduler nor tools/asm_inflight_scan.py (a linear scan) follows well
template <int NKS, int NROWS, int PSTRIDE, int DBG, int PER, bool two, typename FO, typename F>
__device__ __forceinline__ void phase_d_mma(f32x16 (&dacc)[2], unsigned lds_base, int Y_OFF, WD& wd, int row0, int lane, FO&& own,
                                            F&& ahead) {
    const int lhalf = lane >> 5;
    const unsigned yrow = lds_base + Y_OFF + row0 * ROWB;
    const int key = (row0 >> 1) & 7;
    u32x4 pd[2][2];
    auto rd = [&](auto kc, int buf) {
        constexpr int K = decltype(kc)::value, SL = K >> 2, k = K & 3;
        constexpr int OB = SL * NROWS * ROWB, FAR = OB + PSTRIDE * 32 * ROWB >= 65536 ? OB : 0;   // ds offsets are 16 bits
        const unsigned a = yrow + (((2 * k + lhalf) ^ key) << 4) + FAR;
        pd[buf][0] = lds_read_b128_o<OB - FAR>(a);
        if (two) pd[buf][1] = lds_read_b128_o<OB - FAR + PSTRIDE * 32 * ROWB>(a);
    };
    lgkm<0>();
    rd(std::integral_constant<int, 0>{}, 0);
    static_for<0, NKS>([&](auto kc) {
        constexpr int K = decltype(kc)::value;
        if constexpr (K + 1 < NKS && !(DBG & 2)) rd(std::integral_constant<int, K + 1>{}, (K + 1) & 1);
        if constexpr (NKS == 8 || K >= 8) static_for<0, PER>([&](auto jc) { ahead(std::integral_constant<int, (NKS == 8 ? K : K - 8) * PER + decltype(jc)::value>{}); });
        if (DBG & 2) { if (K == 0) lgkm<0>(); } else if (K + 1 < NKS) { if (two) lgkm<2>(); else lgkm<1>(); } else lgkm<0>();
        const u32x4 ww = u4(wd.w[K % 8]);
        dacc[0] = mm<DBG>(ww, pd[(DBG & 2) ? 0 : (K & 1)][0], dacc[0]);
        if (two) dacc[1] = mm<DBG>(ww, pd[(DBG & 2) ? 0 : (K & 1)][1], dacc[1]);
        if constexpr (NKS == 16 && K < 8) {          // slot K is free: fragment 8 + K
            __builtin_amdgcn_sched_barrier(0);
            own(kc);
        }
    });
}

// where a slot's t1 goes in a raster of width RW whose origin sits OFS pixels up-left of the tile origin: LDS byte address of the
// row (0xffffffff: unused slot) and whether the pixel lies inside the frame (else the 3x3's zero padding is stored)
template <int RW, int OFS>
__device__ __forceinline__ void t1_target(const TilePos& tp, int tile, int y0, int x0, unsigned base, unsigned& addr, unsigned& mask) {
    int dy, dx;
    tp.get(tile, dy, dx);
    const int y = y0 + dy, x = x0 + dx;
    mask = (y >= 0 && y < 64 && x >= 0 && x < 64) ? 0xffffffffu : 0u;
    addr = slot_used(tile * 32 + tp.lrow) ? base + ((dy + OFS) * RW + dx + OFS) * T1R : 0xffffffffu;
}
// one branch per tile (unused slots), the frame mask as an AND: no per-store predication
__device__ __forceinline__ void store_t1(const f32x16& d, const float4 (&b)[4], unsigned addr, unsigned mask, int ct, int lhalf) {
    if (addr == 0xffffffffu) return;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        u32x2 pk;
        pk.x = relu2_bf16(d[4 * g] + b[g].x, d[4 * g + 1] + b[g].y) & mask;
        pk.y = relu2_bf16(d[4 * g + 2] + b[g].z, d[4 * g + 3] + b[g].w) & mask;
        lds_write_b64(addr + ((ct * 4 + g) << 4) + 8 * lhalf, pk);
    }
}
}  // namespace

template <bool YS2, int DBG = 0>
__global__ __launch_bounds__(512, 2) void res2_stage_kernel(Res2StageArgs kargs) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[R2_LDS];
    // Scalar loads share lgkmcnt with the LDS reads and return out of order, so an s_load INSIDE the MFMA loops would make the counted
    // s_waitcnt lgkmcnt(N) of the fragment pipeline release before its own reads had landed.  hipcc re-reads kernel arguments that way
    // under SGPR pressure (the persistent variant of this kernel did, and computed garbage).  In this one-tile-per-workgroup form all
    // argument loads sit in the prologue; tools/asm_inflight_scan.py fails the build check (tests/test_cabi.py) if one ever appears
    // behind the first barrier.  (Laundering every argument into SGPR locals also prevents it, but costs 5 %: 64.6 k vs 60.0 k cycles
    // per tile - the 60 live SGPRs are spilled to VGPR lanes and read back with v_readlane in front of every weight load.)
    const Res2StageArgs& p = kargs;
    const int tid = threadIdx.x;
    int lane = tid & 63;
    int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lrow = lane & 31, lhalf = lane >> 5;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const bf16_t* zeros = static_cast<const bf16_t*>(p.zeros);
    const int ntiles = p.B * 32, wslot0 = xcd_remap(blockIdx.x, gridDim.x);    // consecutive logical tiles on one XCD: neighbours share halos through that L2
    if (wslot0 >= ntiles) return;
    const int wslot = p.rev ? ntiles - 1 - wslot0 : wslot0;
    int b, y0, x0;
    const bf16_t* X;
    auto set_tile = [&](int t) {
        b = t >> 5;
        y0 = ((t & 31) >> 2) * 8; x0 = (t & 3) * 16;
        X = static_cast<const bf16_t*>(p.x) + (size_t)b * 64 * 64 * 64;
    };
    // p halo of tile t: 14 x 22 raster, 39 groups of 8 rows, by LDS-DMA
    auto issue_p = [&](int t) {
        const int tb = t >> 5, ty0 = ((t & 31) >> 2) * 8, tx0 = (t & 3) * 16;
        const bf16_t* TX = static_cast<const bf16_t*>(p.x) + (size_t)tb * 64 * 64 * 64;
        const int rsub = lane >> 3, cpos = lane & 7;
        for (int g = wave; g < 39; g += 8) {
            const int row = g * 8 + rsub;
            const int hy = row / 22, hx = row - hy * 22;
            const int y = ty0 - 3 + hy, x = tx0 - 3 + hx;
            const bool ok = row < 308 && y >= 0 && y < 64 && x >= 0 && x < 64;
            const bf16_t* src = ok ? TX + ((size_t)y * 64 + x) * 64 + (cpos ^ ((row >> 1) & 7)) * 8 : zeros;
            dma16(src, lds + P_OFF + g * 1024);
        }
    };
    int stamped = 0;
    auto stamp = [&](int k) {
        if (p.ts && tid == 0 && stamped) p.ts[(size_t)blockIdx.x * 16 + k] = __builtin_amdgcn_s_memtime();
    };
    int bct = wave & 1, bkh = wave >> 2;             // phase B: channel tile, K half
    int cc = wave & 3, hh = wave >> 2;               // phase C: channel tiles cc, cc + 4; pixel tiles hh, hh + 2, ...
    int dct = wave & 1, dv = wave >> 1;              // phase D0 / D1: channel tile, pixel tiles dv, dv + 4
    auto gbar1 = [&]() { wg_barrier(); };
    auto gbar2 = [&]() { wg_barrier(); };

    WB wb;
    WC wc[2];
    WD wd;
    // "ahead" loads: request N of the NEXT phase's weights / biases, issued one or two per k-step of the CURRENT phase's MFMA loop
    // (a burst of 20 loads per wave at a phase boundary blocks the wave at issue: 1.1 MB of fragments per tile cross the texture
    // cache at 64 B/clk, and none of it overlapped the MFMAs)
    auto ld_wb = [&](auto nc, const void* fb, const float* bb) {         // first ring of a K half + bias: 13 requests
        constexpr int N = decltype(nc)::value;
        if constexpr (N < 9) wb.w[N] = (DBG & 1) ? make_uint4(0, 0, 0, 0) : *wfr(fb, bct, 36, bkh * 18 + N, lane);
        else if constexpr (N < 13) wb.b[N - 9] = *reinterpret_cast<const float4*>(bb + bct * 32 + 8 * (N - 9) + 4 * lhalf);
    };
    auto own_wb = [&](auto kc, const void* fb) {                         // refill of slot K inside the phase's own loop
        constexpr int K = decltype(kc)::value;
        wb.w[K] = (DBG & 1) ? make_uint4(0, 0, 0, 0) : *wfr(fb, bct, 36, bkh * 18 + 9 + K, lane);
    };
    auto ld_wc = [&](auto nc, auto ksc, WC& o, const void* fc, const float* bc, int ct) {
        constexpr int N = decltype(nc)::value, KS = decltype(ksc)::value;
        if constexpr (N < KS) o.w[N] = (DBG & 1) ? make_uint4(0, 0, 0, 0) : *wfr(fc, ct, KS, N, lane);
        else if constexpr (N < KS + 4) o.b[N - KS] = *reinterpret_cast<const float4*>(bc + ct * 32 + 8 * (N - KS) + 4 * lhalf);
    };
    auto ld_wd = [&](auto nc, const void* fd, const float* bd, int ct, int ks0) {     // first ring (8 k-steps from ks0) + bias: 12 requests
        constexpr int N = decltype(nc)::value;
        if constexpr (N < 8) wd.w[N] = (DBG & 1) ? make_uint4(0, 0, 0, 0) : *wfr(fd, ct, 16, ks0 + N, lane);
        else if constexpr (N < 12) wd.b[N - 8] = *reinterpret_cast<const float4*>(bd + ct * 32 + 8 * (N - 8) + 4 * lhalf);
    };
    auto own_wd = [&](auto kc, const void* fd, int ct) {
        constexpr int K = decltype(kc)::value;
        wd.w[K] = (DBG & 1) ? make_uint4(0, 0, 0, 0) : *wfr(fd, ct, 16, 8 + K, lane);
    };
    auto none_ahead = [](auto) {};
    struct WA { uint4 w[4]; float4 b[4]; } wa;       // A0: conv1 of block 0, one channel tile + bias
    int act = wave & 1, ga = wave >> 1;              // phase A0: channel tile, pixel tiles ga, ga + 4, ga + 8 (< 10)
    int d2ct = wave & 3, d2v = wave >> 2;            // phase D2: output-channel tile (of 4), pixel tiles d2v, d2v + 2
    auto ld_wa = [&](auto nc) {                      // 8 requests
        constexpr int N = decltype(nc)::value;
        if constexpr (N < 4) wa.w[N] = (DBG & 1) ? make_uint4(0, 0, 0, 0) : *wfr(p.fa0, act, 4, N, lane);
        else if constexpr (N < 8) wa.b[N - 4] = *reinterpret_cast<const float4*>(p.ba0 + act * 32 + 8 * (N - 4) + 4 * lhalf);
    };
    // one tile per workgroup (a persistent, grid-stride variant with the next tile's halo prefetched during block 2 measured
    // SLOWER, 1083 vs 986 us at B = 256: vmcnt retires in order, so every weight wait issued after the prefetch DMA stalls until that
    // DMA has landed, and loads placed under a branch make hipcc drain vmcnt(0) at every block boundary)
    const int tile = wslot;
    set_tile(tile);
    TilePos tp;
    tp.init(lrow);
#ifdef R2_SETPRIO
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);    // (experiment) static priority for the second-dispatched half of the workgroup
#endif
    stamped = 1;
    stamp(0);
    issue_p(tile);
    pin();
    static_for<0, 8>([&](auto n) { ld_wa(n); });
    static_for<0, 13>([&](auto n) { ld_wb(n, p.fb[0], p.bb[0]); });
    pin();
    wait_vmcnt<21>();                                // the halo (older than the 21 weight requests) has landed; the weights may still be on their way
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  {
    // ================================================================ A0: t1_0 = relu(W1 p + b) on the 14 x 22 raster (10 pixel tiles)
    {
        const int ct = act;
        const bool three = ga < 2;
        f32x16 acc[3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        stamp(1);
        const int row0 = ga * 32 + lrow;             // tiles 4 apart: rows 128 apart, same key
        const int key = (row0 >> 1) & 7;
        // plain (compiler-tracked) LDS loads here: no LDS-DMA is in flight during A0, and hipcc may spill / copy the destination
        // of an inline-asm ds_read BEFORE its data has landed (it believes the asm's output is valid at once) — which it did in this
        // low-intensity phase of the persistent kernel.  tools/asm_inflight_scan.py checks the hot loops for that.
        const unsigned char* prow = lds + P_OFF + row0 * ROWB;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int co = ((2 * ks + lhalf) ^ key) << 4;
            const u32x4 ww = u4(wa.w[ks]);
            const u32x4 f0 = *reinterpret_cast<const u32x4*>(prow + co);
            const u32x4 f1 = *reinterpret_cast<const u32x4*>(prow + co + 128 * ROWB);
            acc[0] = mm<DBG>(ww, f0, acc[0]);
            acc[1] = mm<DBG>(ww, f1, acc[1]);
            if (three) {
                const u32x4 f2 = *reinterpret_cast<const u32x4*>(prow + co + 256 * ROWB);
                acc[2] = mm<DBG>(ww, f2, acc[2]);
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (i == 2 && !three) break;
            const int hr = (ga + 4 * i) * 32 + lrow;
            const int hy = hr / 22, hx = hr - hy * 22;
            const int y = y0 - 3 + hy, x = x0 - 3 + hx;
            const unsigned mask = (y >= 0 && y < 64 && x >= 0 && x < 64) ? 0xffffffffu : 0u;
            if (hr < 308) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2 pk;
                    pk.x = relu2_bf16(acc[i][4 * g] + wa.b[g].x, acc[i][4 * g + 1] + wa.b[g].y) & mask;
                    pk.y = relu2_bf16(acc[i][4 * g + 2] + wa.b[g].z, acc[i][4 * g + 3] + wa.b[g].w) & mask;
                    lds_write_b64(lds_base + T1A_OFF + hr * T1R + ((ct * 4 + g) << 4) + 8 * lhalf, pk);
                }
            }
        }
        wg_barrier();                                // P is dead, t1_0 complete
        stamp(2);
    }
    // p once more, now in slot order (C0's downsample operand): 30 groups of 8 compact rows by LDS-DMA, lands during B0
    {
        const int rsub = lane >> 3, cpos = lane & 7;
        for (int g = wave; g < 30; g += 8) {
            const int rc = g * 8 + rsub;
            int dy, dx;
            slot_pos(rc < 180 ? rc : rc + 12, dy, dx);
            const int y = y0 + dy, x = x0 + dx;
            const bool ok = y >= 0 && y < 64 && x >= 0 && x < 64;
            const bf16_t* src = ok ? X + ((size_t)y * 64 + x) * 64 + (cpos ^ ((rc >> 1) & 7)) * 8 : zeros;
            dma16(src, lds + PD_OFF + g * 1024);
        }
    }
    // ================================================================ block 0
    using I8 = std::integral_constant<int, 8>;
    using I4 = std::integral_constant<int, 4>;
    phase_b<8, 22, DBG, 2, R2_DEPTH0>(lds, lds_base, T1A_OFF, T2A_OFF, SB0_OFF, wb, wave, lane, tp, [&](auto k) { own_wb(k, p.fb[0]); },
                           [&](auto n) { ld_wc(n, I8{}, wc[0], p.fc[0], p.bc[0], cc); });
    wait_vmcnt<0>();                                 // this wave's share of PD has landed
    wg_barrier();
    stamp(3);

    unsigned yres[2][3][8];                          // residual of the next block: [channel tile][pixel tile][packed pairs]
    unsigned none[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    {
        f32x16 dacc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) dacc[i][r] = 0.f;
        unsigned ta[2], tin[2];
        t1_target<20, 2>(tp, dv, y0, x0, lds_base + T1B_OFF, ta[0], tin[0]);
        t1_target<20, 2>(tp, dv + 4, y0, x0, lds_base + T1B_OFF, ta[1], tin[1]);
        static_for<0, 2>([&](auto rc_) {
            constexpr int R = decltype(rc_)::value;
            static_for<0, 2>([&](auto hc_) {            // two tile pairs: 4 accumulator tiles + D0's two would not fit beside the residuals
                constexpr int HP = decltype(hc_)::value;
                f32x16 acc[2];
                phase_c_mma<2, true, DBG, 1>(acc, lds_base, T2A_OFF, wc[R], hh + 4 * HP, lane, [&](auto n) {
                    if constexpr (HP == 0 || decltype(n)::value < 4) ld_wd(std::integral_constant<int, 8 * HP + decltype(n)::value>{}, p.fd[0], p.bd[0], dct, 8 * R);
                });
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    unsigned out[8];
                    finish_c<false>(acc[i], none, out, lds_base, Y0_OFF, 256 * ROWB, cc >> 1, (cc & 1) * 4, (hh + 4 * HP + 2 * i) * 32 + lrow, lhalf);
                    if (2 * HP + i < 3) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) yres[R][2 * HP + i][k] = out[k];
                    }
                }
            });
            gbar2();                                 // y0 K half R is in the image (a group's rows are its own: conv3 and the next conv1 are 1x1)
            if constexpr (R == 0) {
                phase_d_mma<8, 256, 4, DBG, 2, true>(dacc, lds_base, Y0_OFF, wd, dv * 32 + lrow, lane, none_ahead, [&](auto n) { ld_wc(n, I8{}, wc[1], p.fc[0], p.bc[0], 4 + cc); });
                gbar2();                             // image free for the second half
            } else {
                phase_d_mma<8, 256, 4, DBG, 2, true>(dacc, lds_base, Y0_OFF, wd, dv * 32 + lrow, lane, none_ahead, [&](auto n) { ld_wb(n, p.fb[1], p.bb[1]); });
            }
        });
        stamp(4);
        store_t1(dacc[0], wd.b, ta[0], tin[0], dct, lhalf);
        store_t1(dacc[1], wd.b, ta[1], tin[1], dct, lhalf);
        wg_barrier();                                // t1_1 complete; y image, t2_0, PD dead
        stamp(5);
    }
    // ================================================================ block 1
    phase_b<6, 20, DBG, 1, R2_DEPTH1>(lds, lds_base, T1B_OFF, T2B_OFF, SB1_OFF, wb, wave, lane, tp, [&](auto k) { own_wb(k, p.fb[1]); },
                           [&](auto n) { ld_wc(n, I4{}, wc[0], p.fc[1], p.bc[1], cc); });
    wg_barrier();
    stamp(6);
    unsigned y1res[2][2][8];
    static_for<0, 2>([&](auto rc_) {
        constexpr int R = decltype(rc_)::value;
        f32x16 acc[3];
        // round 0 also requests round 1's weights (8), both request half of D1's first ring + bias (6 each)
        phase_c_mma<3, false, DBG, 4>(acc, lds_base, T2B_OFF, wc[R], hh, lane, [&](auto n) {
            constexpr int N = decltype(n)::value;
            if constexpr (R == 0 && N < 8) ld_wc(n, I4{}, wc[1], p.fc[1], p.bc[1], 4 + cc);
            else if constexpr (N >= 8 && N < 14) ld_wd(std::integral_constant<int, 6 * R + N - 8>{}, p.fd[1], p.bd[1], dct, 0);
        });
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            unsigned out[8];
            finish_c<true>(acc[i], yres[R][i], out, lds_base, Y1_OFF, 192 * ROWB, (4 * R + cc) >> 1, (cc & 1) * 4, (hh + 2 * i) * 32 + lrow, lhalf);
            if (i < 2) {
#pragma unroll
                for (int k = 0; k < 8; ++k) y1res[R][i][k] = out[k];
            }
        }
    });
    gbar2();                                         // y1 image complete
    stamp(7);
    {
        const bool two = dv < 2;                     // D1: pixel tiles dv, dv + 4 (< 6)
        f32x16 dacc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) dacc[i][r] = 0.f;
        unsigned ta[2], tin[2];
        t1_target<18, 1>(tp, dv, y0, x0, lds_base + T1C_OFF, ta[0], tin[0]);
        t1_target<18, 1>(tp, dv + 4, y0, x0, lds_base + T1C_OFF, ta[1], tin[1]);
        if (two) phase_d_mma<16, 192, 4, DBG, 2, true>(dacc, lds_base, Y1_OFF, wd, dv * 32 + lrow, lane, [&](auto k) { own_wd(k, p.fd[1], dct); },
                                                       [&](auto n) { ld_wb(n, p.fb[2], p.bb[2]); });
        else phase_d_mma<16, 192, 4, DBG, 2, false>(dacc, lds_base, Y1_OFF, wd, dv * 32 + lrow, lane, [&](auto k) { own_wd(k, p.fd[1], dct); },
                                                    [&](auto n) { ld_wb(n, p.fb[2], p.bb[2]); });
        store_t1(dacc[0], wd.b, ta[0], tin[0], dct, lhalf);
        if (two) store_t1(dacc[1], wd.b, ta[1], tin[1], dct, lhalf);
        wg_barrier();                                // t1_2 complete; y image, t2_1 dead
        stamp(8);
    }
    // ================================================================ block 2
    phase_b<4, 18, DBG, 1, R2_DEPTH2>(lds, lds_base, T1C_OFF, T2C_OFF, SB2_OFF, wb, wave, lane, tp, [&](auto k) { own_wb(k, p.fb[2]); },
                           [&](auto n) { ld_wc(n, I4{}, wc[0], p.fc[2], p.bc[2], cc); });
    gbar1();                                         // t2 of the group's pixel tiles complete (both channel tiles and K halves are the group's)
    stamp(9);
    static_for<0, 2>([&](auto rc_) {
        constexpr int R = decltype(rc_)::value;
        f32x16 acc[2];
        phase_c_mma<2, false, DBG, 4>(acc, lds_base, T2C_OFF, wc[R], hh, lane, [&](auto n) {
            constexpr int N = decltype(n)::value;
            if constexpr (R == 0 && N < 8) ld_wc(n, I4{}, wc[1], p.fc[2], p.bc[2], 4 + cc);
            else if constexpr (N >= 8 && N < 14) ld_wd(std::integral_constant<int, 6 * R + N - 8>{}, p.fd[2], p.bd[2], d2ct, 0);
        });
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            unsigned out[8];
            finish_c<true>(acc[i], y1res[R][i], out, lds_base, Y2_OFF, 128 * ROWB, (4 * R + cc) >> 1, (cc & 1) * 4, (hh + 2 * i) * 32 + lrow, lhalf);
        }
    });
    gbar1();                                         // y2 image complete (the group's rows)
    stamp(10);
    // D2: res3's conv1 (256 -> 128) on the 8 x 16 tile -> HBM
    {
        f32x16 dacc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) dacc[i][r] = 0.f;
        phase_d_mma<16, 128, 2, DBG, 1, true>(dacc, lds_base, Y2_OFF, wd, d2v * 32 + lrow, lane, [&](auto k) { own_wd(k, p.fd[2], d2ct); }, none_ahead);
        bf16_t* T1O = static_cast<bf16_t*>(p.t1out) + (size_t)b * 64 * 64 * 128;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = (d2v + 2 * i) * 32 + lrow;
            bf16_t* o = T1O + ((size_t)(y0 + (q >> 4)) * 64 + x0 + (q & 15)) * 128 + d2ct * 32 + 4 * lhalf;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 pk;
                pk.x = relu2_bf16(dacc[i][4 * g] + wd.b[g].x, dacc[i][4 * g + 1] + wd.b[g].y);
                pk.y = relu2_bf16(dacc[i][4 * g + 2] + wd.b[g].z, dacc[i][4 * g + 3] + wd.b[g].w);
                *reinterpret_cast<uint2*>(o + 8 * g) = pk;
            }
        }
    }
    // y2 out: 16-byte chunks from the image, 512 B per pixel contiguous
    {
        constexpr int NJ = (YS2 ? 1024 : 4096) / 512;
        u32x4 v[NJ];
        bf16_t* dst[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int e = tid + 512 * j;
            const int pe = e >> 5, c16 = e & 31;
            int slot;
            if (YS2) {
                slot = (2 * (pe >> 3)) * 16 + 2 * (pe & 7);
                dst[j] = static_cast<bf16_t*>(p.y) + (((size_t)b * 32 + (y0 >> 1) + (pe >> 3)) * 32 + (x0 >> 1) + (pe & 7)) * 256 + c16 * 8;
            } else {
                slot = pe;
                dst[j] = static_cast<bf16_t*>(p.y) + (((size_t)b * 64 + y0 + (pe >> 4)) * 64 + x0 + (pe & 15)) * 256 + c16 * 8;
            }
            v[j] = *reinterpret_cast<const u32x4*>(lds + Y2_OFF + (c16 >> 3) * 128 * ROWB + slot * ROWB + (((c16 & 7) ^ ((slot >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) *reinterpret_cast<uint4*>(dst[j]) = make_uint4(v[j][0], v[j][1], v[j][2], v[j][3]);
    }
    stamp(11);
  }
}

bool res2_stage_ok(const Res2StageArgs& a) {
    if (!a.x || !a.y || !a.t1out || !a.fa0 || !a.ba0 || !a.zeros || a.B <= 0) return false;
    for (int i = 0; i < 3; ++i)
        if (!a.fb[i] || !a.bb[i] || !a.fc[i] || !a.bc[i] || !a.fd[i] || !a.bd[i]) return false;
    return true;
}

void launch_res2_stage(const Res2StageArgs& a, hipStream_t st) {
    const int grid = a.B * 32;                      // one 8 x 16 tile per workgroup
    ConvArgs d{};
    d.B = a.B; d.H = 64; d.W = 64; d.Ho = 64; d.Wo = 64; d.Cin = 64; d.Cout = 256; d.KH = -2; d.KW = -2; d.stride = 1;   // KH = -2: the res2 stage row of the layer report
    void* tok = prof_begin(d, 2, st);
#if IVOSW_ABLATION
    if (a.debug) {
        switch (a.debug) {
            case 1: hipLaunchKernelGGL((res2_stage_kernel<true, 1>), dim3(grid), dim3(512), 0, st, a); break;
            case 2: hipLaunchKernelGGL((res2_stage_kernel<true, 2>), dim3(grid), dim3(512), 0, st, a); break;
            case 3: hipLaunchKernelGGL((res2_stage_kernel<true, 3>), dim3(grid), dim3(512), 0, st, a); break;
            case 4: hipLaunchKernelGGL((res2_stage_kernel<true, 4>), dim3(grid), dim3(512), 0, st, a); break;
            case 5: hipLaunchKernelGGL((res2_stage_kernel<true, 5>), dim3(grid), dim3(512), 0, st, a); break;
            case 6: hipLaunchKernelGGL((res2_stage_kernel<true, 6>), dim3(grid), dim3(512), 0, st, a); break;
            default: hipLaunchKernelGGL((res2_stage_kernel<true, 7>), dim3(grid), dim3(512), 0, st, a); break;
        }
        prof_end(tok, st);
        return;
    }
#endif
    if (a.y_s2) hipLaunchKernelGGL((res2_stage_kernel<true>), dim3(grid), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((res2_stage_kernel<false>), dim3(grid), dim3(512), 0, st, a);
    prof_end(tok, st);
}

}  // namespace ivosw
