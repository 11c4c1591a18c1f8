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
//   B2:    T1C, partial sums [0, 32768) -> T2C [32768, 49152)        C2/D2: T2C, Y [49152, 114688)
#include <type_traits>

#include "conv.h"
#include "mfma_tile.h"

namespace ivosw {

namespace {
constexpr int R2_LDS = 163840;
constexpr int T1R = 144;                             // bytes per padded t1 raster row
constexpr int P_OFF = 0, T1A_OFF = 39936, SB0_OFF = 84288, PD_OFF = 0, T2A_OFF = 30720, Y0_OFF = 63488, T1B_OFF = 129024;
constexpr int T2B_OFF = 0, SB1_OFF = 24576, Y1_OFF = 24576, T1C_OFF = 122880;
constexpr int SB2_OFF = 0, T2C_OFF = 32768, Y2_OFF = 49152;
static_assert(T1A_OFF + 308 * T1R <= SB0_OFF && SB0_OFF + 65536 <= R2_LDS, "A0/B0 map");
static_assert(PD_OFF + 240 * ROWB <= T2A_OFF && T2A_OFF + 256 * ROWB <= Y0_OFF && Y0_OFF + 65536 <= T1B_OFF && T1B_OFF + 240 * T1R <= R2_LDS, "C0/D0 map");
static_assert(T2B_OFF + 192 * ROWB <= SB1_OFF && SB1_OFF + 49152 <= T1B_OFF && Y1_OFF + 98304 <= T1C_OFF && T1C_OFF + 180 * T1R <= R2_LDS, "b1 map");
static_assert(SB2_OFF + 32768 <= T2C_OFF && T2C_OFF + 128 * ROWB <= Y2_OFF && Y2_OFF + 65536 <= T1C_OFF, "b2 map");

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

// Weights and biases of a phase are REQUESTED ONE PHASE AHEAD (plain loads pinned by a scheduling fence: hipcc otherwise sinks
// them to their first use) — the first timeline of this kernel showed every phase opening with an exposed L2 round trip
// (~1-2 k cycles of a 3-6 k cycle phase) and the 3x3's three-k-step weight ring stalling on every refill.
struct WB { uint4 w[18]; float4 b[4]; };             // phase B: one K half (18 k-steps) of one channel tile + bias
struct WC { uint4 w[8]; float4 b[4]; };              // phase C: one channel tile (4 k-steps, 8 with the downsample) + bias
struct WD { uint4 w[16]; float4 b[4]; };             // phase D: one output-channel tile, K = 256 + bias
__device__ __forceinline__ void pin() { __builtin_amdgcn_sched_barrier(0); }
// DBG (probe builds only): 1 = no weight loads (zeros), 2 = pixel fragments read once per phase, 4 = no MFMAs
template <int DBG>
__device__ __forceinline__ void load_wb(WB& o, const void* fb, const float* bb, int ct, int kh, int lane) {
#pragma unroll
    for (int j = 0; j < 18; ++j) o.w[j] = (DBG & 1) ? make_uint4(0, 0, 0, 0) : *wfr(fb, ct, 36, kh * 18 + j, lane);
#pragma unroll
    for (int g = 0; g < 4; ++g) o.b[g] = *reinterpret_cast<const float4*>(bb + ct * 32 + 8 * g + 4 * (lane >> 5));
    pin();
}
template <int KS, int DBG>
__device__ __forceinline__ void load_wc(WC& o, const void* fc, const float* bc, int ct, int lane) {
#pragma unroll
    for (int j = 0; j < KS; ++j) o.w[j] = (DBG & 1) ? make_uint4(0, 0, 0, 0) : *wfr(fc, ct, KS, j, lane);
#pragma unroll
    for (int g = 0; g < 4; ++g) o.b[g] = *reinterpret_cast<const float4*>(bc + ct * 32 + 8 * g + 4 * (lane >> 5));
    pin();
}
template <int NKS, int DBG>
__device__ __forceinline__ void load_wd(WD& o, const void* fd, const float* bd, int ct, int ks0, int lane) {
#pragma unroll
    for (int j = 0; j < NKS; ++j) o.w[j] = (DBG & 1) ? make_uint4(0, 0, 0, 0) : *wfr(fd, ct, 16, ks0 + j, lane);
#pragma unroll
    for (int g = 0; g < 4; ++g) o.b[g] = *reinterpret_cast<const float4*>(bd + ct * 32 + 8 * g + 4 * (lane >> 5));
    pin();
}

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
template <int NPT, int SRCW, int DBG, typename F>
__device__ __forceinline__ void phase_b(unsigned char* lds, unsigned lds_base, int T1_OFF, int T2_OFF, int SB_OFF, const WB& wb, int wave,
                                        int lane, F&& ahead) {
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
        slot_pos((q + 2 * i) * 32 + lrow, dy, dx);
        rb[i] = lds_base + T1_OFF + ((dy + OFS) * SRCW + dx + OFS) * T1R + lhalf * 16;
    }
    auto run_half = [&](auto khc) {
        constexpr int K0 = decltype(khc)::value * 18;
        u32x4 pf[2][NT];
        auto rd = [&](auto kc, int buf) {
            constexpr int kstep = K0 + decltype(kc)::value;
            constexpr int tap = kstep >> 2, ks = kstep & 3, toff = (tap / 3) * SRCW + (tap % 3);
#pragma unroll
            for (int i = 0; i < NT; ++i) pf[buf][i] = lds_read_b128_o<toff * T1R + ks * 32>(rb[i]);
        };
        rd(std::integral_constant<int, 0>{}, 0);
        static_for<0, 18>([&](auto kc) {
            constexpr int KS = decltype(kc)::value;
            if constexpr (KS < 17 && !(DBG & 2)) rd(std::integral_constant<int, KS + 1>{}, (KS + 1) & 1);
            if (DBG & 2) { if (KS == 0) lgkm<0>(); } else if (KS < 17) lgkm<NT>(); else lgkm<0>();
            const u32x4 w = u4(wb.w[KS]);
#pragma unroll
            for (int i = 0; i < NT; ++i) acc[i] = mm<DBG>(w, pf[(DBG & 2) ? 0 : (KS & 1)][i], acc[i]);
        });
    };
    if (kh) run_half(std::integral_constant<int, 1>{}); else run_half(std::integral_constant<int, 0>{});
    ahead();
    // exchange: the lower K half finishes the even tiles of the pair, the upper half the odd ones
    float* scr = reinterpret_cast<float*>(lds + SB_OFF + (wave & 3) * NT * 4096);
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
                pk.x = pack2_bf16(fmaxf(acc[i][4 * g] + o.x + wb.b[g].x, 0.f), fmaxf(acc[i][4 * g + 1] + o.y + wb.b[g].y, 0.f));
                pk.y = pack2_bf16(fmaxf(acc[i][4 * g + 2] + o.z + wb.b[g].z, 0.f), fmaxf(acc[i][4 * g + 3] + o.w + wb.b[g].w, 0.f));
                lds_write_b64(lds_base + T2_OFF + row * ROWB + (((ct * 4 + g) ^ ((row >> 1) & 7)) << 4) + 8 * lhalf, pk);
            }
        }
}

// ---------------------------------------------------------------- phase C for one channel tile x NTC pixel tiles (parity h)
// acc = bias + W3[ct] t2 (4 k-steps) [+ Wd[ct] p (4 k-steps, DS)]; the caller finishes (residual, ReLU, stores)
template <int NTC, bool DS, int DBG>
__device__ __forceinline__ void phase_c_mma(f32x16 (&acc)[NTC], unsigned lds_base, int T2_OFF, const WC& wc, int h, int lane) {
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
    rd(0, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        if (!(DBG & 2)) { if (ks < 3) rd(ks + 1, (ks + 1) & 1); else if (DS) rdp(0, 0); }
        if (DBG & 2) lgkm<0>(); else if (ks < 3 || DS) lgkm<NTC>(); else lgkm<0>();
        const u32x4 ww = u4(wc.w[ks]);
#pragma unroll
        for (int i = 0; i < NTC; ++i) acc[i] = mm<DBG>(ww, pf[(DBG & 2) ? 0 : (ks & 1)][i], acc[i]);
    }
    if constexpr (DS) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < 3 && !(DBG & 2)) rdp(ks + 1, (ks + 1) & 1);
            if (DBG & 2) lgkm<0>(); else if (ks < 3) lgkm<NTC>(); else lgkm<0>();
            const u32x4 ww = u4(wc.w[4 + ks]);
#pragma unroll
            for (int i = 0; i < NTC; ++i) acc[i] = mm<DBG>(ww, pf[(DBG & 2) ? 0 : (ks & 1)][i], acc[i]);
        }
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
        pk.x = pack2_bf16(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f));
        pk.y = pack2_bf16(fmaxf(v[2], 0.f), fmaxf(v[3], 0.f));
        out[2 * g] = pk.x; out[2 * g + 1] = pk.y;
        lds_write_b64(ya + (((cb + g) ^ key) << 4), pk);
    }
}

// ---------------------------------------------------------------- phase D k-steps: dacc[i] += w[k] y[tile i], k < NKS
// y image: slices of 64 channels, NROWS rows each; tiles of a wave are PSTRIDE tiles apart (rows PSTRIDE * 32 apart: same key)
template <int NKS, int NROWS, int PSTRIDE, int DBG>
__device__ __forceinline__ void phase_d_mma(f32x16 (&dacc)[2], bool two, unsigned lds_base, int Y_OFF, const WD& wd, int row0, int lane) {
    const int lhalf = lane >> 5;
    const unsigned yrow = lds_base + Y_OFF + row0 * ROWB;
    const int key = (row0 >> 1) & 7;
    u32x4 pd[2][4][2];
    auto rd = [&](auto slc, int buf) {
        constexpr int SL = decltype(slc)::value;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            constexpr int OB = SL * NROWS * ROWB, FAR = OB + PSTRIDE * 32 * ROWB >= 65536 ? OB : 0;   // ds offsets are 16 bits
            const unsigned a = yrow + (((2 * k + lhalf) ^ key) << 4) + FAR;
            pd[buf][k][0] = lds_read_b128_o<OB - FAR>(a);
            if (two) pd[buf][k][1] = lds_read_b128_o<OB - FAR + PSTRIDE * 32 * ROWB>(a);
        }
    };
    rd(std::integral_constant<int, 0>{}, 0);
    static_for<0, NKS / 4>([&](auto slc) {
        constexpr int SL = decltype(slc)::value;
        if constexpr (SL + 1 < NKS / 4 && !(DBG & 2)) rd(std::integral_constant<int, SL + 1>{}, (SL + 1) & 1);
        if (DBG & 2) lgkm<0>(); else if (SL + 1 < NKS / 4) { if (two) lgkm<8>(); else lgkm<4>(); } else lgkm<0>();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const u32x4 ww = u4(wd.w[SL * 4 + k]);
            dacc[0] = mm<DBG>(ww, pd[(DBG & 2) ? 0 : (SL & 1)][k][0], dacc[0]);
            if (two) dacc[1] = mm<DBG>(ww, pd[(DBG & 2) ? 0 : (SL & 1)][k][1], dacc[1]);
        }
    });
}

// where a slot's t1 goes in a raster of width RW whose origin sits OFS pixels up-left of the tile origin: LDS byte address of the
// row (0xffffffff: unused slot) and whether the pixel lies inside the frame (else the 3x3's zero padding is stored)
template <int RW, int OFS>
__device__ __forceinline__ void t1_target(int slot, int y0, int x0, unsigned base, unsigned& addr, bool& in) {
    int dy, dx;
    slot_pos(slot, dy, dx);
    const int y = y0 + dy, x = x0 + dx;
    in = y >= 0 && y < 64 && x >= 0 && x < 64;
    addr = slot_used(slot) ? base + ((dy + OFS) * RW + dx + OFS) * T1R : 0xffffffffu;
}
__device__ __forceinline__ void store_t1(const f32x16& d, const float4 (&b)[4], unsigned addr, bool in, int ct, int lhalf) {
    if (addr == 0xffffffffu) return;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        u32x2 pk;
        pk.x = in ? pack2_bf16(fmaxf(d[4 * g] + b[g].x, 0.f), fmaxf(d[4 * g + 1] + b[g].y, 0.f)) : 0u;
        pk.y = in ? pack2_bf16(fmaxf(d[4 * g + 2] + b[g].z, 0.f), fmaxf(d[4 * g + 3] + b[g].w, 0.f)) : 0u;
        lds_write_b64(addr + ((ct * 4 + g) << 4) + 8 * lhalf, pk);
    }
}
}  // namespace

template <bool YS2, int DBG = 0>
__global__ __launch_bounds__(512, 2) void res2_stage_kernel(Res2StageArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[R2_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhalf = lane >> 5;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int b = L >> 5, tl = L & 31;
    const int y0 = (tl >> 2) * 8, x0 = (tl & 3) * 16;
    const bf16_t* X = static_cast<const bf16_t*>(p.x) + (size_t)b * 64 * 64 * 64;
    const bf16_t* zeros = static_cast<const bf16_t*>(p.zeros);
    auto stamp = [&](int k) {
        if (p.ts && tid == 0) p.ts[(size_t)blockIdx.x * 16 + k] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);
    const int bct = wave & 1, bkh = wave >> 2;       // phase B: channel tile, K half
    const int cc = wave & 3, hh = wave >> 2;         // phase C: channel tiles cc, cc + 4; pixel tiles hh, hh + 2, ...
    const int dct = wave & 1, dv = wave >> 1;        // phase D0 / D1: channel tile, pixel tiles dv, dv + 4

    // ================================================================ p halo in: 14 x 22 raster, 39 groups of 8 rows
    {
        const int rsub = lane >> 3, cpos = lane & 7;
        for (int g = wave; g < 39; g += 8) {
            const int row = g * 8 + rsub;
            const int hy = row / 22, hx = row - hy * 22;
            const int y = y0 - 3 + hy, x = x0 - 3 + hx;
            const bool ok = row < 308 && y >= 0 && y < 64 && x >= 0 && x < 64;
            const bf16_t* src = ok ? X + ((size_t)y * 64 + x) * 64 + (cpos ^ ((row >> 1) & 7)) * 8 : zeros;
            dma16(src, lds + P_OFF + g * 1024);
        }
    }
    WB wb;
    // ================================================================ A0: t1_0 = relu(W1 p + b) on the 14 x 22 raster (10 pixel tiles)
    {
        const int ct = wave & 1, ga = wave >> 1;
        const bool three = ga < 2;                   // tiles ga, ga + 4, ga + 8 (< 10)
        uint4 w[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) w[ks] = *wfr(p.fa0, ct, 4, ks, lane);
        float4 bq[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const float4*>(p.ba0 + ct * 32 + 8 * g + 4 * lhalf);
        pin();
        load_wb<DBG>(wb, p.fb[0], p.bb[0], bct, bkh, lane);
        f32x16 acc[3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stamp(1);
        const int row0 = ga * 32 + lrow;             // tiles 4 apart: rows 128 apart, same key
        const unsigned prow = lds_base + P_OFF + row0 * ROWB;
        const int key = (row0 >> 1) & 7;
        u32x4 pf[2][3];
        auto rd = [&](int ks, int buf) {
            const unsigned a = prow + (((2 * ks + lhalf) ^ key) << 4);
            pf[buf][0] = lds_read_b128_o<0>(a);
            pf[buf][1] = lds_read_b128_o<128 * ROWB>(a);
            if (three) pf[buf][2] = lds_read_b128_o<256 * ROWB>(a);
        };
        rd(0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < 3) rd(ks + 1, (ks + 1) & 1);
            if (ks < 3) { if (three) lgkm<3>(); else lgkm<2>(); } else lgkm<0>();
            const u32x4 ww = u4(w[ks]);
            acc[0] = mm<DBG>(ww, pf[ks & 1][0], acc[0]);
            acc[1] = mm<DBG>(ww, pf[ks & 1][1], acc[1]);
            if (three) acc[2] = mm<DBG>(ww, pf[ks & 1][2], acc[2]);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (i == 2 && !three) break;
            const int hr = (ga + 4 * i) * 32 + lrow;
            const int hy = hr / 22, hx = hr - hy * 22;
            const int y = y0 - 3 + hy, x = x0 - 3 + hx;
            const bool in = y >= 0 && y < 64 && x >= 0 && x < 64;
            if (hr < 308) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2 pk;
                    pk.x = in ? pack2_bf16(fmaxf(acc[i][4 * g] + bq[g].x, 0.f), fmaxf(acc[i][4 * g + 1] + bq[g].y, 0.f)) : 0u;
                    pk.y = in ? pack2_bf16(fmaxf(acc[i][4 * g + 2] + bq[g].z, 0.f), fmaxf(acc[i][4 * g + 3] + bq[g].w, 0.f)) : 0u;
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
    WC wc[2];
    WD wd;
    phase_b<8, 22, DBG>(lds, lds_base, T1A_OFF, T2A_OFF, SB0_OFF, wb, wave, lane, [&] { load_wc<8, DBG>(wc[0], p.fc[0], p.bc[0], cc, lane); });
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
        unsigned ta[2];
        bool tin[2];
        t1_target<20, 2>(dv * 32 + lrow, y0, x0, lds_base + T1B_OFF, ta[0], tin[0]);
        t1_target<20, 2>((dv + 4) * 32 + lrow, y0, x0, lds_base + T1B_OFF, ta[1], tin[1]);
        static_for<0, 2>([&](auto rc_) {
            constexpr int R = decltype(rc_)::value;
            {
                f32x16 acc[4];
                phase_c_mma<4, true, DBG>(acc, lds_base, T2A_OFF, wc[R], hh, lane);
                load_wd<8, DBG>(wd, p.fd[0], p.bd[0], dct, 8 * R, lane);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    unsigned out[8];
                    finish_c<false>(acc[i], none, out, lds_base, Y0_OFF, 256 * ROWB, cc >> 1, (cc & 1) * 4, (hh + 2 * i) * 32 + lrow, lhalf);
                    if (i < 3) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) yres[R][i][k] = out[k];
                    }
                }
            }
            wg_barrier();                            // y0 K half R is in the image
            phase_d_mma<8, 256, 4, DBG>(dacc, true, lds_base, Y0_OFF, wd, dv * 32 + lrow, lane);
            if (R == 0) {
                load_wc<8, DBG>(wc[1], p.fc[0], p.bc[0], 4 + cc, lane);
                wg_barrier();                        // image free for the second half
            }
        });
        stamp(4);
        load_wb<DBG>(wb, p.fb[1], p.bb[1], bct, bkh, lane);
        store_t1(dacc[0], wd.b, ta[0], tin[0], dct, lhalf);
        store_t1(dacc[1], wd.b, ta[1], tin[1], dct, lhalf);
        wg_barrier();                                // t1_1 complete; y image, t2_0, PD dead
        stamp(5);
    }
    // ================================================================ block 1
    phase_b<6, 20, DBG>(lds, lds_base, T1B_OFF, T2B_OFF, SB1_OFF, wb, wave, lane, [&] {
        load_wc<4, DBG>(wc[0], p.fc[1], p.bc[1], cc, lane);
        load_wc<4, DBG>(wc[1], p.fc[1], p.bc[1], 4 + cc, lane);
    });
    wg_barrier();
    stamp(6);
    unsigned y1res[2][2][8];
    static_for<0, 2>([&](auto rc_) {
        constexpr int R = decltype(rc_)::value;
        f32x16 acc[3];
        phase_c_mma<3, false, DBG>(acc, lds_base, T2B_OFF, wc[R], hh, lane);
        if (R == 1) load_wd<16, DBG>(wd, p.fd[1], p.bd[1], dct, 0, lane);
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
    wg_barrier();                                    // y1 image complete
    stamp(7);
    {
        const bool two = dv < 2;                     // D1: pixel tiles dv, dv + 4 (< 6)
        f32x16 dacc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) dacc[i][r] = 0.f;
        unsigned ta[2];
        bool tin[2];
        t1_target<18, 1>(dv * 32 + lrow, y0, x0, lds_base + T1C_OFF, ta[0], tin[0]);
        t1_target<18, 1>((dv + 4) * 32 + lrow, y0, x0, lds_base + T1C_OFF, ta[1], tin[1]);
        phase_d_mma<16, 192, 4, DBG>(dacc, two, lds_base, Y1_OFF, wd, dv * 32 + lrow, lane);
        load_wb<DBG>(wb, p.fb[2], p.bb[2], bct, bkh, lane);
        store_t1(dacc[0], wd.b, ta[0], tin[0], dct, lhalf);
        if (two) store_t1(dacc[1], wd.b, ta[1], tin[1], dct, lhalf);
        wg_barrier();                                // t1_2 complete; y image, t2_1 dead
        stamp(8);
    }
    // ================================================================ block 2
    phase_b<4, 18, DBG>(lds, lds_base, T1C_OFF, T2C_OFF, SB2_OFF, wb, wave, lane, [&] {
        load_wc<4, DBG>(wc[0], p.fc[2], p.bc[2], cc, lane);
        load_wc<4, DBG>(wc[1], p.fc[2], p.bc[2], 4 + cc, lane);
    });
    wg_barrier();
    stamp(9);
    const int d2ct = wave & 3, d2v = wave >> 2;      // D2: output-channel tile (of 4), pixel tiles d2v, d2v + 2
    static_for<0, 2>([&](auto rc_) {
        constexpr int R = decltype(rc_)::value;
        f32x16 acc[2];
        phase_c_mma<2, false, DBG>(acc, lds_base, T2C_OFF, wc[R], hh, lane);
        if (R == 1) load_wd<16, DBG>(wd, p.fd[2], p.bd[2], d2ct, 0, lane);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            unsigned out[8];
            finish_c<true>(acc[i], y1res[R][i], out, lds_base, Y2_OFF, 128 * ROWB, (4 * R + cc) >> 1, (cc & 1) * 4, (hh + 2 * i) * 32 + lrow, lhalf);
        }
    });
    wg_barrier();                                    // y2 image complete
    stamp(10);
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
            v[j] = lds_read_b128(lds_base + Y2_OFF + (c16 >> 3) * 128 * ROWB + slot * ROWB + (((c16 & 7) ^ ((slot >> 1) & 7)) << 4));
        }
        lds_wait();
#pragma unroll
        for (int j = 0; j < NJ; ++j) *reinterpret_cast<uint4*>(dst[j]) = make_uint4(v[j][0], v[j][1], v[j][2], v[j][3]);
    }
    // D2: res3's conv1 (256 -> 128) on the 8 x 16 tile -> HBM
    {
        f32x16 dacc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) dacc[i][r] = 0.f;
        phase_d_mma<16, 128, 2, DBG>(dacc, true, lds_base, Y2_OFF, wd, d2v * 32 + lrow, lane);
        bf16_t* T1O = static_cast<bf16_t*>(p.t1out) + (size_t)b * 64 * 64 * 128;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = (d2v + 2 * i) * 32 + lrow;
            bf16_t* o = T1O + ((size_t)(y0 + (q >> 4)) * 64 + x0 + (q & 15)) * 128 + d2ct * 32 + 4 * lhalf;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 pk;
                pk.x = pack2_bf16(fmaxf(dacc[i][4 * g] + wd.b[g].x, 0.f), fmaxf(dacc[i][4 * g + 1] + wd.b[g].y, 0.f));
                pk.y = pack2_bf16(fmaxf(dacc[i][4 * g + 2] + wd.b[g].z, 0.f), fmaxf(dacc[i][4 * g + 3] + wd.b[g].w, 0.f));
                *reinterpret_cast<uint2*>(o + 8 * g) = pk;
            }
        }
    }
    stamp(11);
}

bool res2_stage_ok(const Res2StageArgs& a) {
    if (!a.x || !a.y || !a.t1out || !a.fa0 || !a.ba0 || !a.zeros || a.B <= 0) return false;
    for (int i = 0; i < 3; ++i)
        if (!a.fb[i] || !a.bb[i] || !a.fc[i] || !a.bc[i] || !a.fd[i] || !a.bd[i]) return false;
    return true;
}

void launch_res2_stage(const Res2StageArgs& a, hipStream_t st) {
    ConvArgs d{};
    d.B = a.B; d.H = 64; d.W = 64; d.Ho = 64; d.Wo = 64; d.Cin = 64; d.Cout = 256; d.KH = -2; d.KW = -2; d.stride = 1;   // KH = -2: the res2 stage row of the layer report
    void* tok = prof_begin(d, 2, st);
#if IVOSW_ABLATION
    if (a.debug) {
        switch (a.debug) {
            case 1: hipLaunchKernelGGL((res2_stage_kernel<true, 1>), dim3(a.B * 32), dim3(512), 0, st, a); break;
            case 2: hipLaunchKernelGGL((res2_stage_kernel<true, 2>), dim3(a.B * 32), dim3(512), 0, st, a); break;
            case 3: hipLaunchKernelGGL((res2_stage_kernel<true, 3>), dim3(a.B * 32), dim3(512), 0, st, a); break;
            case 4: hipLaunchKernelGGL((res2_stage_kernel<true, 4>), dim3(a.B * 32), dim3(512), 0, st, a); break;
            case 5: hipLaunchKernelGGL((res2_stage_kernel<true, 5>), dim3(a.B * 32), dim3(512), 0, st, a); break;
            case 6: hipLaunchKernelGGL((res2_stage_kernel<true, 6>), dim3(a.B * 32), dim3(512), 0, st, a); break;
            default: hipLaunchKernelGGL((res2_stage_kernel<true, 7>), dim3(a.B * 32), dim3(512), 0, st, a); break;
        }
        prof_end(tok, st);
        return;
    }
#endif
    if (a.y_s2) hipLaunchKernelGGL((res2_stage_kernel<true>), dim3(a.B * 32), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((res2_stage_kernel<false>), dim3(a.B * 32), dim3(512), 0, st, a);
    prof_end(tok, st);
}

}  // namespace ivosw
