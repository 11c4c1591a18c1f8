// Fused stem of the assessment encoder (bf16 throughput mode): 7x7/2 convolution over the 4-channel ROI tile
// (R,G,B normalised | P), folded BN, ReLU and the 3x3/2 max-pool in ONE kernel.
//
// Reference arithmetic: Encoder.forward, models/assessment.py:54-57
//     x = conv1(f) + conv1_p(p);  x = bn1(x);  c1 = relu(x);  x = maxpool(c1)
// (conv1 | conv1_p are concatenated along Cin and the BN scale is folded into the weights by pack_stem_kernel).
//
// Why: layer by layer the 128x128x64 stem output (2 MB per frame in bf16) is written to HBM and read back by the
// pool, and the generic implicit-GEMM kernel spends most of its time gathering 8-byte pixels (225 TFLOP/s).  Here a
// workgroup owns an 8x8 tile of POOLED outputs: the 39x39-pixel input patch (12 KB) and the whole 64x7x8x4 filter
// bank (28 KB) sit in LDS, im2col is pure fragment addressing (a lane's 8 K-elements = 2 neighbouring pixels x 4
// channels = 16 contiguous bytes of the patch), the 17x17 conv outputs the tile needs go to LDS as bf16 and are
// pooled from there.  Only the ROI tile comes in (0.5 MB per frame) and the pooled map goes out (0.5 MB per frame).
//
// MFMA view ("transposed": A = filters, B = pixels, so a lane holds 4 consecutive channels of one pixel):
//   channels 64 = 2 tiles, conv pixels 289 -> 10 tiles of 32, K = 7 filter rows x 8 taps x 4 channels = 14 steps of 16
//   (tap 7 of every filter row carries zero weights; the patch has a 40th zero column for it).
#include <type_traits>

#include "conv.h"
#include "mfma_tile.h"

namespace ivosw {

namespace {
constexpr int PT = 8;                         // pooled tile edge
constexpr int CT = 2 * PT + 1;                // 17 conv outputs per edge
constexpr int NPIX = CT * CT;                 // 289
constexpr int PR = 2 * (CT - 1) + 7;          // 39 input rows / cols
constexpr int PW = 40;                        // patch row stride in pixels (col 39 = zeros for the padding tap)
constexpr int PATCH_OFF = 0, PATCH_BYTES = 12544;           // 39 * 320 = 12480, padded
constexpr int WL_OFF = PATCH_OFF + PATCH_BYTES, WL_BYTES = 7 * 4 * 64 * 16;   // 28672
constexpr int CO_OFF = WL_OFF + WL_BYTES, CO_BYTES = NPIX * 128;             // 36992
constexpr int STEM_LDS = CO_OFF + CO_BYTES;                                   // 78208: two workgroups per CU
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
}  // namespace

// Persistent: each workgroup loads the filter bank once and walks over tiles (tile id = first + k * stride); the next
// tile's input patch is fetched into registers while the current tile is in the matrix cores.
__global__ __launch_bounds__(256, 2) void stem_pool_kernel(const bf16_t* __restrict__ roi, const bf16_t* __restrict__ wpk,
                                                        const float* __restrict__ bias, bf16_t* __restrict__ out, int ntiles, int rev) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[STEM_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, kh = lane >> 5;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    // workgroups of one XCD (blockIdx % 8) take neighbouring tiles, so the 7-pixel halos are shared in that XCD's L2
    const int per_xcd = gridDim.x >> 3;
    const int first = (gridDim.x & 7) ? blockIdx.x : (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);

    // patch entry j of this thread: pixel (r, c) of the 39 x 40 patch
    constexpr int NPRE = (PR * PW + 255) / 256;   // 7
    int pr_[NPRE], pc_[NPRE];
#pragma unroll
    for (int j = 0; j < NPRE; ++j) {
        const int i = tid + 256 * j;
        pr_[j] = i / PW;
        pc_[j] = i - pr_[j] * PW;
    }
    uint2 pre[NPRE];
    auto patch_fetch = [&](int t_) {             // tile t -> registers (zero outside the 256 x 256 ROI tile and in col 39)
        const int t = rev ? ntiles - 1 - t_ : t_;    // (rev: the tiles in descending order, see BneckWideArgs::rev)
        const int b = t >> 6, tl = t & 63;
        const bf16_t* img = roi + (size_t)b * 256 * 256 * 4;
        const int iy0 = 4 * (tl >> 3) * PT - 5, ix0 = 4 * (tl & 7) * PT - 5;
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            const int iy = iy0 + pr_[j], ix = ix0 + pc_[j];
            const bool ok = pr_[j] < PR && pc_[j] < PR && iy >= 0 && iy < 256 && ix >= 0 && ix < 256;
            const uint2* src = reinterpret_cast<const uint2*>(ok ? img + ((size_t)iy * 256 + ix) * 4 : wpk);   // always a valid address
            const uint2 v = *src;
            pre[j] = ok ? v : make_uint2(0, 0);
        }
    };
    auto patch_store = [&]() {
#pragma unroll
        for (int j = 0; j < NPRE; ++j)
            if (tid + 256 * j < PR * PW) *reinterpret_cast<uint2*>(lds + PATCH_OFF + (tid + 256 * j) * 8) = pre[j];
    };

    int t = first;
    if (t < ntiles) patch_fetch(t);
    // filters: packed [64 ch][8 ky][8 taps][4] -> LDS [ky][tap pair q][ch][16 B] (consecutive channels 16 B apart)
    for (int i = tid; i < 7 * 4 * 64; i += 256) {
        const int ch = i & 63, kq = i >> 6, ky = kq >> 2, q = kq & 3;
        *reinterpret_cast<uint4*>(lds + WL_OFF + i * 16) = *reinterpret_cast<const uint4*>(wpk + ((ch * 8 + ky) * 8 + 2 * q) * 4);
    }
    const int ct = wave & 1, pq = wave >> 1;      // conv: wave = (channel tile ct, pixel tiles pq, pq+2, .., pq+8)
    // Pixel tile T of the 17 x 17 conv outputs (the MFMA's 32 pixel columns may be ANY 32 pixels): T < 8 is the 4-row x 8-column
    // block (T >> 1, T & 1) of the 16 x 16 core, tile 8 is row 16 (17 pixels), tile 9 is column 16 (rows 0..15).  A conv row is
    // PW = 40 sixteen-byte chunks of the patch, 40 = 8 (mod 16): the four 8-pixel row segments a 16-lane group of a ds_read_b128
    // touches then fall on four different quarters of the 64 banks - conflict-free.  (32 CONSECUTIVE pixels of the 17-wide
    // map wrap to the next row somewhere inside almost every lane group and read two chunks of the same banks.  rocprofv3, per
    // 128 frames: SQ_LDS_BANK_CONFLICT 15.8 M -> 10.8 M cycles, SQ_LDS_IDX_ACTIVE 32.7 M -> 27.7 M; the launch time did not move
    // (164 us per 256 frames either way): the kernel is bound by its ~820 non-MFMA vector instructions per tile, not by the LDS.)
    auto tile_px = [&](int T, int& cy, int& cx) -> bool {
        if (T < 8) { cy = 4 * (T >> 1) + (lrow >> 3); cx = 8 * (T & 1) + (lrow & 7); return true; }
        if (T == 8) { cy = 16; cx = min(lrow, 16); return lrow <= 16; }
        cy = min(lrow, 15); cx = 16; return lrow < 16;
    };
    unsigned pb[5];                               // LDS address of this lane's pixel pair at (ky, tap half) = (0, 0)
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        int cy, cx;
        (void)tile_px(pq + 2 * i, cy, cx);       // padding lanes read a valid pixel, their results are dropped
        pb[i] = lds_base + PATCH_OFF + ((2 * cy) * PW + 2 * cx + 2 * kh) * 8;
    }
    const unsigned wb = lds_base + WL_OFF + (kh * 64 + ct * 32 + lrow) * 16;
    float4 bq[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const float4*>(bias + ct * 32 + 8 * g + 4 * kh);

    for (; t < ntiles; t += gridDim.x) {
        const int tt = rev ? ntiles - 1 - t : t;
        const int b = tt >> 6, tl = tt & 63;
        const int py0 = (tl >> 3) * PT, px0 = (tl & 7) * PT;
        patch_store();                            // (the previous tile's pooling is behind the barrier at the loop end)
        __syncthreads();
        asm volatile("" ::: "memory");
        if (t + (int)gridDim.x < ntiles) patch_fetch(t + gridDim.x);   // in flight under the MFMAs

        f32x16 acc[5];
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {            // accumulators start at the bias: no add in the epilogue
                acc[i][4 * g] = bq[g].x; acc[i][4 * g + 1] = bq[g].y; acc[i][4 * g + 2] = bq[g].z; acc[i][4 * g + 3] = bq[g].w;
            }
        {
            u32x4 wf[2], pf[2][5];
            // step S: filter row S >> 1, taps 4*(S&1) .. +3; the per-step offsets are immediates of the ds_read (the kernel is
            // VALU-bound: ~820 non-MFMA vector instructions per tile against 70 MFMAs, a fifth of them address adds)
            auto frag_read = [&](auto sc, int buf) {
                constexpr int S = decltype(sc)::value, ky = S >> 1, h2 = S & 1;
                wf[buf] = lds_read_b128_o<((ky * 4 + 2 * h2) * 64) * 16>(wb);
#pragma unroll
                for (int i = 0; i < 5; ++i) pf[buf][i] = lds_read_b128_o<ky * (PW * 8) + h2 * 32>(pb[i]);
            };
            auto step = [&](auto sc) {
                constexpr int S = decltype(sc)::value;
                lds_wait();
                if constexpr (S < 13) frag_read(std::integral_constant<int, S + 1>{}, (S + 1) & 1);
#pragma unroll
                for (int i = 0; i < 5; ++i) acc[i] = mfma_bf16(wf[S & 1], pf[S & 1][i], acc[i]);
            };
            frag_read(std::integral_constant<int, 0>{}, 0);
            step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
            step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
            step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
            step(std::integral_constant<int, 9>{}); step(std::integral_constant<int, 10>{}); step(std::integral_constant<int, 11>{});
            step(std::integral_constant<int, 12>{}); step(std::integral_constant<int, 13>{});
        }
        // + bias, ReLU, zero outside the 128x128 conv map (pool padding: every window holds a valid value >= 0)
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            int cyl, cxl;
            if (tile_px(pq + 2 * i, cyl, cxl)) {
                const int p = cyl * CT + cxl;
                const int cy = 2 * py0 - 1 + cyl, cx = 2 * px0 - 1 + cxl;
                const unsigned in = (cy >= 0 && cy < 128 && cx >= 0 && cx < 128) ? 0xffffffffu : 0u;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    // ReLU as ONE packed-int16 max AFTER the bf16 conversion (relu2_bf16: the same bits as max-then-round, a quarter
                    // of the v_max), the map border as an AND
                    uint2 pk;
                    pk.x = relu2_bf16(acc[i][4 * g], acc[i][4 * g + 1]) & in;
                    pk.y = relu2_bf16(acc[i][4 * g + 2], acc[i][4 * g + 3]) & in;
                    *reinterpret_cast<uint2*>(lds + CO_OFF + p * 128 + (((ct * 4 + g) ^ (p & 7)) << 4) + 8 * kh) = pk;
                }
            }
        }
        __syncthreads();                          // conv tile complete; every wave is done reading the patch
        // 3x3/2 max-pool out of LDS; non-negative bf16 order like their bit patterns: packed u16 max
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int item = it * 256 + tid;
            const int pp = item >> 3, cg = item & 7;
            const int ppy = pp >> 3, ppx = pp & 7;
            u16x8 m = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int p = (2 * ppy + dy) * CT + 2 * ppx + dx;
                    const u16x8 v = *reinterpret_cast<const u16x8*>(lds + CO_OFF + p * 128 + ((cg ^ (p & 7)) << 4));
                    m = __builtin_elementwise_max(m, v);
                }
            *reinterpret_cast<u16x8*>(out + (((size_t)b * 64 + py0 + ppy) * 64 + px0 + ppx) * 64 + cg * 8) = m;
        }
        // the next iteration's patch_store only touches the patch (free since the barrier above); its epilogue writes
        // the conv tile after ITS first barrier, which every wave reaches only after finishing this pooling pass
    }
}

// ---------------------------------------------------------------- the same for the three-pass mode (IVOSW_F32X3, round 6)
// Layer by layer that mode's stem is the generic implicit-GEMM kernel on the fp32 ROI tile (gathering 16-byte pixels, splitting every
// fragment in every wave that reads it), a 4 MB-per-frame split conv map written and read back, and a pooling launch: 8.4 % of the mode's
// kernel time.  Here: the 39 x 39 patch is split ONCE when it is stored to LDS (hi = truncation, lo = RNE(x - hi): split8_x3's arithmetic),
// the pre-split filter bank (launch_split_weights_x3: one 128-byte [32 hi | 32 lo] group per channel and filter row) sits in LDS as a hi
// and a lo copy, a k-step is three MFMAs per pixel tile in ktile_mma_x3's order (p_hi w_hi, p_hi w_lo, p_lo w_hi), and the conv tile goes
// to LDS as the fp32 value the split activation format would hold (hi + lo of relu(acc + bias)), is pooled there and split once on the way
// out - value for value what conv + maxpool_x3_kernel produce.
// 256 threads = one wave per SIMD, one workgroup per CU (153 KB of LDS); wave = (channel tile, pixel tiles pq, pq + 2, .., pq + 8).
namespace {
constexpr int X_PH_OFF = 0, X_PL_OFF = PATCH_BYTES;                            // patch hi | lo: 8 bytes per pixel each
constexpr int X_WH_OFF = 2 * PATCH_BYTES, X_WL_OFF = X_WH_OFF + WL_BYTES;      // filters hi | lo: [ky][tap pair][ch][16 B]
constexpr int X_CO_OFF = X_WL_OFF + WL_BYTES, X_CO_BYTES = NPIX * 256;         // conv tile: 64 floats per pixel, 16-byte chunks XOR (pixel & 15)
constexpr int STEM_X3_LDS = X_CO_OFF + X_CO_BYTES;                             // 156 416
static_assert(STEM_X3_LDS <= 163840, "LDS budget");
}  // namespace

__global__ __launch_bounds__(256) void stem_pool_x3_kernel(const float* __restrict__ roi, const unsigned char* __restrict__ wsp,
                                                           const float* __restrict__ bias, uint4* __restrict__ out, int ntiles, int rev) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[STEM_X3_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, kh = lane >> 5;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int per_xcd = gridDim.x >> 3;
    const int first = (gridDim.x & 7) ? blockIdx.x : (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);

    constexpr int NPRE = (PR * PW + 255) / 256;   // 7
    int pr_[NPRE], pc_[NPRE];
#pragma unroll
    for (int j = 0; j < NPRE; ++j) {
        const int i = tid + 256 * j;
        pr_[j] = i / PW;
        pc_[j] = i - pr_[j] * PW;
    }
    float4 pre[NPRE];
    auto patch_fetch = [&](int t_) {             // tile t -> registers (zero outside the 256 x 256 ROI tile and in col 39)
        const int t = rev ? ntiles - 1 - t_ : t_;
        const int b = t >> 6, tl = t & 63;
        const float* img = roi + (size_t)b * 256 * 256 * 4;
        const int iy0 = 4 * (tl >> 3) * PT - 5, ix0 = 4 * (tl & 7) * PT - 5;
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            const int iy = iy0 + pr_[j], ix = ix0 + pc_[j];
            const bool ok = pr_[j] < PR && pc_[j] < PR && iy >= 0 && iy < 256 && ix >= 0 && ix < 256;
            const float4 v = *reinterpret_cast<const float4*>(ok ? img + ((size_t)iy * 256 + ix) * 4 : bias);   // always a valid address
            pre[j] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto patch_store = [&]() {                   // the split of split8_x3, once per pixel
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            if (tid + 256 * j < PR * PW) {
                const unsigned x[4] = {__float_as_uint(pre[j].x), __float_as_uint(pre[j].y), __float_as_uint(pre[j].z), __float_as_uint(pre[j].w)};
                uint2 hi, lo;
                hi.x = __builtin_amdgcn_perm(x[1], x[0], 0x07060302u);
                hi.y = __builtin_amdgcn_perm(x[3], x[2], 0x07060302u);
                lo.x = pack2_bf16(pre[j].x - __uint_as_float(x[0] & 0xffff0000u), pre[j].y - __uint_as_float(x[1] & 0xffff0000u));
                lo.y = pack2_bf16(pre[j].z - __uint_as_float(x[2] & 0xffff0000u), pre[j].w - __uint_as_float(x[3] & 0xffff0000u));
                *reinterpret_cast<uint2*>(lds + X_PH_OFF + (tid + 256 * j) * 8) = hi;
                *reinterpret_cast<uint2*>(lds + X_PL_OFF + (tid + 256 * j) * 8) = lo;
            }
        }
    };

    int t = first;
    if (t < ntiles) patch_fetch(t);
    // filters: split arena [64 ch][7 ky][128 B = 32 hi | 32 lo], k = 4 tap + channel  ->  LDS [ky][tap pair q][ch][16 B], hi and lo copies
    for (int i = tid; i < 7 * 4 * 64; i += 256) {
        const int ch = i & 63, kq = i >> 6, ky = kq >> 2, q = kq & 3;
        const uint4* g = reinterpret_cast<const uint4*>(wsp + ((size_t)ch * 7 + ky) * 128);
        *reinterpret_cast<uint4*>(lds + X_WH_OFF + i * 16) = g[q];
        *reinterpret_cast<uint4*>(lds + X_WL_OFF + i * 16) = g[4 + q];
    }
    const int ct = wave & 1, pq = wave >> 1;
    auto tile_px = [&](int T, int& cy, int& cx) -> bool {          // (as stem_pool_kernel)
        if (T < 8) { cy = 4 * (T >> 1) + (lrow >> 3); cx = 8 * (T & 1) + (lrow & 7); return true; }
        if (T == 8) { cy = 16; cx = min(lrow, 16); return lrow <= 16; }
        cy = min(lrow, 15); cx = 16; return lrow < 16;
    };
    unsigned pb[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        int cy, cx;
        (void)tile_px(pq + 2 * i, cy, cx);
        pb[i] = lds_base + X_PH_OFF + ((2 * cy) * PW + 2 * cx + 2 * kh) * 8;
    }
    const unsigned wb = lds_base + X_WH_OFF + (kh * 64 + ct * 32 + lrow) * 16;
    float4 bq[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const float4*>(bias + ct * 32 + 8 * g + 4 * kh);

    for (; t < ntiles; t += gridDim.x) {
        const int tt = rev ? ntiles - 1 - t : t;
        const int b = tt >> 6, tl = tt & 63;
        const int py0 = (tl >> 3) * PT, px0 = (tl & 7) * PT;
        patch_store();
        __syncthreads();
        asm volatile("" ::: "memory");
        if (t + (int)gridDim.x < ntiles) patch_fetch(t + gridDim.x);   // in flight under the MFMAs

        f32x16 acc[5];
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        {
            u32x4 wh[2], wl[2], ph[2][5], pl[2][5];
            auto frag_read = [&](auto sc, int buf) {
                constexpr int S = decltype(sc)::value, ky = S >> 1, h2 = S & 1;
                wh[buf] = lds_read_b128_o<((ky * 4 + 2 * h2) * 64) * 16>(wb);
                wl[buf] = lds_read_b128_o<((ky * 4 + 2 * h2) * 64) * 16 + (X_WL_OFF - X_WH_OFF)>(wb);
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    ph[buf][i] = lds_read_b128_o<ky * (PW * 8) + h2 * 32>(pb[i]);
                    pl[buf][i] = lds_read_b128_o<ky * (PW * 8) + h2 * 32 + (X_PL_OFF - X_PH_OFF)>(pb[i]);
                }
            };
            auto step = [&](auto sc) {
                constexpr int S = decltype(sc)::value;
                lds_wait();
                if constexpr (S < 13) frag_read(std::integral_constant<int, S + 1>{}, (S + 1) & 1);
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    acc[i] = mfma_bf16(wh[S & 1], ph[S & 1][i], acc[i]);
                    acc[i] = mfma_bf16(wl[S & 1], ph[S & 1][i], acc[i]);
                    acc[i] = mfma_bf16(wh[S & 1], pl[S & 1][i], acc[i]);
                }
            };
            frag_read(std::integral_constant<int, 0>{}, 0);
            step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
            step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
            step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
            step(std::integral_constant<int, 9>{}); step(std::integral_constant<int, 10>{}); step(std::integral_constant<int, 11>{});
            step(std::integral_constant<int, 12>{}); step(std::integral_constant<int, 13>{});
        }
        // + bias, ReLU, the value the split format holds (hi + lo), zero outside the 128 x 128 conv map (every pool window holds a valid value >= 0)
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            int cyl, cxl;
            if (tile_px(pq + 2 * i, cyl, cxl)) {
                const int p = cyl * CT + cxl;
                const int cy = 2 * py0 - 1 + cyl, cx = 2 * px0 - 1 + cxl;
                const bool in = cy >= 0 && cy < 128 && cx >= 0 && cx < 128;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float bb[4] = {bq[g].x, bq[g].y, bq[g].z, bq[g].w};
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float r = fmaxf(acc[i][4 * g + e] + bb[e], 0.f);
                        const float h = __uint_as_float(__float_as_uint(r) & 0xffff0000u);
                        const float l = __uint_as_float(pack2_bf16(r - h, 0.f) << 16);
                        v[e] = in ? h + l : 0.f;
                    }
                    const int c16 = ct * 8 + 2 * g + kh;          // channels ct * 32 + 8 g + 4 kh .. + 3
                    *reinterpret_cast<float4*>(lds + X_CO_OFF + p * 256 + ((c16 ^ (p & 15)) << 4)) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
        __syncthreads();                          // conv tile complete; every wave is done reading the patch
        // 3x3/2 max-pool out of LDS, split on the way out: thread = (pooled pixel, 8-channel chunk)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int item = it * 256 + tid;
            const int pp = item >> 3, cg = item & 7;
            const int ppy = pp >> 3, ppx = pp & 7;
            float m[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int p = (2 * ppy + dy) * CT + 2 * ppx + dx;
                    const float4 v0 = *reinterpret_cast<const float4*>(lds + X_CO_OFF + p * 256 + (((2 * cg) ^ (p & 15)) << 4));
                    const float4 v1 = *reinterpret_cast<const float4*>(lds + X_CO_OFF + p * 256 + (((2 * cg + 1) ^ (p & 15)) << 4));
                    m[0] = fmaxf(m[0], v0.x); m[1] = fmaxf(m[1], v0.y); m[2] = fmaxf(m[2], v0.z); m[3] = fmaxf(m[3], v0.w);
                    m[4] = fmaxf(m[4], v1.x); m[5] = fmaxf(m[5], v1.y); m[6] = fmaxf(m[6], v1.z); m[7] = fmaxf(m[7], v1.w);
                }
            uint32_t hi[4], lo[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {              // split8_store_x3 (conv.hip)
                const unsigned a = __float_as_uint(m[2 * q]), bbits = __float_as_uint(m[2 * q + 1]);
                hi[q] = __builtin_amdgcn_perm(bbits, a, 0x07060302u);
                lo[q] = pack2_bf16(m[2 * q] - __uint_as_float(a & 0xffff0000u), m[2 * q + 1] - __uint_as_float(bbits & 0xffff0000u));
            }
            uint4* o = out + (((size_t)b * 64 + py0 + ppy) * 64 + px0 + ppx) * 16 + (cg >> 2) * 8 + (cg & 3);
            o[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            o[4] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        }
    }
}

void launch_stem_pool_x3(const void* roi, const void* wsplit, const float* bias, int B, void* out, hipStream_t st, int rev) {
    ConvArgs d{};
    d.B = B; d.H = 256; d.W = 256; d.Cin = 4; d.Ho = 128; d.Wo = 128; d.Cout = 64; d.KH = 7; d.KW = 7; d.stride = 2;
    void* tok = prof_begin(d, 4, st);
    const int ntiles = B * 64;
    static int ncu = 0;
    if (!ncu) { int dev = 0; hipDeviceProp_t pr; (void)hipGetDevice(&dev); ncu = hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256; }
    const int grid = ntiles < ncu ? ntiles : ncu;               // one workgroup per CU, persistent over the tiles
    hipLaunchKernelGGL(stem_pool_x3_kernel, dim3(grid), dim3(256), 0, st, static_cast<const float*>(roi), static_cast<const unsigned char*>(wsplit),
                       bias, static_cast<uint4*>(out), ntiles, rev);
    prof_end(tok, st);
}

void launch_stem_pool(const void* roi, const void* w, const float* bias, int B, void* out, hipStream_t st, int rev) {
    ConvArgs d{};
    d.B = B; d.H = 256; d.W = 256; d.Cin = 4; d.Ho = 128; d.Wo = 128; d.Cout = 64; d.KH = 7; d.KW = 7; d.stride = 2;
    void* tok = prof_begin(d, 2, st);
    const int ntiles = B * 64;
    const int grid = ntiles < 512 ? ntiles : 512;             // 2 workgroups per CU (78 KB of LDS each), persistent over the tiles
    hipLaunchKernelGGL(stem_pool_kernel, dim3(grid), dim3(256), 0, st, static_cast<const bf16_t*>(roi), static_cast<const bf16_t*>(w),
                       bias, static_cast<bf16_t*>(out), ntiles, rev);
    prof_end(tok, st);
}

}  // namespace ivosw
