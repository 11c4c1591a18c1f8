// Whole-bottleneck fusion for the wide stages (bf16): the identity blocks of res4 (C = 256, 16x16 frames, 1024 -> 256 ->
// 256 (3x3) -> 1024 + residual, one frame per workgroup) and res5 (C = 512, 8x8 frames, two frames per workgroup).
//
// Reference arithmetic: torchvision ResNet-50 v1.5 Bottleneck as Encoder.forward runs it (models/assessment.py:60),
//     out = relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1(x)))))))) + x),  BN folded into weights + bias.
//
// Layer by layer these blocks sit at 2.3 TB/s of HBM and 25 % MFMA busy: each of the three launches is bound by the
// LDS-DMA fill path (weights AND activations stream through one in-order queue per wave and retire at the pace of its
// DRAM misses) with its phases serialised by a barrier per K-tile.  The dataflow here is different from bottleneck.hip:
//
//   * one workgroup = one FRAME (16 x 16 = 256 pixels): no halo, no recompute; the 3x3's zero padding is a lane
//     predicate on the shifted fragment read;
//   * the intermediates t1 and t2 (256 px x 256 ch bf16 = 128 KB) live in LDS as four 64-channel slices of the usual
//     swizzled [row][128 B] image, t2 overwriting t1;
//   * wave w owns output-channel tile w (32 channels) x all 8 pixel tiles (128 accumulator registers) in every
//     contraction, so a weight fragment is needed by exactly ONE wave: weights never touch LDS.  They are stored
//     pre-arranged in fragment order (launch_fragpack: [channel tile][k step][lane][16 B], 1 KB per MFMA operand) and
//     each wave streams its own slice with plain coalesced 16-B loads, two K-tiles ahead of use, with no barrier.
//     Only x (phase A) goes through an LDS-DMA ring; phases B and C run barrier-free.
//   * "transposed" MFMAs (A = weights, B = pixels): a lane holds 4 consecutive channels of one pixel, so t1 / t2 are
//     written with ds_write_b64 and the final tile goes through a 4-KB per-wave staging tile to 64-B row segments.
//
// LDS: [0, 131072) x ring (4 x 32 KB) -> t1 -> t2 ; [131072, 163840) per-wave store staging (8 x 4 KB).
#ifdef IVOSW_PROBES
#include "../../include/ivosw_probe.h"
#endif
#include <type_traits>

#include "conv.h"
#include "gemm_bt.h"
#include "mfma_tile.h"

namespace ivosw {

namespace {
#ifndef HALO_ABL
#define HALO_ABL 0                                   // tuning builds only (tools/build_variant.sh abl3 "-DHALO_ABL=3" bottleneck_wide.hip): the res3 / res2
#endif                                               // halo kernel with 1 no residual loads, 2 no y stores, 4 no x halo DMA (phase cycles: DESIGN 8)
constexpr int WIDE_LDS = 163840;

// one MFMA weight operand of the fragment-ordered copy: channel tile ct, k-step ks of KS per tile
__device__ __forceinline__ const uint4* wfrag(const void* base, int ct, int KS, int ks, int lane) {
    return reinterpret_cast<const uint4*>(static_cast<const char*>(base) + ((size_t)(ct * KS + ks) * 64 + lane) * 16);
}
__device__ __forceinline__ u32x4 as_u32x4(uint4 v) {
    u32x4 r = {v.x, v.y, v.z, v.w};
    return r;
}
template <int N>
__device__ __forceinline__ void lgkm_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
}  // namespace

// C = mid width (256: res4, 512: res5), HW = frame edge (16 / 8), FR = frames per workgroup (1 / 2): C * FR * HW^2 * 2 B
// = 128 KB of t1 / t2 either way, and every wave owns CPW channel tiles x NPT pixel tiles = 8 accumulator tiles.
template <int C, int HW, int FR>
__device__ __forceinline__ void bneck_wide_body(const BneckWideArgs& p, unsigned char* lds) {
    constexpr int CIN = 4 * C, NPX = FR * HW * HW, NPT = NPX / 32, NCT = C / 32, CPW = NCT / 8, NSL = C / 64;
    constexpr int SLICE = NPX * ROWB;                // one 64-channel slice of the pixel image
    constexpr int IMG_BYTES = NSL * SLICE, STG_OFF = IMG_BYTES;
    constexpr int HP = NPT / 2;                      // pixel tiles per rolling half
    constexpr int NACC = CPW * NPT;                  // accumulator tiles per wave: 8 (128 registers), 4 for the one-frame res5 variant
    static_assert((IMG_BYTES == 131072 && NACC == 8) || (IMG_BYTES == 65536 && NACC == 4), "tile geometry");
    static_assert(IMG_BYTES + 8 * 4096 <= WIDE_LDS, "LDS budget");
    static_assert(4 * SLICE <= IMG_BYTES, "x ring: 4 slots");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhalf = lane >> 5;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const bf16_t* X = static_cast<const bf16_t*>(p.x) + (size_t)blockIdx.x * NPX * CIN;
    bf16_t* Y = static_cast<bf16_t*>(p.y) + (size_t)blockIdx.x * NPX * CIN;

    auto stamp = [&](int k) {
        if (p.ts && tid == 0) p.ts[(size_t)blockIdx.x * 8 + k] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);
    if (tid < 32) *reinterpret_cast<unsigned*>(lds + STG_OFF + tid * 4) = 0u;   // the zero row of phase B (visible after phase A's barriers)
    f32x16 acc[NACC];                                // [channel tile c of the wave][pixel tile i] at c * NPT + i
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    };
    // pixel-operand fragment reads in two rolling halves of the pixel tiles: the half just consumed is re-read for the
    // next k-step while the other half's MFMAs run
    u32x4 pf[NPT];
    auto rd_tiles = [&](unsigned a, int half) {      // pixel tiles half*HP .. of the image row block at `a` (tile t = +t*4096 bytes)
        static_assert(NPT == 8 || NPT == 4 || NPT == 2, "tile offsets are immediates");
        if (NPT == 2) {
            if (half == 0) pf[0] = lds_read_b128_o<0>(a); else pf[NPT - 1] = lds_read_b128_o<4096>(a);
        } else if (NPT == 8) {
            if (half == 0) { pf[0] = lds_read_b128_o<0>(a); pf[1] = lds_read_b128_o<4096>(a); pf[2] = lds_read_b128_o<8192>(a); pf[3] = lds_read_b128_o<12288>(a); }
            else { pf[4] = lds_read_b128_o<16384>(a); pf[5] = lds_read_b128_o<20480>(a); pf[6] = lds_read_b128_o<24576>(a); pf[NPT - 1] = lds_read_b128_o<28672>(a); }
        } else {
            if (half == 0) { pf[0] = lds_read_b128_o<0>(a); pf[1] = lds_read_b128_o<4096>(a); }
            else { pf[2] = lds_read_b128_o<8192>(a); pf[NPT - 1] = lds_read_b128_o<12288>(a); }
        }
    };
    auto mm_half = [&](const u32x4 (&w)[CPW], int half) {
#pragma unroll
        for (int c = 0; c < CPW; ++c)
#pragma unroll
            for (int i = 0; i < HP; ++i) acc[c * NPT + half * HP + i] = mfma_bf16(w[c], pf[half * HP + i], acc[c * NPT + half * HP + i]);
    };

    // ================================================================ phase A: t1 = relu(Wa x + ba), K = CIN
    {
        constexpr int NK = CIN / 64, XG = NPX / 64;  // K-tiles of x through a 4-slot ring, XG 1-KB row groups per wave and tile
        const int rsub = lane >> 3, cpos = lane & 7;
        const bf16_t* xsrc[XG];
#pragma unroll
        for (int i = 0; i < XG; ++i) {
            const int row = (wave * XG + i) * 8 + rsub;
            xsrc[i] = X + (size_t)row * CIN + (cpos ^ ((row >> 1) & 7)) * 8;
        }
        auto issue_x = [&](int kt) {
#pragma unroll
            for (int i = 0; i < XG; ++i) dma16(xsrc[i] + kt * 64, lds + (kt & 3) * SLICE + (wave * XG + i) * 1024);
        };
        u32x4 wq[2][4][CPW];                         // weight fragments of two K-tiles (inline-asm loads: counted by hand
        auto load_w = [&](int kt, int set) {         // next to the LDS-DMA queue, hipcc would drain it at every use)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int c = 0; c < CPW; ++c)
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(wq[set][ks][c]) : "v"(wfrag(p.fa, wave * CPW + c, CIN / 16, kt * 4 + ks, lane)) : "memory");
        };
        zero_acc();
        // queue per wave: W0 X0 X1 | iter kt: W(kt+1) X(kt+2) -> at the top of iter kt only X(kt+1) is younger than W(kt)
        load_w(0, 0);
        issue_x(0);
        issue_x(1);
        for (int kt2 = 0; kt2 < NK; kt2 += 2)
#pragma unroll
        for (int par = 0; par < 2; ++par) {          // par == kt & 1: register-set index is a compile-time constant
            const int kt = kt2 + par;
            if (kt + 1 < NK) wait_vmcnt<XG>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();            // x tile kt landed for every wave; tile kt-1 is done: slot (kt+2)&3 is free
            asm volatile("" ::: "memory");
            if (kt + 1 < NK) load_w(kt + 1, par ^ 1);
            if (kt + 2 < NK) issue_x(kt + 2);
            const unsigned xb = lds_base + (kt & 3) * SLICE;
            // pixel tile t of this lane = row t*32 + lrow: one swizzle key for all tiles -> one address per k-step, the tile
            // as an immediate offset
            const unsigned xrow = xb + lrow * ROWB;
            auto rd = [&](int ks, int half) {
                const unsigned a = xrow + (((2 * ks + lhalf) ^ ((lrow >> 1) & 7)) << 4);
                rd_tiles(a, half);
            };
            rd(0, 0);
            rd(0, 1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                lgkm_wait<HP>();
                mm_half(wq[par][ks], 0);
                if (ks < 3) rd(ks + 1, 0);
                if (ks < 3) lgkm_wait<HP>(); else lgkm_wait<0>();
                mm_half(wq[par][ks], 1);
                if (ks < 3) rd(ks + 1, 1);
            }
        }
        __builtin_amdgcn_s_barrier();                // every wave is done with the ring: its space becomes t1
        asm volatile("" ::: "memory");
    }
    // accumulators -> relu(acc + bias) -> bf16 -> image: channel tile ct lives in slice ct >> 1, chunks 4 * (ct & 1) + g
    auto store_img = [&](const float* bias) {
        // the write addresses depend only on the lane: left alone, hipcc computes them once for both calls and keeps them
        // live across phase B, i.e. spills them — and the second call then waits on ~15 serialised scratch reloads
        // (21.5k cycles for the t2 store against 4.2k for the identical t1 store).  Laundering the lane ids makes each call
        // recompute them.
        int lrow_ = lrow, lhalf_ = lhalf;
        asm volatile("" : "+v"(lrow_), "+v"(lhalf_));
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            const int ct = wave * CPW + c;
            float4 bq[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const float4*>(bias + ct * 32 + 8 * g + 4 * lhalf_);
#pragma unroll
            for (int i = 0; i < NPT; ++i) {
                const int px = i * 32 + lrow_;
                const f32x16& a = acc[c * NPT + i];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2 pk;
                    pk.x = relu2_bf16(a[4 * g] + bq[g].x, a[4 * g + 1] + bq[g].y);
                    pk.y = relu2_bf16(a[4 * g + 2] + bq[g].z, a[4 * g + 3] + bq[g].w);
                    lds_write_b64(lds_base + (ct >> 1) * SLICE + px * ROWB + ((((ct & 1) * 4 + g) ^ ((px >> 1) & 7)) << 4) + 8 * lhalf_, pk);
                }
            }
        }
        lds_wait();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    stamp(1);
    store_img(p.ba);
    stamp(2);

    // ================================================================ phase B: t2 = relu(Wb (*) t1 + bb), 9 taps x NSL slices
    {
        zero_acc();
        constexpr int KSB = 9 * C / 16;
        // this lane's pixel in tile i: index i*32 + lrow -> (frame, y, x)
        int pyy[NPT], pxx[NPT], pfr[NPT];
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int q = i * 32 + lrow;
            pfr[i] = q / (HW * HW);
            pyy[i] = (q / HW) % HW;
            pxx[i] = q % HW;
        }
        uint4 wn[4][CPW];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int c = 0; c < CPW; ++c) wn[ks][c] = *wfrag(p.fb, wave * CPW + c, KSB, ks, lane);
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3 - 1, kx = tap - (tap / 3) * 3 - 1;
            // shifted source row of this lane in every pixel tile; the 3x3's zero padding: out-of-frame lanes read a
            // 128-B row of zeros parked in the (still unused) store-staging area instead of being masked afterwards
            unsigned roff[NPT];                      // byte offset of the row; its swizzle key is (row >> 1) & 7 = (roff >> 8) & 7
            unsigned vmask = 0;
            // the lane's pixel coordinates are rebuilt from the lane id HERE, once per tap (two v_mbcnt + a few ALU ops): kept across the phase they
            // were spilled at 256 registers, and their reload - a scratch load followed by vmcnt(0) - opened every tap (round 6, late)
            unsigned lid;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lid));
            const int lr_ = (int)(lid & 31);
#pragma unroll
            for (int i = 0; i < NPT; ++i) {
                const int q_ = i * 32 + lr_;
                pfr[i] = q_ / (HW * HW); pyy[i] = (q_ / HW) % HW; pxx[i] = q_ % HW;
            }
#pragma unroll
            for (int i = 0; i < NPT; ++i) {
                const int y = pyy[i] + ky, x = pxx[i] + kx;
                const bool ok = (unsigned)y < (unsigned)HW && (unsigned)x < (unsigned)HW;
                roff[i] = ok ? ((pfr[i] * HW + y) * HW + x) * ROWB : STG_OFF;
                vmask |= ok ? (1u << i) : 0u;
            }
#pragma unroll 1
            for (int sl = 0; sl < NSL; ++sl) {
                u32x4 wc[4][CPW];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int c = 0; c < CPW; ++c) wc[ks][c] = as_u32x4(wn[ks][c]);
                const int nstep = (tap * NSL + sl + 1) * 4;     // next (tap, slice)'s fragments: in flight under this one's MFMAs
                if (nstep < KSB) {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                        for (int c = 0; c < CPW; ++c) wn[ks][c] = *wfrag(p.fb, wave * CPW + c, KSB, nstep + ks, lane);
                }
                unsigned rowa[NPT];                  // row address in slice sl (the zero row is not per slice)
#pragma unroll
                for (int t = 0; t < NPT; ++t) rowa[t] = lds_base + roff[t] + (((vmask >> t) & 1u) ? sl * SLICE : 0);
                auto rd = [&](int ks, int half) {
                    const int ch = 2 * ks + lhalf;
#pragma unroll
                    for (int i = 0; i < HP; ++i) {
                        const int t = half * HP + i;
                        pf[t] = lds_read_b128(rowa[t] + ((ch ^ ((roff[t] >> 8) & 7)) << 4));
                    }
                };
                rd(0, 0);
                rd(0, 1);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    lgkm_wait<HP>();
                    mm_half(wc[ks], 0);
                    if (ks < 3) rd(ks + 1, 0);
                    if (ks < 3) lgkm_wait<HP>(); else lgkm_wait<0>();
                    mm_half(wc[ks], 1);
                    if (ks < 3) rd(ks + 1, 1);
                }
            }
        }
        __builtin_amdgcn_s_barrier();                // every wave is done reading t1: t2 takes its place
        asm volatile("" ::: "memory");
    }
    stamp(3);
    store_img(p.bb);
    stamp(4);

    // ================================================================ phase C: y = relu(Wc t2 + bc + x), 4 chunks of 8*CPW channel tiles
    {
        constexpr int KSC = C / 16, NCH = (CIN / 32) / (8 * CPW), NSTEP = NCH * NSL;
        static_assert(NCH == 4, "four output chunks");
        float* stg = reinterpret_cast<float*>(lds + STG_OFF + wave * 4096);
        const int u = lane & 3, prr = lane >> 2;     // store pass: 8-channel group u of a 32-channel tile, pixel sub-row prr
        uint4 wn[4][CPW];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int c = 0; c < CPW; ++c) wn[ks][c] = *wfrag(p.fc, wave * CPW + c, KSC, ks, lane);
#pragma unroll 1
        for (int chunk = 0; chunk < NCH; ++chunk) {
            const int ct0 = (chunk * 8 + wave) * CPW;        // first output channel tile of this wave in this chunk
#pragma unroll
            for (int c = 0; c < CPW; ++c) {
                float4 bq[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const float4*>(p.bc + (ct0 + c) * 32 + 8 * g + 4 * lhalf);
#pragma unroll
                for (int i = 0; i < NPT; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x16& a = acc[c * NPT + i];
                        a[4 * g] = bq[g].x; a[4 * g + 1] = bq[g].y; a[4 * g + 2] = bq[g].z; a[4 * g + 3] = bq[g].w;
                    }
            }
#pragma unroll 1
            for (int sl = 0; sl < NSL; ++sl) {       // K = C: slice sl of t2, 4 k-steps each
                u32x4 wc[4][CPW];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int c = 0; c < CPW; ++c) wc[ks][c] = as_u32x4(wn[ks][c]);
                const int nxt = chunk * NSL + sl + 1;        // next (chunk, slice)
                if (nxt < NSTEP) {
                    const int nct0 = ((nxt / NSL) * 8 + wave) * CPW, nks = (nxt % NSL) * 4;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                        for (int c = 0; c < CPW; ++c) wn[ks][c] = *wfrag(p.fc, nct0 + c, KSC, nks + ks, lane);
                }
                const unsigned tb = lds_base + sl * SLICE;
                const unsigned trow = tb + lrow * ROWB;
                auto rd = [&](int ks, int half) {
                    const unsigned a = trow + (((2 * ks + lhalf) ^ ((lrow >> 1) & 7)) << 4);
                    rd_tiles(a, half);
                };
                rd(0, 0);
                rd(0, 1);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    lgkm_wait<HP>();
                    mm_half(wc[ks], 0);
                    if (ks < 3) rd(ks + 1, 0);
                    if (ks < 3) lgkm_wait<HP>(); else lgkm_wait<0>();
                    mm_half(wc[ks], 1);
                    if (ks < 3) rd(ks + 1, 1);
                }
            }
            if (chunk == NCH - 1) stamp(5);
            // store pass, one (channel tile, pixel tile) at a time through the wave's 4-KB staging tile [32 px][32 ch] fp32
            // (16-B slots XOR-swizzled by the pixel row): + residual -> ReLU -> bf16 -> 64-B row segments
            uint4 rr[2];
#pragma unroll
            for (int it = 0; it < 2; ++it) rr[it] = *reinterpret_cast<const uint4*>(X + (size_t)(it * 16 + prr) * CIN + (size_t)ct0 * 32 + 8 * u);
#pragma unroll
            for (int item = 0; item < NACC; ++item) {
                const int c = item / NPT, i = item % NPT;
                const size_t cofs = (size_t)(ct0 + c) * 32 + 8 * u;
                const f32x16& a = acc[item];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int slot = (2 * g + lhalf) ^ (lrow & 7);
                    *reinterpret_cast<float4*>(stg + lrow * 32 + slot * 4) = make_float4(a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]);
                }
                uint4 rn[2];
                if (item < NACC - 1) {
                    const int c1 = (item + 1) / NPT, i1 = (item + 1) % NPT;
#pragma unroll
                    for (int it = 0; it < 2; ++it)
                        rn[it] = *reinterpret_cast<const uint4*>(X + (size_t)(i1 * 32 + it * 16 + prr) * CIN + (size_t)(ct0 + c1) * 32 + 8 * u);
                }
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int pr = it * 16 + prr;
                    const float4 v0 = *reinterpret_cast<const float4*>(stg + pr * 32 + (((2 * u) ^ (pr & 7)) << 2));
                    const float4 v1 = *reinterpret_cast<const float4*>(stg + pr * 32 + (((2 * u + 1) ^ (pr & 7)) << 2));
                    const unsigned w4[4] = {rr[it].x, rr[it].y, rr[it].z, rr[it].w};
                    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                    unsigned pk[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        pk[k] = relu2_bf16(v[2 * k] + __uint_as_float(w4[k] << 16), v[2 * k + 1] + __uint_as_float(w4[k] & 0xffff0000u));
                    // ablation (IVOSW_ABLATION builds, tunable YS2ABL): the stage's LAST block keeps only the even pixels of y - what a forwarded
                    // conv1 + a stride-2 reader would need (how much clock do the written bytes cost?)
                    if (ABL(p.debug, 16) && ((((i * 32 + pr) / HW) | (i * 32 + pr)) & 1)) continue;
                    *reinterpret_cast<uint4*>(Y + (size_t)(i * 32 + pr) * CIN + cofs) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                }
                if (item < NACC - 1) { rr[0] = rn[0]; rr[1] = rn[1]; }
            }
        }
        stamp(6);
    }
}

template <int C, int HW, int FR>
__global__ __launch_bounds__(512) void bneck_wide_kernel(BneckWideArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[WIDE_LDS];   // the ONLY LDS object
    bneck_wide_body<C, HW, FR>(p, lds);
}

// A RUN of identity blocks in one launch: a workgroup owns a frame in every block, and the only consumer of a block's
// output frame is the same workgroup in the next block — no inter-workgroup dependency, so the blocks of a stage can be
// chained inside the kernel.  Saves the launch boundaries (drain + ramp of a 256-workgroup, single-wave grid) and lets
// the workgroups drift apart, so that the HBM-heavy phases (A: x in, C: y out + residual) of some overlap the MFMA-heavy
// phase B of others instead of all 256 CUs hitting HBM in lockstep.  Between blocks: a workgroup-scope release /
// acquire pair around a barrier (the next block's x is what this workgroup just wrote, through other waves).
template <int C, int HW, int FR>
__global__ __launch_bounds__(512) void bneck_wide_stage_kernel(BneckStageArgs s) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[WIDE_LDS];   // the ONLY LDS object
    if (s.stagger > 0 && ((blockIdx.x >> 3) & 1)) {
        // block b runs on XCD b % 8: (b >> 3) & 1 splits the workgroups of EVERY XCD into two halves, so that the memory phases
        // of one half meet the MFMA-only phase B of the other on the same XCD-to-memory link
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)s.stagger) __builtin_amdgcn_s_sleep(32);
    }
    for (int j = 0; j < s.n; ++j) {
        if (j > 0) {
            // workgroup scope: producer and consumer waves share this CU's write-through L1 (an agent-scope release writes the
            // XCD's whole L2 back each time: 1272 us for the 5-block run against 885 us as separate launches)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        bneck_wide_body<C, HW, FR>(s.blk[j], lds);
    }
}

// ---------------------------------------------------------------- halo variant: res3 (C = 128, 32x32 frames) and res2 (C = 64, 64x64)
// Same dataflow (wave-private fragment-ordered weights, t1 / t2 in LDS, transposed MFMAs), but the frame does not fit
// one workgroup: a workgroup owns a 16x16-pixel tile and phase A computes t1 on its 18x18 halo (352 rows = 11 pixel
// tiles, zeros stored for halo pixels outside the frame, so phase B needs no padding logic).  With NCT = C/32 channel
// tiles of t1 / t2 the 8 waves form NCT x NG groups (NG = 8/NCT = 2 | 4):
//   phase A  wave = (channel tile w % NCT, pixel tiles of group w / NCT: 6,5 | 3,3,3,2 of the 11)
//   phase B  wave = (channel tile, 8/NG pixel tiles), shifted rows hb + ky*18 + kx of the halo image
//   phase C  (4C/32)/8 chunks x (8 channel tiles = 8 waves) x 8 pixel tiles, store pass as in the frame kernel
// LDS: [0, 125952) x ring (3 slots of 41 row groups) -> t1 (C/64 slices x 352 rows) -> t2 (C/64 x 256 rows); staging at 128 KB.
// WL (round 5, C = 128): the weight fragments of phases A and B reach the waves THROUGH LDS - one LDS-DMA copy per workgroup, read by
// the two waves that share a channel tile - instead of each wave streaming its own copy from L2 into registers.  A CU takes only
// 18 - 25 B/clk through its L1 (profiles/r03_res3_ablation.txt: the 262 KB of phase A's duplicated fragments alone hold its loop at
// 14.6 k cycles; profiles/r05_gemm_tile_bench.txt), and 426 KB of the 1 855 KB a tile moves were second copies.  Same fragments,
// same k order: bit-identical to WL = false.  Rings: phase A two 16-KB K-tiles in the (then idle) staging area; phase B two 32-KB
// double steps in [t1 end, bias block) and the staging area, one workgroup barrier per double step.
template <int C, bool WL = false>
__global__ __launch_bounds__(512) void bneck_halo_kernel(BneckWideArgs p) {
    constexpr int CIN = 4 * C, HW = (C == 128) ? 32 : 64, BT = 16, HT = 18, HR = HT * HT, MH = 352, NGA = 41;
    static_assert(!WL || C == 128, "the LDS weight rings are laid out for C = 128");
    constexpr int TPX = HW / BT, TPF = TPX * TPX;    // tiles per frame edge / per frame
    constexpr int NSL = C / 64, NCT = C / 32, NG = 8 / NCT;
    constexpr int TPG = (11 + NG - 1) / NG;          // pixel tiles per group in phase A (6 | 3), the last group has one less
    constexpr int HA0 = (TPG + 1) / 2, HA1 = TPG - HA0;   // rolling halves in phase A
    constexpr int PB = 8 / NG, HB = PB / 2;          // pixel tiles per wave in phase B and per rolling half (4,2 | 2,1)
    constexpr int SLOT = NGA * 1024;                 // 41984 B per x K-tile
    constexpr int T1S = MH * ROWB, T2S = 256 * ROWB; // slice sizes of the t1 / t2 images
    constexpr int STG_OFF = 131072;
    constexpr int BIAS_OFF = 3 * SLOT;               // ba | bb | bc as floats in the gap below the staging area instead of a global
                                                     // load at the head of every epilogue / chunk (measured neutral here: 248 vs 254 us
                                                     // within box variance; it mattered in the two-workgroup res2 kernel)
    static_assert(BIAS_OFF + 6 * C * 4 <= STG_OFF, "bias block");
    static_assert(3 * SLOT <= STG_OFF && NSL * T1S <= STG_OFF && 2 * SLOT + MH * ROWB <= WIDE_LDS && TPG * (NG - 1) + TPG - 1 == 11, "geometry");
    __shared__ __attribute__((aligned(16))) unsigned char lds[WIDE_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhalf = lane >> 5;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int L = p.rev ? (int)gridDim.x - 1 - xcd_remap(blockIdx.x, gridDim.x) : xcd_remap(blockIdx.x, gridDim.x);
    const int b = L / TPF, tl = L - b * TPF;
    const int y0 = (tl / TPX) * BT, x0 = (tl % TPX) * BT;
    const bf16_t* X = static_cast<const bf16_t*>(p.x) + (size_t)b * HW * HW * CIN;
    bf16_t* Y = static_cast<bf16_t*>(p.y) + (size_t)b * HW * HW * CIN;
    const bf16_t* zeros = static_cast<const bf16_t*>(p.zeros);
    const int ctw = wave % NCT, grp = wave / NCT;
    if (tid < 2 * C) *reinterpret_cast<float*>(lds + BIAS_OFF + tid * 4) = tid < C ? p.ba[tid] : p.bb[tid - C];
    if (tid < 4 * C) *reinterpret_cast<float*>(lds + BIAS_OFF + 2 * C * 4 + tid * 4) = p.bc[tid];      // visible after phase A's barriers
    uint4 wn[4];                                     // first weight fragments of the next phase, requested one phase early
    auto stamp = [&](int k) {
        if (p.ts && tid == 0) p.ts[(size_t)blockIdx.x * 8 + k] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);
    // WL: double step d of phase B = steps 2d, 2d + 1 = 8 consecutive fragments (8 KB) of each of the 4 channel tiles -> ring slot d & 1
    constexpr int WB0_OFF = NSL * T1S, WB1_OFF = STG_OFF;
    static_assert(!WL || (WB0_OFF + 32768 <= BIAS_OFF), "phase B weight ring");
    auto wb_issue = [&](int d) {
        unsigned char* slot = lds + ((d & 1) ? WB1_OFF : WB0_OFF);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ct = wave >> 1, sub = (wave & 1) * 4 + j;
            dma16(wfrag(p.fb, ct, 9 * C / 16, d * 8 + sub, lane), slot + (ct * 8 + sub) * 1024);
        }
    };
    // (Tried: picking the residual of output chunk 0 out of the ring while its x K-tile is resident in phase A, to save
    // re-reading half of x in phase C — 64 more live registers on top of phase A's 96 accumulator + 32 weight registers
    // = 141 VGPR spills.  Dropped.)

    // ================================================================ phase A: t1 = relu(Wa x + ba) on the halo, K = CIN
    {
        constexpr int NK = CIN / 64;
        f32x16 acc[TPG];
#pragma unroll
        for (int i = 0; i < TPG; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const int rsub = lane >> 3, cpos = lane & 7;
        const int np = (NGA - wave + 7) / 8;         // row groups of this wave: wave, wave+8, ... (6 for wave 0, else 5)
        const bf16_t* xsrc[6];
        unsigned okmask = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int g = wave + 8 * i;
            const int hr = g * 8 + rsub;
            const int hy = hr / HT, hx = hr - hy * HT;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const bool ok = g < NGA && hr < HR && y >= 0 && y < HW && x >= 0 && x < HW;
            xsrc[i] = ok ? X + ((size_t)y * HW + x) * CIN + (cpos ^ ((hr >> 1) & 7)) * 8 : zeros;
            okmask |= ok ? (1u << i) : 0u;
        }
        auto issue_x = [&](int kt) {
            unsigned char* sb = lds + (kt % 3) * SLOT;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int g = wave + 8 * i;
                if (g < NGA) dma16((HALO_ABL & 4) ? zeros : xsrc[i] + (((okmask >> i) & 1u) ? kt * 64 : 0), sb + g * 1024);
            }
        };
        u32x4 wq[2][4];
        auto load_w = [&](int kt, int set) {
            if constexpr (WL) {
                // K-tile kt of all NCT channel tiles = 16 fragments = 16 KB into ring slot kt & 1: wave w copies fragments (ct = w >> 1,
                // ks = 2 (w & 1), + 1); the queue order W(kt + 1) | X(kt + 2) per iteration is that of the register path
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int ct = wave >> 1, ks = (wave & 1) * 2 + j;
                    dma16(wfrag(p.fa, ct, CIN / 16, kt * 4 + ks, lane), lds + STG_OFF + (kt & 1) * 16384 + (ct * 4 + ks) * 1024);
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(wq[set][ks]) : "v"(wfrag(p.fa, ctw, CIN / 16, kt * 4 + ks, lane)) : "memory");
            }
        };
        const int pt0 = grp * TPG;                   // pixel tiles pt0 .. ; the last group lacks its final tile (there are 11)
        const bool full = grp + 1 < NG;
        load_w(0, 0);
        issue_x(0);
        issue_x(1);
        for (int kt2 = 0; kt2 < NK; kt2 += 2)
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const int kt = kt2 + par;
            if (kt + 1 < NK) wait_vmcnt_n(np); else wait_vmcnt<0>();   // only X(kt+1) is younger than W(kt)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + 1 < NK) load_w(kt + 1, par ^ 1);
            if (kt + 2 < NK) issue_x(kt + 2);        // slot (kt+2) % 3 == slot of tile kt-1: done for every wave
            const unsigned xb = lds_base + (kt % 3) * SLOT;
            u32x4 pf[TPG];
            // pixel tile pt0 + t of this lane = row (pt0 + t)*32 + lrow: one swizzle key for every t -> one address per k-step,
            // the tile as an immediate offset
            const unsigned xrow = xb + (pt0 * 32 + lrow) * ROWB;
            auto rd = [&](int ks, int half) {
                const unsigned a = xrow + (((2 * ks + lhalf) ^ ((lrow >> 1) & 7)) << 4);
                auto one = [&](auto tc) {
                    constexpr int t = decltype(tc)::value;
                    if (t >= (half ? HA0 : 0) && t < (half ? TPG : HA0) && (t < TPG - 1 || full)) pf[t] = lds_read_b128_o<t * 4096>(a);
                };
                one(std::integral_constant<int, 0>{}); one(std::integral_constant<int, 1>{}); one(std::integral_constant<int, 2>{});
                if constexpr (TPG == 6) { one(std::integral_constant<int, 3>{}); one(std::integral_constant<int, 4>{}); one(std::integral_constant<int, 5>{}); }
                static_assert(TPG == 3 || TPG == 6, "tile offsets are immediates");
            };
            if constexpr (WL) {                      // this wave's four fragments of the K-tile: issued first, so every counted wait below covers them
                const unsigned wa = lds_base + STG_OFF + (kt & 1) * 16384 + ctw * 4096 + lane * 16;
                wq[par][0] = lds_read_b128_o<0>(wa); wq[par][1] = lds_read_b128_o<1024>(wa);
                wq[par][2] = lds_read_b128_o<2048>(wa); wq[par][3] = lds_read_b128_o<3072>(wa);
            }
            rd(0, 0);
            rd(0, 1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const u32x4 w = wq[par][ks];
                if (full) lgkm_wait<HA1>(); else lgkm_wait<HA1 - 1>();       // half 0 landed (half 1 may be in flight)
#pragma unroll
                for (int t = 0; t < HA0; ++t) acc[t] = mfma_bf16(w, pf[t], acc[t]);
                if (ks < 3) rd(ks + 1, 0);
                if (ks < 3) lgkm_wait<HA0>(); else lgkm_wait<0>();
#pragma unroll
                for (int t = HA0; t < TPG; ++t)
                    if (t < TPG - 1 || full) acc[t] = mfma_bf16(w, pf[t], acc[t]);
                if (ks < 3) rd(ks + 1, 1);
            }
        }
        __builtin_amdgcn_s_barrier();                // the ring is dead: its space becomes t1
        asm volatile("" ::: "memory");
        stamp(1);
        if constexpr (WL) {
            // phase B's first two double steps go out under the t1 epilogue: slot 0 = [t1 end, bias block) - part of the x ring until
            // the barrier above -, slot 1 = the staging area (phase A's weight ring, dead as well)
            wb_issue(0);
            wb_issue(1);
        } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) wn[ks] = *wfrag(p.fb, ctw, 9 * C / 16, ks, lane);
        }
        float4 bq[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const float4*>(lds + BIAS_OFF + (ctw * 32 + 8 * g + 4 * lhalf) * 4);
#pragma unroll
        for (int i = 0; i < TPG; ++i) {
            if (i == TPG - 1 && !full) break;
            const int hr = (pt0 + i) * 32 + lrow;
            const int hy = hr / HT, hx = hr - hy * HT;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const bool in = y >= 0 && y < HW && x >= 0 && x < HW;
            if (hr < HR) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2 pk;
                    pk.x = relu2_bf16(acc[i][4 * g] + bq[g].x, acc[i][4 * g + 1] + bq[g].y) & (in ? 0xffffffffu : 0u);
                    pk.y = relu2_bf16(acc[i][4 * g + 2] + bq[g].z, acc[i][4 * g + 3] + bq[g].w) & (in ? 0xffffffffu : 0u);
                    // swizzle key from the COLUMN of the halo raster, not from the row index (round 5): phase B's 16-lane read groups are
                    // {x .. x+3, x+12 .. x+15} of one raster line and {x+4 .. x+11} of the next - 16 consecutive columns, so (hx & 1, hx >> 1)
                    // puts them on 16 different 16-byte slots (the line pitch 18 is even: a row's 128-byte half is its column's parity);
                    // with (hr >> 1) & 7 two pairs of every group shared a slot (PMC: 32 % of the kernel's LDS cycles were conflicts)
                    lds_write_b64(lds_base + (ctw >> 1) * T1S + hr * ROWB + ((((ctw & 1) * 4 + g) ^ ((hx >> 1) & 7)) << 4) + 8 * lhalf, pk);
                }
            }
        }
        lds_wait();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stamp(2);
    }

    // ================================================================ phase B: t2 = relu(Wb (*) t1 + bb), 9 taps x NSL slices
    {
        constexpr int KSB = 9 * C / 16, NSTEP = 9 * NSL;
        f32x16 acc[PB];
#pragma unroll
        for (int i = 0; i < PB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        int hb[PB];                                  // halo row of this lane's output pixel at tap (0,0)
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int q = (grp * PB + i) * 32 + lrow;
            hb[i] = (q >> 4) * HT + (q & 15);
        }
        if constexpr (WL) {
            // Software-pipelined form (round 5): the fragment reads of a k-step - 4 pixel tiles - are issued a WHOLE k-step ahead into the
            // other of two register sets, and across the step boundary the next step's four weight fragments and first pixel fragments
            // go out under the last k-step's MFMAs; the workgroup barrier of a double step sits in front of that prefetch.  (The
            // half-step lookahead of the register-weights form left every step opening with ~10 exposed LDS reads.)
            static_assert(!WL || (NSTEP % 2 == 0 && PB == 4), "double steps of four pixel tiles");
            u32x4 pfb[2][PB], wcb[2][4];
            auto rd_all = [&](int buf, int step, int ks) {
                const int tap = step / NSL, sl = step - tap * NSL;
                const unsigned key = (((lrow & 15) + tap % 3) >> 1) & 7;     // the raster column of every pixel tile of this lane: (q & 15) + kx
                const unsigned x = ((2 * ks + lhalf) ^ key) << 4;
#pragma unroll
                for (int i = 0; i < PB; ++i) pfb[buf][i] = lds_read_b128(lds_base + sl * T1S + (hb[i] + (tap / 3) * HT + (tap % 3)) * ROWB + x);
            };
            auto rd_w = [&](int buf, int step) {
                const unsigned wa = lds_base + (((step >> 1) & 1) ? WB1_OFF : WB0_OFF) + (ctw * 8 + (step & 1) * 4) * 1024 + lane * 16;
                wcb[buf][0] = lds_read_b128_o<0>(wa); wcb[buf][1] = lds_read_b128_o<1024>(wa);
                wcb[buf][2] = lds_read_b128_o<2048>(wa); wcb[buf][3] = lds_read_b128_o<3072>(wa);
            };
            // double step 0 has landed for this wave (double step 1's four pieces may be younger); the barrier publishes it
            wait_vmcnt<4>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            rd_w(0, 0);
            rd_all(0, 0, 0);
#pragma unroll 1
            for (int d = 0; d < NSTEP / 2; ++d) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int step = 2 * d + h;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        if (ks < 3) {
                            rd_all((ks + 1) & 1, step, ks + 1);
                            lgkm_wait<PB>();
                        } else if (step + 1 < NSTEP) {
                            if (h == 1) {
                                // double step d + 1: this wave's pieces have landed (nothing younger is in flight), the barrier publishes
                                // them and says every wave has read the last weights of double step d: its slot takes d + 2
                                wait_vmcnt<0>();
                                __builtin_amdgcn_s_barrier();
                                asm volatile("" ::: "memory");
                                if (d + 2 < NSTEP / 2) wb_issue(d + 2);
                            }
                            rd_w(h ^ 1, step + 1);
                            rd_all(0, step + 1, 0);
                            lgkm_wait<PB + 4>();
                        } else {
                            lgkm_wait<0>();
                        }
#pragma unroll
                        for (int i = 0; i < PB; ++i) acc[i] = mfma_bf16(wcb[h][ks], pfb[ks & 1][i], acc[i]);
                    }
                }
            }
        } else {
#pragma unroll 1
        for (int step = 0; step < NSTEP; ++step) {   // step = tap * NSL + slice
            const int tap = step / NSL, sl = step - tap * NSL;
            u32x4 wc[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) wc[ks] = as_u32x4(wn[ks]);
            if (step + 1 < NSTEP) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) wn[ks] = *wfrag(p.fb, ctw, KSB, (step + 1) * 4 + ks, lane);
            }
            unsigned rowa[PB];
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const int hr = hb[i] + (tap / 3) * HT + (tap % 3);
                rowa[i] = lds_base + sl * T1S + hr * ROWB;
            }
            const unsigned rkey = (((lrow & 15) + tap % 3) >> 1) & 7;    // the raster column of every pixel tile of this lane: (q & 15) + kx
            u32x4 pf[PB];
            auto rd = [&](int ks, int half) {
                const int ch = 2 * ks + lhalf;
#pragma unroll
                for (int i = 0; i < HB; ++i) pf[half * HB + i] = lds_read_b128(rowa[half * HB + i] + ((ch ^ rkey) << 4));
            };
            rd(0, 0);
            rd(0, 1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                lgkm_wait<HB>();
#pragma unroll
                for (int i = 0; i < HB; ++i) acc[i] = mfma_bf16(wc[ks], pf[i], acc[i]);
                if (ks < 3) rd(ks + 1, 0);
                if (ks < 3) lgkm_wait<HB>(); else lgkm_wait<0>();
#pragma unroll
                for (int i = HB; i < PB; ++i) acc[i] = mfma_bf16(wc[ks], pf[i], acc[i]);
                if (ks < 3) rd(ks + 1, 1);
            }
        }
        }
        __builtin_amdgcn_s_barrier();                // every wave is done reading t1: t2 takes its place
        asm volatile("" ::: "memory");
        stamp(3);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wn[ks] = *wfrag(p.fc, wave, C / 16, ks, lane);
        float4 bq[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const float4*>(lds + BIAS_OFF + (C + ctw * 32 + 8 * g + 4 * lhalf) * 4);
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int px = (grp * PB + i) * 32 + lrow;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 pk;
                pk.x = relu2_bf16(acc[i][4 * g] + bq[g].x, acc[i][4 * g + 1] + bq[g].y);
                pk.y = relu2_bf16(acc[i][4 * g + 2] + bq[g].z, acc[i][4 * g + 3] + bq[g].w);
                lds_write_b64(lds_base + (ctw >> 1) * T2S + px * ROWB + ((((ctw & 1) * 4 + g) ^ ((px >> 1) & 7)) << 4) + 8 * lhalf, pk);
            }
        }
        lds_wait();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stamp(4);
    }

    // ================================================================ phase C: y = relu(Wc t2 + bc + x), chunks of 8 channel tiles
    {
        constexpr int KSC = C / 16, NCH = (CIN / 32) / 8, NSTEP = NCH * NSL;
        f32x16 acc[8];
        float* stg = reinterpret_cast<float*>(lds + STG_OFF + wave * 4096);
        const int u = lane & 3, prr = lane >> 2;
        auto pix = [&](int q) { return (size_t)((y0 + (q >> 4)) * HW + x0 + (q & 15)) * CIN; };   // tile pixel q -> frame offset
#pragma unroll 1
        for (int chunk = 0; chunk < NCH; ++chunk) {
            const int ct = chunk * 8 + wave;
            const size_t cofs = (size_t)ct * 32 + 8 * u;
            uint4 rr[2];                             // residual of the first pixel tile: in flight under the chunk's MFMAs
#pragma unroll
            for (int it = 0; it < 2; ++it) rr[it] = (HALO_ABL & 1) ? make_uint4(0, 0, 0, 0) : *reinterpret_cast<const uint4*>(X + pix(it * 16 + prr) + cofs);
            {
                float4 bq[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const float4*>(lds + BIAS_OFF + (2 * C + ct * 32 + 8 * g + 4 * lhalf) * 4);
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        acc[i][4 * g] = bq[g].x; acc[i][4 * g + 1] = bq[g].y; acc[i][4 * g + 2] = bq[g].z; acc[i][4 * g + 3] = bq[g].w;
                    }
            }
#pragma unroll 1
            for (int sl = 0; sl < NSL; ++sl) {
                u32x4 wc[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) wc[ks] = as_u32x4(wn[ks]);
                const int nxt = chunk * NSL + sl + 1;
                if (nxt < NSTEP) {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) wn[ks] = *wfrag(p.fc, (nxt / NSL) * 8 + wave, KSC, (nxt % NSL) * 4 + ks, lane);
                }
                const unsigned tb = lds_base + sl * T2S;
                u32x4 pf[8];
                const unsigned trow = tb + lrow * ROWB;
                auto rd = [&](int ks, int half) {
                    const unsigned a = trow + (((2 * ks + lhalf) ^ ((lrow >> 1) & 7)) << 4);
                    if (half == 0) { pf[0] = lds_read_b128_o<0>(a); pf[1] = lds_read_b128_o<4096>(a); pf[2] = lds_read_b128_o<8192>(a); pf[3] = lds_read_b128_o<12288>(a); }
                    else { pf[4] = lds_read_b128_o<16384>(a); pf[5] = lds_read_b128_o<20480>(a); pf[6] = lds_read_b128_o<24576>(a); pf[7] = lds_read_b128_o<28672>(a); }
                };
                rd(0, 0);
                rd(0, 1);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    lgkm_wait<4>();
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = mfma_bf16(wc[ks], pf[i], acc[i]);
                    if (ks < 3) rd(ks + 1, 0);
                    if (ks < 3) lgkm_wait<4>(); else lgkm_wait<0>();
#pragma unroll
                    for (int i = 4; i < 8; ++i) acc[i] = mfma_bf16(wc[ks], pf[i], acc[i]);
                    if (ks < 3) rd(ks + 1, 1);
                }
            }
            if (chunk == NCH - 1) stamp(5);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int slot = (2 * g + lhalf) ^ (lrow & 7);
                    *reinterpret_cast<float4*>(stg + lrow * 32 + slot * 4) = make_float4(acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]);
                }
                uint4 rn[2];
                if (i < 7) {
#pragma unroll
                    for (int it = 0; it < 2; ++it) rn[it] = (HALO_ABL & 1) ? make_uint4(0, 0, 0, 0) : *reinterpret_cast<const uint4*>(X + pix((i + 1) * 32 + it * 16 + prr) + cofs);
                }
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int pr = it * 16 + prr;
                    const float4 v0 = *reinterpret_cast<const float4*>(stg + pr * 32 + (((2 * u) ^ (pr & 7)) << 2));
                    const float4 v1 = *reinterpret_cast<const float4*>(stg + pr * 32 + (((2 * u + 1) ^ (pr & 7)) << 2));
                    const unsigned w4[4] = {rr[it].x, rr[it].y, rr[it].z, rr[it].w};
                    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                    unsigned pk[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        pk[k] = relu2_bf16(v[2 * k] + __uint_as_float(w4[k] << 16), v[2 * k + 1] + __uint_as_float(w4[k] & 0xffff0000u));
                    if ((HALO_ABL & 2) && pk[0] != 0x12345678u) continue;
                    if (ABL(p.debug, 16) && ((((i * 32 + pr) >> 4) | (i * 32 + pr)) & 1)) continue;      // ablation YS2ABL: even pixels only (see bneck_wide_body)
                    *reinterpret_cast<uint4*>(Y + pix(i * 32 + pr) + cofs) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                }
                if (i < 7) { rr[0] = rn[0]; rr[1] = rn[1]; }
            }
        }
        stamp(6);
    }
}

// ---------------------------------------------------------------- res3, small tile: two workgroups per CU
// bneck_halo_kernel<128> above runs ONE workgroup per CU: its memory phases (A: 360 KB x halo in, C: residual in / y out) and
// its MFMA-bound phase B alternate (31 k + 27 k + 27 k cycles per tile) instead of overlapping.  Same dataflow on an 8 x 16-pixel
// tile (10 x 18 halo = 180 rows = 6 pixel tiles): 75 KB of LDS and <= 128 VGPRs, so two workgroups share a CU and one sits in
// the matrix cores while the other waits on memory (what bneck_halo64s_kernel does for res2).
//   phase A  wave = (channel tile w & 3, pixel tiles 3 g .. 3 g + 2 of the 6 with g = w >> 2); x through a 3-slot LDS-DMA ring
//   phase B  wave = (channel tile w & 3, pixel tiles 2 g, 2 g + 1 of the 4 output tiles), 18 steps (tap, slice) of 4 k-steps
//   phase C  2 chunks x (8 channel tiles = 8 waves) x 4 pixel tiles, store pass through the per-wave staging tile
// LDS: [0, 73728) ring (3 x 24 KB) -> t1 (2 slices x 192 rows) [0, 49152) -> t2 (2 x 128 rows) [0, 32768) | staging [32768, 65536);
//      [73728, 76800) biases.  Per pixel the arithmetic and its order are those of bneck_halo_kernel<128> (bit-identical output).
// MEASURED SLOWER and therefore off by default (tunable HALO128S=1 selects it; tests keep it bit-identical): 272 against 230 us per
// block at B = 256 (tower 3.73 against 3.54 ms).  What works for res2 does not carry over: with C = 128 every wave re-streams ~1 MB of
// weight fragments per tile from L2 through the texture cache (phase B alone 590 KB) - per 128 pixels now instead of per 256 -
// which is ~16 k cache accesses per tile against ~19 k cycles of MFMA issue, the halo grows from 1.27 x to 1.41 x of the tile, and
// the 128-VGPR budget of two workgroups per CU costs 31 spilled registers.
__global__ __launch_bounds__(512, 4) void bneck_halo128s_kernel(BneckWideArgs p) {
    constexpr int C = 128, CIN = 512, HW = 32, BTY = 8, BTX = 16, HTX = 18, HR = 180, MH = 192, NGA = 23;
    constexpr int TPX = HW / BTX, TPF = (HW / BTY) * TPX;          // 2 tiles across, 8 per frame
    constexpr int NSL = 2, NCT = 4, TPG = 3;
    constexpr int SLOT = 24576;
    constexpr int T1S = MH * ROWB, T2S = 128 * ROWB;
    constexpr int STG_OFF = 32768, BIAS_OFF = 3 * SLOT, LDS_BYTES = BIAS_OFF + 6 * C * 4;       // 76800
    static_assert(NSL * T1S <= 3 * SLOT && NSL * T2S <= STG_OFF && STG_OFF + 8 * 4096 <= 3 * SLOT && 2 * LDS_BYTES <= WIDE_LDS, "LDS map");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhalf = lane >> 5;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int b = L / TPF, tl = L - b * TPF;
    const int y0 = (tl / TPX) * BTY, x0 = (tl % TPX) * BTX;
    const bf16_t* X = static_cast<const bf16_t*>(p.x) + (size_t)b * HW * HW * CIN;
    bf16_t* Y = static_cast<bf16_t*>(p.y) + (size_t)b * HW * HW * CIN;
    const bf16_t* zeros = static_cast<const bf16_t*>(p.zeros);
    const int ctw = wave & 3, grp = wave >> 2;
    if (tid < 2 * C) *reinterpret_cast<float*>(lds + BIAS_OFF + tid * 4) = tid < C ? p.ba[tid] : p.bb[tid - C];
    *reinterpret_cast<float*>(lds + BIAS_OFF + 2 * C * 4 + tid * 4) = p.bc[tid];             // 512 threads, 512 channels; visible after phase A's barriers
    uint4 wn[4];                                     // first weight fragments of the next phase, requested one phase early
    auto stamp = [&](int k) {
        if (p.ts && tid == 0) p.ts[(size_t)blockIdx.x * 8 + k] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);

    // ================================================================ phase A: t1 = relu(Wa x + ba) on the halo, K = CIN
    {
        constexpr int NK = CIN / 64;
        f32x16 acc[TPG];
#pragma unroll
        for (int i = 0; i < TPG; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const int rsub = lane >> 3, cpos = lane & 7;
        const int np = (NGA - wave + 7) / 8;         // row groups of this wave: wave, wave + 8, wave + 16 (< 23)
        const bf16_t* xsrc[3];
        unsigned okmask = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int g = wave + 8 * i;
            const int hr = g * 8 + rsub;
            const int hy = hr / HTX, hx = hr - hy * HTX;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const bool ok = g < NGA && hr < HR && y >= 0 && y < HW && x >= 0 && x < HW;
            xsrc[i] = ok ? X + ((size_t)y * HW + x) * CIN + (cpos ^ ((hr >> 1) & 7)) * 8 : zeros;
            okmask |= ok ? (1u << i) : 0u;
        }
        auto issue_x = [&](int kt) {
            unsigned char* sb = lds + (kt % 3) * SLOT;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int g = wave + 8 * i;
                if (g < NGA) dma16(xsrc[i] + (((okmask >> i) & 1u) ? kt * 64 : 0), sb + g * 1024);
            }
        };
        u32x4 wq[2][4];
        auto load_w = [&](int kt, int set) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(wq[set][ks]) : "v"(wfrag(p.fa, ctw, CIN / 16, kt * 4 + ks, lane)) : "memory");
        };
        const int pt0 = grp * TPG;
        load_w(0, 0);
        issue_x(0);
        issue_x(1);
        for (int kt2 = 0; kt2 < NK; kt2 += 2)
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const int kt = kt2 + par;
            if (kt + 1 < NK) wait_vmcnt_n(np); else wait_vmcnt<0>();   // only X(kt+1) is younger than W(kt)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + 1 < NK) load_w(kt + 1, par ^ 1);
            if (kt + 2 < NK) issue_x(kt + 2);        // slot (kt+2) % 3 == slot of tile kt-1: done for every wave
            const unsigned xb = lds_base + (kt % 3) * SLOT;
            u32x4 pf[TPG];
            const unsigned xrow = xb + (pt0 * 32 + lrow) * ROWB;
            auto rd = [&](int ks, int half) {
                const unsigned a = xrow + (((2 * ks + lhalf) ^ ((lrow >> 1) & 7)) << 4);
                if (half == 0) { pf[0] = lds_read_b128_o<0>(a); pf[1] = lds_read_b128_o<4096>(a); }
                else pf[2] = lds_read_b128_o<8192>(a);
            };
            rd(0, 0);
            rd(0, 1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const u32x4 w = wq[par][ks];
                lgkm_wait<1>();                      // half 0 landed (half 1 may be in flight)
                acc[0] = mfma_bf16(w, pf[0], acc[0]);
                acc[1] = mfma_bf16(w, pf[1], acc[1]);
                if (ks < 3) rd(ks + 1, 0);
                if (ks < 3) lgkm_wait<2>(); else lgkm_wait<0>();
                acc[2] = mfma_bf16(w, pf[2], acc[2]);
                if (ks < 3) rd(ks + 1, 1);
            }
        }
        __builtin_amdgcn_s_barrier();                // the ring is dead: its space becomes t1
        asm volatile("" ::: "memory");
        stamp(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wn[ks] = *wfrag(p.fb, ctw, 9 * C / 16, ks, lane);
        float4 bq[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const float4*>(lds + BIAS_OFF + (ctw * 32 + 8 * g + 4 * lhalf) * 4);
#pragma unroll
        for (int i = 0; i < TPG; ++i) {
            const int hr = (pt0 + i) * 32 + lrow;
            const int hy = hr / HTX, hx = hr - hy * HTX;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const bool in = y >= 0 && y < HW && x >= 0 && x < HW;
            if (hr < HR) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2 pk;
                    pk.x = relu2_bf16(acc[i][4 * g] + bq[g].x, acc[i][4 * g + 1] + bq[g].y) & (in ? 0xffffffffu : 0u);
                    pk.y = relu2_bf16(acc[i][4 * g + 2] + bq[g].z, acc[i][4 * g + 3] + bq[g].w) & (in ? 0xffffffffu : 0u);
                    lds_write_b64(lds_base + (ctw >> 1) * T1S + hr * ROWB + ((((ctw & 1) * 4 + g) ^ ((hr >> 1) & 7)) << 4) + 8 * lhalf, pk);
                }
            }
        }
        lds_wait();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stamp(2);
    }

    // ================================================================ phase B: t2 = relu(Wb (*) t1 + bb), 9 taps x NSL slices
    {
        constexpr int KSB = 9 * C / 16, NSTEP = 9 * NSL;
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        int hb[2];                                   // halo row of this lane's output pixel at tap (0,0)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = (grp * 2 + i) * 32 + lrow;
            hb[i] = (q >> 4) * HTX + (q & 15);
        }
#pragma unroll 1
        for (int step = 0; step < NSTEP; ++step) {   // step = tap * NSL + slice
            const int tap = step / NSL, sl = step - tap * NSL;
            u32x4 wc[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) wc[ks] = as_u32x4(wn[ks]);
            if (step + 1 < NSTEP) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) wn[ks] = *wfrag(p.fb, ctw, KSB, (step + 1) * 4 + ks, lane);
            }
            unsigned rowa[2], rkey[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int hr = hb[i] + (tap / 3) * HTX + (tap % 3);
                rowa[i] = lds_base + sl * T1S + hr * ROWB;
                rkey[i] = (hr >> 1) & 7;
            }
            u32x4 pf[2];
            auto rd = [&](int ks, int half) {
                const int ch = 2 * ks + lhalf;
                pf[half] = lds_read_b128(rowa[half] + ((ch ^ rkey[half]) << 4));
            };
            rd(0, 0);
            rd(0, 1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                lgkm_wait<1>();
                acc[0] = mfma_bf16(wc[ks], pf[0], acc[0]);
                if (ks < 3) rd(ks + 1, 0);
                if (ks < 3) lgkm_wait<1>(); else lgkm_wait<0>();
                acc[1] = mfma_bf16(wc[ks], pf[1], acc[1]);
                if (ks < 3) rd(ks + 1, 1);
            }
        }
        __builtin_amdgcn_s_barrier();                // every wave is done reading t1: t2 takes its place
        asm volatile("" ::: "memory");
        stamp(3);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wn[ks] = *wfrag(p.fc, wave, C / 16, ks, lane);
        float4 bq[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const float4*>(lds + BIAS_OFF + (C + ctw * 32 + 8 * g + 4 * lhalf) * 4);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int px = (grp * 2 + i) * 32 + lrow;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 pk;
                pk.x = relu2_bf16(acc[i][4 * g] + bq[g].x, acc[i][4 * g + 1] + bq[g].y);
                pk.y = relu2_bf16(acc[i][4 * g + 2] + bq[g].z, acc[i][4 * g + 3] + bq[g].w);
                lds_write_b64(lds_base + (ctw >> 1) * T2S + px * ROWB + ((((ctw & 1) * 4 + g) ^ ((px >> 1) & 7)) << 4) + 8 * lhalf, pk);
            }
        }
        lds_wait();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stamp(4);
    }

    // ================================================================ phase C: y = relu(Wc t2 + bc + x), 2 chunks of 8 channel tiles
    {
        constexpr int KSC = C / 16, NCH = (CIN / 32) / 8, NSTEP = NCH * NSL;
        f32x16 acc[4];
        float* stg = reinterpret_cast<float*>(lds + STG_OFF + wave * 4096);
        const int u = lane & 3, prr = lane >> 2;
        auto pix = [&](int q) { return (size_t)((y0 + (q >> 4)) * HW + x0 + (q & 15)) * CIN; };   // tile pixel q -> frame offset
#pragma unroll 1
        for (int chunk = 0; chunk < NCH; ++chunk) {
            const int ct = chunk * 8 + wave;
            const size_t cofs = (size_t)ct * 32 + 8 * u;
            uint4 rr[2];                             // residual of the first pixel tile: in flight under the chunk's MFMAs
#pragma unroll
            for (int it = 0; it < 2; ++it) rr[it] = *reinterpret_cast<const uint4*>(X + pix(it * 16 + prr) + cofs);
            {
                float4 bq[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const float4*>(lds + BIAS_OFF + (2 * C + ct * 32 + 8 * g + 4 * lhalf) * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        acc[i][4 * g] = bq[g].x; acc[i][4 * g + 1] = bq[g].y; acc[i][4 * g + 2] = bq[g].z; acc[i][4 * g + 3] = bq[g].w;
                    }
            }
#pragma unroll 1
            for (int sl = 0; sl < NSL; ++sl) {
                u32x4 wc[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) wc[ks] = as_u32x4(wn[ks]);
                const int nxt = chunk * NSL + sl + 1;
                if (nxt < NSTEP) {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) wn[ks] = *wfrag(p.fc, (nxt / NSL) * 8 + wave, KSC, (nxt % NSL) * 4 + ks, lane);
                }
                const unsigned tb = lds_base + sl * T2S;
                u32x4 pf[4];
                const unsigned trow = tb + lrow * ROWB;
                auto rd = [&](int ks, int half) {
                    const unsigned a = trow + (((2 * ks + lhalf) ^ ((lrow >> 1) & 7)) << 4);
                    if (half == 0) { pf[0] = lds_read_b128_o<0>(a); pf[1] = lds_read_b128_o<4096>(a); }
                    else { pf[2] = lds_read_b128_o<8192>(a); pf[3] = lds_read_b128_o<12288>(a); }
                };
                rd(0, 0);
                rd(0, 1);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    lgkm_wait<2>();
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[i] = mfma_bf16(wc[ks], pf[i], acc[i]);
                    if (ks < 3) rd(ks + 1, 0);
                    if (ks < 3) lgkm_wait<2>(); else lgkm_wait<0>();
#pragma unroll
                    for (int i = 2; i < 4; ++i) acc[i] = mfma_bf16(wc[ks], pf[i], acc[i]);
                    if (ks < 3) rd(ks + 1, 1);
                }
            }
            if (chunk == NCH - 1) stamp(5);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int slot = (2 * g + lhalf) ^ (lrow & 7);
                    *reinterpret_cast<float4*>(stg + lrow * 32 + slot * 4) = make_float4(acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]);
                }
                uint4 rn[2];
                if (i < 3) {
#pragma unroll
                    for (int it = 0; it < 2; ++it) rn[it] = *reinterpret_cast<const uint4*>(X + pix((i + 1) * 32 + it * 16 + prr) + cofs);
                }
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int pr = it * 16 + prr;
                    const float4 v0 = *reinterpret_cast<const float4*>(stg + pr * 32 + (((2 * u) ^ (pr & 7)) << 2));
                    const float4 v1 = *reinterpret_cast<const float4*>(stg + pr * 32 + (((2 * u + 1) ^ (pr & 7)) << 2));
                    const unsigned w4[4] = {rr[it].x, rr[it].y, rr[it].z, rr[it].w};
                    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                    unsigned pk[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        pk[k] = relu2_bf16(v[2 * k] + __uint_as_float(w4[k] << 16), v[2 * k + 1] + __uint_as_float(w4[k] & 0xffff0000u));
                    *reinterpret_cast<uint4*>(Y + pix(i * 32 + pr) + cofs) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                }
                if (i < 3) { rr[0] = rn[0]; rr[1] = rn[1]; }
            }
        }
        stamp(6);
    }
}

// ---------------------------------------------------------------- res4, half-frame tile: small batches
// The frame kernel above gives a frame to ONE workgroup, so a launch of B frames occupies B of the 256 CUs: an evaluation-size
// batch (a sequence's 50 - 130 (frame, object) units, or one half of it after the two-stream split) leaves most of the chip idle for
// the five res4 identity blocks.  Here a workgroup owns an 8 x 16-pixel HALF frame (10 x 16 halo = 180 rows with the two
// out-of-frame columns zero, 6 pixel tiles): 2 B workgroups per launch.  Dataflow of bneck_halo_kernel with C = 256 (wave =
// channel tile in phases A and B, = output-channel tile of a chunk in phase C); per pixel the arithmetic and its order are those
// of the frame kernel, so a frame's result does not depend on which of the two ran it (bit-identical: tests).
// LDS: [0, 73728) x ring (3 x 24 KB) -> t1 (4 slices x 192 rows) [0, 98304) -> t2 (4 x 128 rows) [0, 65536) | staging [65536, 98304);
//      [98304, 104448) biases.
__global__ __launch_bounds__(512) void bneck_half16_kernel(BneckWideArgs p) {
    constexpr int C = 256, CIN = 1024, HW = 16, BTY = 8, HTX = 18, HR = 180, MH = 192, NGA = 23;
    constexpr int TPF = HW / BTY;                    // 2 tiles per frame
    constexpr int NSL = 4;
    constexpr int SLOT = 24576;
    constexpr int T1S = MH * ROWB, T2S = 128 * ROWB;
    constexpr int STG_OFF = NSL * T2S, BIAS_OFF = NSL * T1S, LDS_BYTES = BIAS_OFF + 6 * C * 4;       // 65536, 98304, 104448
    static_assert(3 * SLOT <= NSL * T1S && STG_OFF + 8 * 4096 <= BIAS_OFF && LDS_BYTES <= WIDE_LDS, "LDS map");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhalf = lane >> 5;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int b = L / TPF, tl = L - b * TPF;
    const int y0 = tl * BTY, x0 = 0;
    const bf16_t* X = static_cast<const bf16_t*>(p.x) + (size_t)b * HW * HW * CIN;
    bf16_t* Y = static_cast<bf16_t*>(p.y) + (size_t)b * HW * HW * CIN;
    const bf16_t* zeros = static_cast<const bf16_t*>(p.zeros);
    const int ctw = wave;
    *reinterpret_cast<float*>(lds + BIAS_OFF + tid * 4) = tid < C ? p.ba[tid] : p.bb[tid - C];
    *reinterpret_cast<float*>(lds + BIAS_OFF + 2 * C * 4 + tid * 4) = p.bc[tid];                     // visible after phase A's barriers
    *reinterpret_cast<float*>(lds + BIAS_OFF + 2 * C * 4 + (tid + 512) * 4) = p.bc[tid + 512];
    uint4 wn[4];                                     // first weight fragments of the next phase, requested one phase early

    // ================================================================ phase A: t1 = relu(Wa x + ba) on the halo, K = CIN
    {
        constexpr int NK = CIN / 64;
        f32x16 acc[6];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const int rsub = lane >> 3, cpos = lane & 7;
        const int np = (NGA - wave + 7) / 8;         // row groups of this wave: wave, wave + 8, wave + 16 (< 23)
        const bf16_t* xsrc[3];
        unsigned okmask = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int g = wave + 8 * i;
            const int hr = g * 8 + rsub;
            const int hy = hr / HTX, hx = hr - hy * HTX;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const bool ok = g < NGA && hr < HR && y >= 0 && y < HW && x >= 0 && x < HW;
            xsrc[i] = ok ? X + ((size_t)y * HW + x) * CIN + (cpos ^ ((hr >> 1) & 7)) * 8 : zeros;
            okmask |= ok ? (1u << i) : 0u;
        }
        auto issue_x = [&](int kt) {
            unsigned char* sb = lds + (kt % 3) * SLOT;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int g = wave + 8 * i;
                if (g < NGA) dma16(xsrc[i] + (((okmask >> i) & 1u) ? kt * 64 : 0), sb + g * 1024);
            }
        };
        u32x4 wq[2][4];
        auto load_w = [&](int kt, int set) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(wq[set][ks]) : "v"(wfrag(p.fa, ctw, CIN / 16, kt * 4 + ks, lane)) : "memory");
        };
        load_w(0, 0);
        issue_x(0);
        issue_x(1);
        for (int kt2 = 0; kt2 < NK; kt2 += 2)
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const int kt = kt2 + par;
            if (kt + 1 < NK) wait_vmcnt_n(np); else wait_vmcnt<0>();   // only X(kt+1) is younger than W(kt)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + 1 < NK) load_w(kt + 1, par ^ 1);
            if (kt + 2 < NK) issue_x(kt + 2);        // slot (kt+2) % 3 == slot of tile kt-1: done for every wave
            const unsigned xb = lds_base + (kt % 3) * SLOT;
            u32x4 pf[6];
            const unsigned xrow = xb + lrow * ROWB;
            auto rd = [&](int ks, int half) {
                const unsigned a = xrow + (((2 * ks + lhalf) ^ ((lrow >> 1) & 7)) << 4);
                if (half == 0) { pf[0] = lds_read_b128_o<0>(a); pf[1] = lds_read_b128_o<4096>(a); pf[2] = lds_read_b128_o<8192>(a); }
                else { pf[3] = lds_read_b128_o<12288>(a); pf[4] = lds_read_b128_o<16384>(a); pf[5] = lds_read_b128_o<20480>(a); }
            };
            rd(0, 0);
            rd(0, 1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const u32x4 w = wq[par][ks];
                lgkm_wait<3>();                      // half 0 landed (half 1 may be in flight)
#pragma unroll
                for (int t = 0; t < 3; ++t) acc[t] = mfma_bf16(w, pf[t], acc[t]);
                if (ks < 3) rd(ks + 1, 0);
                if (ks < 3) lgkm_wait<3>(); else lgkm_wait<0>();
#pragma unroll
                for (int t = 3; t < 6; ++t) acc[t] = mfma_bf16(w, pf[t], acc[t]);
                if (ks < 3) rd(ks + 1, 1);
            }
        }
        __builtin_amdgcn_s_barrier();                // the ring is dead: its space becomes t1
        asm volatile("" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wn[ks] = *wfrag(p.fb, ctw, 9 * C / 16, ks, lane);
        float4 bq[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const float4*>(lds + BIAS_OFF + (ctw * 32 + 8 * g + 4 * lhalf) * 4);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int hr = i * 32 + lrow;
            const int hy = hr / HTX, hx = hr - hy * HTX;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const bool in = y >= 0 && y < HW && x >= 0 && x < HW;
            if (hr < HR) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2 pk;
                    pk.x = relu2_bf16(acc[i][4 * g] + bq[g].x, acc[i][4 * g + 1] + bq[g].y) & (in ? 0xffffffffu : 0u);
                    pk.y = relu2_bf16(acc[i][4 * g + 2] + bq[g].z, acc[i][4 * g + 3] + bq[g].w) & (in ? 0xffffffffu : 0u);
                    lds_write_b64(lds_base + (ctw >> 1) * T1S + hr * ROWB + ((((ctw & 1) * 4 + g) ^ ((hx >> 1) & 7)) << 4) + 8 * lhalf, pk);   // column key: see bneck_halo_kernel
                }
            }
        }
        lds_wait();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

    // ================================================================ phase B: t2 = relu(Wb (*) t1 + bb), 9 taps x NSL slices
    {
        constexpr int KSB = 9 * C / 16, NSTEP = 9 * NSL;
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        int hb[4];                                   // halo row of this lane's output pixel at tap (0,0)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = i * 32 + lrow;
            hb[i] = (q >> 4) * HTX + (q & 15);
        }
#pragma unroll 1
        for (int step = 0; step < NSTEP; ++step) {   // step = tap * NSL + slice
            const int tap = step / NSL, sl = step - tap * NSL;
            u32x4 wc[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) wc[ks] = as_u32x4(wn[ks]);
            if (step + 1 < NSTEP) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) wn[ks] = *wfrag(p.fb, ctw, KSB, (step + 1) * 4 + ks, lane);
            }
            unsigned rowa[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int hr = hb[i] + (tap / 3) * HTX + (tap % 3);
                rowa[i] = lds_base + sl * T1S + hr * ROWB;
            }
            const unsigned rkey = (((lrow & 15) + tap % 3) >> 1) & 7;    // column key (see bneck_halo_kernel)
            u32x4 pf[4];
            auto rd = [&](int ks, int half) {
                const int ch = 2 * ks + lhalf;
#pragma unroll
                for (int i = 0; i < 2; ++i) pf[half * 2 + i] = lds_read_b128(rowa[half * 2 + i] + ((ch ^ rkey) << 4));
            };
            rd(0, 0);
            rd(0, 1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                lgkm_wait<2>();
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i] = mfma_bf16(wc[ks], pf[i], acc[i]);
                if (ks < 3) rd(ks + 1, 0);
                if (ks < 3) lgkm_wait<2>(); else lgkm_wait<0>();
#pragma unroll
                for (int i = 2; i < 4; ++i) acc[i] = mfma_bf16(wc[ks], pf[i], acc[i]);
                if (ks < 3) rd(ks + 1, 1);
            }
        }
        __builtin_amdgcn_s_barrier();                // every wave is done reading t1: t2 takes its place
        asm volatile("" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wn[ks] = *wfrag(p.fc, wave, C / 16, ks, lane);
        float4 bq[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const float4*>(lds + BIAS_OFF + (C + ctw * 32 + 8 * g + 4 * lhalf) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px = i * 32 + lrow;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 pk;
                pk.x = relu2_bf16(acc[i][4 * g] + bq[g].x, acc[i][4 * g + 1] + bq[g].y);
                pk.y = relu2_bf16(acc[i][4 * g + 2] + bq[g].z, acc[i][4 * g + 3] + bq[g].w);
                lds_write_b64(lds_base + (ctw >> 1) * T2S + px * ROWB + ((((ctw & 1) * 4 + g) ^ ((px >> 1) & 7)) << 4) + 8 * lhalf, pk);
            }
        }
        lds_wait();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

    // ================================================================ phase C: y = relu(Wc t2 + bc + x), 4 chunks of 8 channel tiles
    {
        constexpr int KSC = C / 16, NCH = (CIN / 32) / 8, NSTEP = NCH * NSL;
        f32x16 acc[4];
        float* stg = reinterpret_cast<float*>(lds + STG_OFF + wave * 4096);
        const int u = lane & 3, prr = lane >> 2;
        auto pix = [&](int q) { return (size_t)((y0 + (q >> 4)) * HW + x0 + (q & 15)) * CIN; };   // tile pixel q -> frame offset
#pragma unroll 1
        for (int chunk = 0; chunk < NCH; ++chunk) {
            const int ct = chunk * 8 + wave;
            const size_t cofs = (size_t)ct * 32 + 8 * u;
            uint4 rr[2];                             // residual of the first pixel tile: in flight under the chunk's MFMAs
#pragma unroll
            for (int it = 0; it < 2; ++it) rr[it] = *reinterpret_cast<const uint4*>(X + pix(it * 16 + prr) + cofs);
            {
                float4 bq[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const float4*>(lds + BIAS_OFF + (2 * C + ct * 32 + 8 * g + 4 * lhalf) * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        acc[i][4 * g] = bq[g].x; acc[i][4 * g + 1] = bq[g].y; acc[i][4 * g + 2] = bq[g].z; acc[i][4 * g + 3] = bq[g].w;
                    }
            }
#pragma unroll 1
            for (int sl = 0; sl < NSL; ++sl) {
                u32x4 wc[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) wc[ks] = as_u32x4(wn[ks]);
                const int nxt = chunk * NSL + sl + 1;
                if (nxt < NSTEP) {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) wn[ks] = *wfrag(p.fc, (nxt / NSL) * 8 + wave, KSC, (nxt % NSL) * 4 + ks, lane);
                }
                const unsigned tb = lds_base + sl * T2S;
                u32x4 pf[4];
                const unsigned trow = tb + lrow * ROWB;
                auto rd = [&](int ks, int half) {
                    const unsigned a = trow + (((2 * ks + lhalf) ^ ((lrow >> 1) & 7)) << 4);
                    if (half == 0) { pf[0] = lds_read_b128_o<0>(a); pf[1] = lds_read_b128_o<4096>(a); }
                    else { pf[2] = lds_read_b128_o<8192>(a); pf[3] = lds_read_b128_o<12288>(a); }
                };
                rd(0, 0);
                rd(0, 1);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    lgkm_wait<2>();
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[i] = mfma_bf16(wc[ks], pf[i], acc[i]);
                    if (ks < 3) rd(ks + 1, 0);
                    if (ks < 3) lgkm_wait<2>(); else lgkm_wait<0>();
#pragma unroll
                    for (int i = 2; i < 4; ++i) acc[i] = mfma_bf16(wc[ks], pf[i], acc[i]);
                    if (ks < 3) rd(ks + 1, 1);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int slot = (2 * g + lhalf) ^ (lrow & 7);
                    *reinterpret_cast<float4*>(stg + lrow * 32 + slot * 4) = make_float4(acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]);
                }
                uint4 rn[2];
                if (i < 3) {
#pragma unroll
                    for (int it = 0; it < 2; ++it) rn[it] = *reinterpret_cast<const uint4*>(X + pix((i + 1) * 32 + it * 16 + prr) + cofs);
                }
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int pr = it * 16 + prr;
                    const float4 v0 = *reinterpret_cast<const float4*>(stg + pr * 32 + (((2 * u) ^ (pr & 7)) << 2));
                    const float4 v1 = *reinterpret_cast<const float4*>(stg + pr * 32 + (((2 * u + 1) ^ (pr & 7)) << 2));
                    const unsigned w4[4] = {rr[it].x, rr[it].y, rr[it].z, rr[it].w};
                    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                    unsigned pk[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        pk[k] = relu2_bf16(v[2 * k] + __uint_as_float(w4[k] << 16), v[2 * k + 1] + __uint_as_float(w4[k] & 0xffff0000u));
                    *reinterpret_cast<uint4*>(Y + pix(i * 32 + pr) + cofs) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                }
                if (i < 3) { rr[0] = rn[0]; rr[1] = rn[1]; }
            }
        }
    }
}

// ---------------------------------------------------------------- res2, small tile: two workgroups per CU
// The 16x16-tile kernel above runs ONE workgroup per CU, so its memory phases (x halo in, residual in / y out) and its
// MFMA phases alternate instead of overlapping: 37k cycles per tile against ~10k of MFMA issue and ~18k of HBM time.
// Here a workgroup owns an 8 x 16-pixel tile (10 x 18 halo = 180 rows, 6 pixel tiles), needs 72 KB of LDS and <= 128
// VGPRs, and two of them share a CU: one is in the matrix cores while the other waits on memory.
//   phase A  wave = (channel tile w & 1, pixel tiles g, g + 4 of the 6 with g = w >> 1); x through a 3-slot LDS-DMA ring
//   phase B  wave = (channel tile w & 1, pixel-tile pair (w >> 1) & 1, K half w >> 2): 18 of the 36 k-steps each, so a
//            Wb fragment serves two MFMAs; the upper K half hands its partial sums over through LDS (fp32)
//   phase C  wave = output-channel tile w, 4 pixel tiles; store pass as above
// LDS: [0, 73728) ring (3 x 24 KB) -> t1 [0, 23040) -> t2 [24576, 40960) ; [40960, 73728) partial sums, then staging.
//
// TIN / ND (conv1 forwarding): with TIN the workgroup does not read the 256-channel x halo (92 KB per tile) to compute t1 on
// it; it loads the 64-channel t1 halo (23 KB) that the previous block's phase D left in HBM.  With ND > 0 the store pass also
// parks the finished y tile in LDS (two halves of 64 pixels x 256 channels, 32 KB, over the dead t1 / t2 / x images) and phase D
// applies the NEXT block's conv1 (256 -> ND, K = 256) to it: t1out = relu(Wd y + bd).  Per tile: 23 + 64 (residual) KB in,
// 64 + 16 KB out instead of 92 + 64 in, 64 out; no halo recompute; and res3's first 1x1 layer (a 0.8 GB HBM pass) disappears.
// YS2: the consumer of y is a stride-2 1x1 convolution only (res3's downsample; its conv1 was forwarded), so only the even
// pixels are written, compactly, as [B,32,32,256]: 16 instead of 64 KB per tile.
template <bool DS, bool TIN, int ND, bool YS2 = false>
__global__ __launch_bounds__(512, 4) void bneck_halo64s_kernel(BneckWideArgs p) {
    static_assert(!(DS && TIN) && (ND == 0 || ND == 64 || ND == 128) && (!YS2 || ND > 0), "variants");
    constexpr int C = 64, CIN = DS ? 64 : 256, COUT = 256, NKA = CIN / 64, HW = 64, BTY = 8, BTX = 16, HTX = 18, HR = 180, NGA = 23;
    constexpr int SLOT = 24576, T1_OFF = DS ? SLOT : 0, T2_OFF = SLOT, STG_OFF = DS ? 2 * SLOT : 40960;
    // biases wait in LDS (a global load at the head of every epilogue would expose an L2 round trip each time): behind the
    // staging area, or (DS: no room left under 80 KB) in the never-used tails of ring slot 0 and of the t1 image
    constexpr int BAB_OFF = DS ? SLOT + 23040 : STG_OFF + 32768, BC_OFF = DS ? 23552 : BAB_OFF + 512;
    constexpr int LDS_BYTES = STG_OFF + 32768 + (DS ? 0 : 1536);   // 75264 | 81920: two workgroups per CU either way
    constexpr int TPXX = HW / BTX, TPF = (HW / BTY) * TPXX;   // 4 tiles across, 32 per frame
    // t1 rows: the identity kernel pads them to 144 B instead of XOR-swizzling the 16-byte chunks — consecutive rows then sit
    // 36 banks apart (conflict-free for the 16-lane groups of a ds_read_b128), and a phase-B fragment address becomes
    // "row base + compile-time offset": no per-read arithmetic (216 vector instructions per tile in a kernel that is
    // instruction-issue-bound).  180 x 144 B = 25 920 B runs 1 344 B into the t2 area, which is only written once every
    // wave is done reading t1.  (DS keeps the swizzle: its LDS map has no spare bytes next to t1.)
    constexpr bool PAD = !DS;
    constexpr int T1R = PAD ? 144 : ROWB;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhalf = lane >> 5;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int b = L / TPF, tl = L - b * TPF;
    const int y0 = (tl / TPXX) * BTY, x0 = (tl % TPXX) * BTX;
    const bf16_t* X = static_cast<const bf16_t*>(p.x) + (size_t)b * HW * HW * CIN;
    bf16_t* Y = static_cast<bf16_t*>(p.y) + (size_t)b * HW * HW * COUT;
    const bf16_t* zeros = static_cast<const bf16_t*>(p.zeros);
    const int ctw = wave & 1;
    auto stamp = [&](int k) {
        if (p.ts && tid == 0) p.ts[(size_t)blockIdx.x * 8 + k] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);
    float bias_v = 0.f;                              // thread t: ba[t] | bb[t - 64] | bc[t - 128]
    if (tid < 64) bias_v = p.ba[tid]; else if (tid < 128) bias_v = p.bb[tid - 64]; else if (tid < 384) bias_v = p.bc[tid - 128];
    uint4 wnb[3], wnb2[3], wnc[4], wd[4];            // weight fragments of phases B (two alternating sets) and C, requested early
    float4 bq[4];                                    // conv1 bias of this wave's channel tile (registers: used before the LDS copy is visible)
#pragma unroll
    for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const float4*>(p.ba + ctw * 32 + 8 * g + 4 * lhalf);

    // ================================================================ phase A
    if constexpr (TIN) {
        // t1 arrives from the previous block (its phase D): 180 halo rows x 128 B, rows outside the frame are the 3x3's zero
        // padding.  1440 16-byte chunks over 512 threads, through registers into the padded-row image.
        const bf16_t* T1 = static_cast<const bf16_t*>(p.t1in) + (size_t)b * HW * HW * C;
        uint4 tv[3];
        unsigned ta[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int id = tid + 512 * i;
            const int hr = id >> 3, ch = id & 7;
            const int hy = hr / HTX, hx = hr - hy * HTX;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const bool ok = hr < HR && y >= 0 && y < HW && x >= 0 && x < HW;
            tv[i] = ok ? *reinterpret_cast<const uint4*>(T1 + ((size_t)y * HW + x) * C + ch * 8) : make_uint4(0, 0, 0, 0);
            ta[i] = hr < HR ? lds_base + T1_OFF + hr * T1R + ch * 16 : 0xffffffffu;
        }
        if (tid < 128) *reinterpret_cast<float*>(lds + BAB_OFF + tid * 4) = bias_v;
        else if (tid < 384) *reinterpret_cast<float*>(lds + BC_OFF + (tid - 128) * 4) = bias_v;
#pragma unroll
        for (int j = 0; j < 3; ++j) wnb[j] = *wfrag(p.fb, ctw, 36, (wave >> 2) * 18 + j, lane);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (ta[i] != 0xffffffffu) {
                u32x4 v = {tv[i].x, tv[i].y, tv[i].z, tv[i].w};
                asm volatile("ds_write_b128 %0, %1" ::"v"(ta[i]), "v"(v) : "memory");
            }
        stamp(1);
        lds_wait();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stamp(2);
    } else {
        const int grp = wave >> 1;
        const bool two = grp < 2;                    // pixel tiles grp and grp + 4 (< 6)
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const int rsub = lane >> 3, cpos = lane & 7;
        const int np = wave < 7 ? 3 : 2;             // row groups wave, wave + 8, wave + 16 (< 23)
        const bf16_t* xsrc[3];
        unsigned okmask = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int g = wave + 8 * i;
            const int hr = g * 8 + rsub;
            const int hy = hr / HTX, hx = hr - hy * HTX;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const bool ok = g < NGA && hr < HR && y >= 0 && y < HW && x >= 0 && x < HW;
            xsrc[i] = ok ? X + ((size_t)y * HW + x) * CIN + (cpos ^ ((hr >> 1) & 7)) * 8 : zeros;
            okmask |= ok ? (1u << i) : 0u;
        }
        auto issue_x = [&](int kt) {
            unsigned char* sb = lds + (kt % 3) * SLOT;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int g = wave + 8 * i;
                if (g < NGA) dma16(xsrc[i] + (((okmask >> i) & 1u) ? kt * 64 : 0), sb + g * 1024);
            }
        };
        u32x4 wq[2][4];                              // Wa k-steps of K-tile kt in set kt & 1
        auto load_w = [&](int kt) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(wq[kt & 1][ks]) : "v"(wfrag(p.fa, ctw, CIN / 16, kt * 4 + ks, lane)) : "memory");
        };
        load_w(0);
        if (NKA > 1) load_w(1);
        issue_x(0);
        if (NKA > 1) { issue_x(1); issue_x(2); }
#pragma unroll
        for (int kt = 0; kt < NKA; ++kt) {
            // in-order queue: W0 W1 X0 X1 X2 | W2 X3 (iteration 1) | W3 (iteration 2): younger than {W(kt), X(kt)} are
            // X1 X2 | X2 | X3 | nothing
            if (NKA == 1) wait_vmcnt<0>(); else
            if (kt == 0) wait_vmcnt_n(2 * np); else if (kt < 3) wait_vmcnt_n(np); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt == 1 || kt == 2) load_w(kt + 1); // set (kt+1) & 1 held K-tile kt-1
            if (kt == 1) issue_x(3);                 // slot 0: every wave is past K-tile 0
            const unsigned xb = lds_base + (kt % 3) * SLOT;
            u32x4 pf[2][2];
            const unsigned xrow = xb + (grp * 32 + lrow) * ROWB;       // tile grp + 4 = + 16384 bytes, same swizzle key
            auto rd = [&](int ks, int buf) {
                const unsigned a = xrow + (((2 * ks + lhalf) ^ ((lrow >> 1) & 7)) << 4);
                pf[buf][0] = lds_read_b128_o<0>(a);
                if (two) pf[buf][1] = lds_read_b128_o<16384>(a);
            };
            rd(0, 0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks < 3) rd(ks + 1, (ks + 1) & 1);
                if (ks < 3) { if (two) lgkm_wait<2>(); else lgkm_wait<1>(); } else lgkm_wait<0>();
                acc[0] = mfma_bf16(wq[kt & 1][ks], pf[ks & 1][0], acc[0]);
                if (two) acc[1] = mfma_bf16(wq[kt & 1][ks], pf[ks & 1][1], acc[1]);
            }
        }
        stamp(1);
        __builtin_amdgcn_s_barrier();                // the ring is dead: slot 0 becomes t1 (DS: x stays, t1 goes to slot 1)
        asm volatile("" ::: "memory");
        if (tid < 128) *reinterpret_cast<float*>(lds + BAB_OFF + tid * 4) = bias_v;
        else if (tid < 384) *reinterpret_cast<float*>(lds + BC_OFF + (tid - 128) * 4) = bias_v;
#pragma unroll
        for (int j = 0; j < 3; ++j) wnb[j] = *wfrag(p.fb, ctw, 36, (wave >> 2) * 18 + j, lane);

#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i == 1 && !two) break;
            const int hr = (grp + 4 * i) * 32 + lrow;
            const int hy = hr / HTX, hx = hr - hy * HTX;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const bool in = y >= 0 && y < HW && x >= 0 && x < HW;
            if (hr < HR) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2 pk;
                    pk.x = relu2_bf16(acc[i][4 * g] + bq[g].x, acc[i][4 * g + 1] + bq[g].y) & (in ? 0xffffffffu : 0u);
                    pk.y = relu2_bf16(acc[i][4 * g + 2] + bq[g].z, acc[i][4 * g + 3] + bq[g].w) & (in ? 0xffffffffu : 0u);
                    if (PAD) lds_write_b64(lds_base + T1_OFF + hr * T1R + ((ctw * 4 + g) << 4) + 8 * lhalf, pk);
                    else lds_write_b64(lds_base + T1_OFF + hr * ROWB + (((ctw * 4 + g) ^ ((hr >> 1) & 7)) << 4) + 8 * lhalf, pk);
                }
            }
        }
        lds_wait();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stamp(2);
    }

    // ================================================================ phase B: 36 k-steps (tap * 4 + ks), split over wave >> 2
    {
        const int pp = (wave >> 1) & 1, kh = wave >> 2;
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        int hb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = (pp * 2 + i) * 32 + lrow;
            hb[i] = (q >> 4) * HTX + (q & 15);
        }
        // the K half is a compile-time constant inside (one runtime branch on wave >> 2): taps and k-steps of every read
        // are then constants, the weight sets alternate without register copies
        unsigned rb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) rb[i] = lds_base + T1_OFF + hb[i] * T1R + lhalf * 16;
        auto run_half = [&](auto khc) {
            constexpr int K0 = decltype(khc)::value * 18;
            u32x4 pf[2][2];
            auto rd = [&](auto kc, int buf) {
                constexpr int kstep = K0 + decltype(kc)::value;
                constexpr int tap = kstep >> 2, ks = kstep & 3, toff = (tap / 3) * HTX + (tap % 3);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (PAD) pf[buf][i] = lds_read_b128_o<toff * T1R + ks * 32>(rb[i]);
                    else {
                        const int hr = hb[i] + toff, ch = 2 * ks + lhalf;
                        pf[buf][i] = lds_read_b128(lds_base + T1_OFF + hr * ROWB + ((ch ^ ((hr >> 1) & 7)) << 4));
                    }
                }
            };
            auto chunk = [&](auto cc, uint4 (&wcur)[3], uint4 (&wnext)[3]) {      // 3 k-steps; the next 3 fragments in flight
                constexpr int C3 = decltype(cc)::value;
                if (C3 < 5) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) wnext[j] = *wfrag(p.fb, ctw, 36, K0 + (C3 + 1) * 3 + j, lane);
                }
                auto step = [&](auto jc) {
                    constexpr int J = decltype(jc)::value, KS = C3 * 3 + J;     // k-step of this half, 0..17
                    if constexpr (KS < 17) rd(std::integral_constant<int, KS + 1>{}, (KS + 1) & 1);
                    if (KS < 17) lgkm_wait<2>(); else lgkm_wait<0>();
                    const u32x4 w = as_u32x4(wcur[J]);
                    acc[0] = mfma_bf16(w, pf[KS & 1][0], acc[0]);
                    acc[1] = mfma_bf16(w, pf[KS & 1][1], acc[1]);
                };
                step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
            };
            rd(std::integral_constant<int, 0>{}, 0);
            chunk(std::integral_constant<int, 0>{}, wnb, wnb2);
            chunk(std::integral_constant<int, 1>{}, wnb2, wnb);
            chunk(std::integral_constant<int, 2>{}, wnb, wnb2);
            chunk(std::integral_constant<int, 3>{}, wnb2, wnb);
            chunk(std::integral_constant<int, 4>{}, wnb, wnb2);
            chunk(std::integral_constant<int, 5>{}, wnb2, wnb);
        };
        if (kh) run_half(std::integral_constant<int, 1>{}); else run_half(std::integral_constant<int, 0>{});
        stamp(3);
        constexpr int KSC = DS ? 8 : 4;              // DS: [conv3 | downsample] concatenated along K, second half reads x
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wnc[ks] = *wfrag(p.fc, wave, KSC, ks, lane);
        if (DS) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) wd[ks] = *wfrag(p.fc, wave, KSC, 4 + ks, lane);
        }
        // upper K half -> partial sums to LDS; lower K half adds them, + bias, ReLU -> t2
        float* scr = reinterpret_cast<float*>(lds + STG_OFF + (wave & 3) * 8192);
        if (kh) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(scr + ((i * 4 + g) * 64 + lane) * 4) = make_float4(acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]);
        }
        __syncthreads();
        if (!kh) {
            float4 bq[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const float4*>(lds + BAB_OFF + 256 + (ctw * 32 + 8 * g + 4 * lhalf) * 4);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int px = (pp * 2 + i) * 32 + lrow;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 o = *reinterpret_cast<const float4*>(scr + ((i * 4 + g) * 64 + lane) * 4);
                    u32x2 pk;
                    pk.x = relu2_bf16(acc[i][4 * g] + o.x + bq[g].x, acc[i][4 * g + 1] + o.y + bq[g].y);
                    pk.y = relu2_bf16(acc[i][4 * g + 2] + o.z + bq[g].z, acc[i][4 * g + 3] + o.w + bq[g].w);
                    lds_write_b64(lds_base + T2_OFF + px * ROWB + (((ctw * 4 + g) ^ ((px >> 1) & 7)) << 4) + 8 * lhalf, pk);
                }
            }
        }
        __syncthreads();
        stamp(4);
    }

    // ================================================================ phase C: wave = output-channel tile
    {
        f32x16 acc[4];
        float* stg = reinterpret_cast<float*>(lds + STG_OFF + wave * 4096);
        const int u = lane & 3, prr = lane >> 2;
        auto pix = [&](int q) { return (size_t)((y0 + (q >> 4)) * HW + x0 + (q & 15)) * COUT; };

        const size_t cofs = (size_t)wave * 32 + 8 * u;
        // element (pixel tile i, half it) of this lane = tile pixel i*32 + it*16 + prr = frame row y0 + 2i + it, column x0 + prr:
        // one base pointer + a row pitch per step instead of a 64-bit index computation per access
        constexpr size_t RPITCH = (size_t)HW * COUT;
        const bf16_t* xrow = X + pix(prr) + cofs;
        bf16_t* yrow = Y + pix(prr) + cofs;
        if constexpr (YS2)                           // compact even-pixel output: row (y0 + 2i) / 2, column (x0 + prr) / 2
            yrow = static_cast<bf16_t*>(p.y) + (size_t)b * (HW / 2) * (HW / 2) * COUT + ((size_t)(y0 / 2) * (HW / 2) + (x0 + prr) / 2) * COUT + cofs;
        uint4 rr[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) rr[it] = DS ? make_uint4(0, 0, 0, 0) : *reinterpret_cast<const uint4*>(xrow + it * RPITCH);
        {
            float4 bq[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const float4*>(lds + BC_OFF + (wave * 32 + 8 * g + 4 * lhalf) * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    acc[i][4 * g] = bq[g].x; acc[i][4 * g + 1] = bq[g].y; acc[i][4 * g + 2] = bq[g].z; acc[i][4 * g + 3] = bq[g].w;
                }
        }
        const unsigned tb = lds_base + T2_OFF;
        u32x4 pf[4];
        const unsigned trow = tb + lrow * ROWB;
        auto rd = [&](int ks, int half) {
            const unsigned a = trow + (((2 * ks + lhalf) ^ ((lrow >> 1) & 7)) << 4);
            if (half == 0) { pf[0] = lds_read_b128_o<0>(a); pf[1] = lds_read_b128_o<4096>(a); }
            else { pf[2] = lds_read_b128_o<8192>(a); pf[3] = lds_read_b128_o<12288>(a); }
        };
        rd(0, 0);
        rd(0, 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const u32x4 w = as_u32x4(wnc[ks]);
            lgkm_wait<2>();
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = mfma_bf16(w, pf[i], acc[i]);
            if (ks < 3) rd(ks + 1, 0);
            if (ks < 3) lgkm_wait<2>(); else lgkm_wait<0>();
#pragma unroll
            for (int i = 2; i < 4; ++i) acc[i] = mfma_bf16(w, pf[i], acc[i]);
            if (ks < 3) rd(ks + 1, 1);
        }
        if (DS) {                                    // + Wd x on the tile's centre pixels, still in ring slot 0
            unsigned xrow[4], xkey[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = i * 32 + lrow;
                const int hr = ((q >> 4) + 1) * HTX + (q & 15) + 1;
                xrow[i] = lds_base + hr * ROWB;
                xkey[i] = (hr >> 1) & 7;
            }
            auto rdx = [&](int ks, int half) {
                const int ch = 2 * ks + lhalf;
#pragma unroll
                for (int i = 0; i < 2; ++i) pf[half * 2 + i] = lds_read_b128(xrow[half * 2 + i] + ((ch ^ xkey[half * 2 + i]) << 4));
            };
            rdx(0, 0);
            rdx(0, 1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const u32x4 w = as_u32x4(wd[ks]);
                lgkm_wait<2>();
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i] = mfma_bf16(w, pf[i], acc[i]);
                if (ks < 3) rdx(ks + 1, 0);
                if (ks < 3) lgkm_wait<2>(); else lgkm_wait<0>();
#pragma unroll
                for (int i = 2; i < 4; ++i) acc[i] = mfma_bf16(w, pf[i], acc[i]);
                if (ks < 3) rdx(ks + 1, 1);
            }
        }
        stamp(5);
        // phase D (ND > 0): the next block's conv1 on this tile's y, half a tile (64 pixels) at a time.  The y half sits in LDS
        // as four 64-channel slices of the swizzled [64 rows][128 B] image at [0, 32768): t1 / t2 / x are dead by now.
        constexpr int YS = 64 * ROWB;                // one slice of the half image
        bf16_t* T1O = ND > 0 ? static_cast<bf16_t*>(p.t1out) + (size_t)b * HW * HW * ND : nullptr;
        auto phase_d = [&](int half) {
            lds_wait();
            __builtin_amdgcn_s_barrier();            // every wave's part of the y half is in LDS
            asm volatile("" ::: "memory");
            constexpr int NT = (ND / 32) * 2;        // output tiles of the half: channel tiles x 2 pixel tiles
            if (wave < NT) {
                const int ct = wave % (ND / 32), pt = wave / (ND / 32);
                f32x16 ad;                           // from zero, bias added after the sum: the order of the kernels this replaces
#pragma unroll
                for (int r = 0; r < 16; ++r) ad[r] = 0.f;
                const int row = pt * 32 + lrow;
                // addresses as (one register + immediates): left to itself hipcc computes all 16 weight pointers and 16 LDS
                // addresses ahead of the store pass and spills them (25 dwords of scratch traffic per tile)
                unsigned ya[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) ya[ks] = lds_base + row * ROWB + (((2 * ks + lhalf) ^ ((row >> 1) & 7)) << 4);
                const uint4* wb = wfrag(p.fd, ct, 16, 0, lane);
                asm volatile("" : "+v"(wb), "+v"(ya[0]), "+v"(ya[1]), "+v"(ya[2]), "+v"(ya[3]));
                uint4 wv[2][4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) wv[0][ks] = wb[ks * 64];
                auto slice = [&](auto slc) {
                    constexpr int SL = decltype(slc)::value;
                    if constexpr (SL < 3) {
                        const uint4* wn_ = wb + (SL + 1) * 256;
                        asm volatile("" : "+v"(wn_));
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) wv[(SL + 1) & 1][ks] = wn_[ks * 64];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    u32x4 pd[4];
                    pd[0] = lds_read_b128_o<SL * YS>(ya[0]); pd[1] = lds_read_b128_o<SL * YS>(ya[1]);
                    pd[2] = lds_read_b128_o<SL * YS>(ya[2]); pd[3] = lds_read_b128_o<SL * YS>(ya[3]);
                    lds_wait();
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) ad = mfma_bf16(as_u32x4(wv[SL & 1][ks]), pd[ks], ad);
                    __builtin_amdgcn_sched_barrier(0);
                };
                slice(std::integral_constant<int, 0>{}); slice(std::integral_constant<int, 1>{});
                slice(std::integral_constant<int, 2>{}); slice(std::integral_constant<int, 3>{});
                const int q = half * 64 + row;
                bf16_t* o = T1O + ((size_t)(y0 + (q >> 4)) * HW + x0 + (q & 15)) * ND + ct * 32 + 4 * lhalf;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bq = *reinterpret_cast<const float4*>(p.bd + ct * 32 + 8 * g + 4 * lhalf);
                    uint2 pk;
                    pk.x = relu2_bf16(ad[4 * g] + bq.x, ad[4 * g + 1] + bq.y);
                    pk.y = relu2_bf16(ad[4 * g + 2] + bq.z, ad[4 * g + 3] + bq.w);
                    *reinterpret_cast<uint2*>(o + 8 * g) = pk;
                }
            }
            if (half == 0) {
                __builtin_amdgcn_s_barrier();        // the image is free for the second half
                asm volatile("" ::: "memory");
            }
        };
        if constexpr (ND > 0) {
            __builtin_amdgcn_s_barrier();            // every wave is past its last read of t2 (DS: and of x)
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int slot = (2 * g + lhalf) ^ (lrow & 7);
                *reinterpret_cast<float4*>(stg + lrow * 32 + slot * 4) = make_float4(acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]);
            }
            uint4 rn[2];
            const bool carry = !(ND > 0 && i == 1);  // phase D sits behind item 1: the next residual is requested after it (registers)
            if (i < 3 && carry) {
#pragma unroll
                for (int it = 0; it < 2; ++it) rn[it] = DS ? make_uint4(0, 0, 0, 0) : *reinterpret_cast<const uint4*>(xrow + (2 * (i + 1) + it) * RPITCH);
            }
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int pr = it * 16 + prr;
                const float4 v0 = *reinterpret_cast<const float4*>(stg + pr * 32 + (((2 * u) ^ (pr & 7)) << 2));
                const float4 v1 = *reinterpret_cast<const float4*>(stg + pr * 32 + (((2 * u + 1) ^ (pr & 7)) << 2));
                const unsigned w4[4] = {rr[it].x, rr[it].y, rr[it].z, rr[it].w};
                const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                unsigned pk[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    pk[k] = relu2_bf16(v[2 * k] + __uint_as_float(w4[k] << 16), v[2 * k + 1] + __uint_as_float(w4[k] & 0xffff0000u));
                if constexpr (YS2) {
                    if (it == 0 && !(prr & 1)) *reinterpret_cast<uint4*>(yrow + (size_t)i * (HW / 2) * COUT) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                } else {
                    *reinterpret_cast<uint4*>(yrow + (2 * i + it) * RPITCH) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                }
                if constexpr (ND > 0) {              // the same 8 channels of this pixel into the y half image
                    const int row = (i & 1) * 32 + pr, c16 = wave * 4 + u;
                    const u32x4 v = {pk[0], pk[1], pk[2], pk[3]};
                    const unsigned a = lds_base + (c16 >> 3) * YS + row * ROWB + (((c16 & 7) ^ ((row >> 1) & 7)) << 4);
                    asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(v) : "memory");
                }
            }
            if constexpr (ND > 0) {
                if (i == 1) phase_d(0);
                if (i == 3) phase_d(1);
            }
            if (i < 3 && !carry) {
#pragma unroll
                for (int it = 0; it < 2; ++it) rn[it] = DS ? make_uint4(0, 0, 0, 0) : *reinterpret_cast<const uint4*>(xrow + (2 * (i + 1) + it) * RPITCH);
            }
            if (i < 3) { rr[0] = rn[0]; rr[1] = rn[1]; }
        }
        stamp(6);
    }
}

// ---------------------------------------------------------------- 1x1 convolution in the same style (bf16, Cout % 256 == 0)
// Phase A of the frame kernel as a layer of its own: a workgroup owns 256 output pixels x 256 output channels, the
// pixel operand streams through a 4-slot LDS-DMA ring (32 KB per 64-channel K-tile), wave w streams the fragment-ordered
// weights of channel tile 8*tile_n + w straight into registers, and the accumulators (initialised with the bias) leave
// through the per-wave staging tile with the optional residual.  Against conv_igemm_ws_kernel (256 x 128 tile, weights
// through LDS): half the LDS-DMA bytes per MFMA and no weight ring to synchronise on.  Serves the K-heavy 1x1 layers:
// first-block reductions, conv3 + folded downsample (K-extension: second pixel source x2 sampled at stride2), res5.
// NPT = pixel tiles of 32 per workgroup: 8 (256 pixels) or 4 (round 5: evaluation-size launches whose 256-pixel grid leaves most of the chip
// empty run on 128-pixel tiles - twice the workgroups, two of them per CU at 80 KB of LDS; the K order per output element is the same, so a
// frame's result does not depend on which of the two ran)
template <int NPT>
__global__ __launch_bounds__(512) void conv1x1_wide_kernel(ConvArgs p, const void* __restrict__ fw) {
    constexpr int BMT = NPT * 32, HP = NPT / 2, NDP = NPT / 2;          // rows per tile, pixel tiles per half of a k-step, DMA pieces per wave and K-tile
    // NPT = 4: the per-wave staging tiles of the store pass alias the ring (behind a barrier), 64 KB in all: two workgroups per CU
    constexpr int SLICE = BMT * ROWB, STG_OFF = NPT == 8 ? 4 * SLICE : 0;
    static_assert(NPT == 8 || NPT == 4, "256- or 128-pixel tiles");
    __shared__ __attribute__((aligned(16))) unsigned char lds[NPT == 8 ? 4 * SLICE + 8 * 4096 : 4 * SLICE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhalf = lane >> 5;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int nbn = p.Cout / 256;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = L % nbn, tile_m = p.rev ? (int)gridDim.x / nbn - 1 - L / nbn : L / nbn;
    const int m0 = tile_m * BMT;
    const int M = p.B * p.Ho * p.Wo;
    const bf16_t* X = static_cast<const bf16_t*>(p.x);
    const bf16_t* X2 = static_cast<const bf16_t*>(p.x2);
    const bf16_t* zeros = static_cast<const bf16_t*>(p.zeros);
    const int K = p.Cin + (X2 ? p.Cin2 : 0);
    const int NK = K / 64, NK1 = p.Cin / 64, KS = K / 16;
    const int ct = tile_n * 8 + wave;

    f32x16 acc[NPT];
    {
        float4 bq[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const float4*>(p.bias + ct * 32 + 8 * g + 4 * lhalf);
#pragma unroll
        for (int i = 0; i < NPT; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                acc[i][4 * g] = bq[g].x; acc[i][4 * g + 1] = bq[g].y; acc[i][4 * g + 2] = bq[g].z; acc[i][4 * g + 3] = bq[g].w;
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the bias loads are out of the queue before the counted part starts
    {
        const int rsub = lane >> 3, cpos = lane & 7;
        const bf16_t* xs1[NDP];
        const bf16_t* xs2[NDP];
#pragma unroll
        for (int i = 0; i < NDP; ++i) {
            const int row = (wave * NDP + i) * 8 + rsub;
            const int m = m0 + row;
            const int chunk = (cpos ^ ((row >> 1) & 7)) * 8;
            xs1[i] = xs2[i] = nullptr;
            if (m < M) {
                const int bb = m / (p.Ho * p.Wo);
                const int rem = m - bb * (p.Ho * p.Wo);
                const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                xs1[i] = X + (((size_t)bb * p.H + oy * p.stride) * p.W + ox * p.stride) * p.Cin + chunk;
                if (X2) xs2[i] = X2 + (((size_t)bb * p.H2 + oy * p.stride2) * p.W2 + ox * p.stride2) * p.Cin2 + chunk;
            }
        }
        auto issue_x = [&](int kt) {
#pragma unroll
            for (int i = 0; i < NDP; ++i) {
                const bf16_t* src = kt < NK1 ? (xs1[i] ? xs1[i] + kt * 64 : zeros) : (xs2[i] ? xs2[i] + (kt - NK1) * 64 : zeros);
                dma16(src, lds + (kt & 3) * SLICE + (wave * NDP + i) * 1024);
            }
        };
        u32x4 wq[2][4];
        auto load_w = [&](int kt, int set) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(wq[set][ks]) : "v"(wfrag(fw, ct, KS, kt * 4 + ks, lane)) : "memory");
        };
        u32x4 pf[NPT];
        // queue per wave: W0 X0 X1 | iter kt: W(kt+1) X(kt+2) -> at the top of iter kt only X(kt+1) is younger than W(kt)
        load_w(0, 0);
        issue_x(0);
        if (NK > 1) issue_x(1);
        for (int kt2 = 0; kt2 < NK; kt2 += 2)
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const int kt = kt2 + par;
            if (kt < NK) {
                if (kt + 1 < NK) wait_vmcnt<NDP>(); else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (kt + 1 < NK) load_w(kt + 1, par ^ 1);
                if (kt + 2 < NK) issue_x(kt + 2);
                const unsigned xb = lds_base + (kt & 3) * SLICE;
                const unsigned xrow = xb + lrow * ROWB;
                auto rd = [&](int ks, int half) {
                    const unsigned a = xrow + (((2 * ks + lhalf) ^ ((lrow >> 1) & 7)) << 4);
                    if constexpr (NPT == 8) {
                        if (half == 0) { pf[0] = lds_read_b128_o<0>(a); pf[1] = lds_read_b128_o<4096>(a); pf[2] = lds_read_b128_o<8192>(a); pf[3] = lds_read_b128_o<12288>(a); }
                        else { pf[4] = lds_read_b128_o<16384>(a); pf[5] = lds_read_b128_o<20480>(a); pf[6] = lds_read_b128_o<24576>(a); pf[7] = lds_read_b128_o<28672>(a); }
                    } else {
                        if (half == 0) { pf[0] = lds_read_b128_o<0>(a); pf[1] = lds_read_b128_o<4096>(a); }
                        else { pf[2] = lds_read_b128_o<8192>(a); pf[3] = lds_read_b128_o<12288>(a); }
                    }
                };
                rd(0, 0);
                rd(0, 1);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const u32x4 w = wq[par][ks];
                    lgkm_wait<HP>();
#pragma unroll
                    for (int i = 0; i < HP; ++i) acc[i] = mfma_bf16(w, pf[i], acc[i]);
                    if (ks < 3) rd(ks + 1, 0);
                    if (ks < 3) lgkm_wait<HP>(); else lgkm_wait<0>();
#pragma unroll
                    for (int i = HP; i < NPT; ++i) acc[i] = mfma_bf16(w, pf[i], acc[i]);
                    if (ks < 3) rd(ks + 1, 1);
                }
            }
        }
    }
    // store pass (NPT = 8: the staging area is outside the ring, no barrier needed; NPT = 4: it aliases the ring - every wave must be past
    // its last fragment read)
    if constexpr (NPT == 4) {
        lds_wait();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    float* stg = reinterpret_cast<float*>(lds + STG_OFF + wave * 4096);
    const int u = lane & 3, prr = lane >> 2;
    const bf16_t* R = static_cast<const bf16_t*>(p.res);
    bf16_t* Y = static_cast<bf16_t*>(p.y);
    const size_t cofs = (size_t)ct * 32 + 8 * u;
    auto res_load = [&](int i, uint4 (&r)[2]) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int m = min(m0 + i * 32 + it * 16 + prr, M - 1);
            r[it] = R ? *reinterpret_cast<const uint4*>(R + (size_t)m * p.Cout + cofs) : make_uint4(0, 0, 0, 0);
        }
    };
    uint4 rr[2];
    res_load(0, rr);
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int slot = (2 * g + lhalf) ^ (lrow & 7);
            *reinterpret_cast<float4*>(stg + lrow * 32 + slot * 4) = make_float4(acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]);
        }
        uint4 rn[2];
        if (i < NPT - 1) res_load(i + 1, rn);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int pr = it * 16 + prr;
            const int m = m0 + i * 32 + pr;
            const float4 v0 = *reinterpret_cast<const float4*>(stg + pr * 32 + (((2 * u) ^ (pr & 7)) << 2));
            const float4 v1 = *reinterpret_cast<const float4*>(stg + pr * 32 + (((2 * u + 1) ^ (pr & 7)) << 2));
            const unsigned w4[4] = {rr[it].x, rr[it].y, rr[it].z, rr[it].w};
            const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            unsigned pk[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                pk[k] = act2_bf16(v[2 * k] + __uint_as_float(w4[k] << 16), v[2 * k + 1] + __uint_as_float(w4[k] & 0xffff0000u), p.relu != 0);
            }
            if (m < M) {
                u32x4 ov = {pk[0], pk[1], pk[2], pk[3]};
                if (p.nt & 1) __builtin_nontemporal_store(ov, reinterpret_cast<u32x4*>(Y + (size_t)m * p.Cout + cofs));
                else *reinterpret_cast<u32x4*>(Y + (size_t)m * p.Cout + cofs) = ov;
            }
        }
        if (i < NPT - 1) { rr[0] = rn[0]; rr[1] = rn[1]; }
    }
}

bool conv1x1_wide_ok(const ConvArgs& a) {
    // 256 x 256 tiles: res5's 2048 -> 512 reduction is half a workgroup per frame (128 workgroups at batch 256, half the
    // chip empty) and stays on the 256 x 128 tiles of conv_igemm_ws_kernel.  The rule looks at the layer shape only,
    // never at the batch: a frame's result must not depend on the chunk it was processed in.
    const int wgs = a.Ho * a.Wo * (a.Cout / 256) >= 256 ? 192 : 0;
    return a.KH == 1 && a.KW == 1 && a.pad == 0 && a.Cout % 256 == 0 && a.Cin % 64 == 0 && (!a.x2 || a.Cin2 % 64 == 0) &&
           (a.Cin + (a.x2 ? a.Cin2 : 0)) >= tune_get("WIDE1X1_K", 384) && wgs >= tune_get("WIDE1X1_WGS", 192);
}

void launch_conv1x1_wide(const ConvArgs& a_in, const void* fw, hipStream_t st) {
    ConvArgs a = a_in;
    a.nt = tune_get("NT", 3);
    void* tok = prof_begin(a, 2, st);
    const int M = a.B * a.Ho * a.Wo;
    const int grid = ((M + 255) / 256) * (a.Cout / 256);
    // WIDE_SMALL (off by default): launches of fewer than that many 256-pixel tiles run on 128-pixel tiles, two workgroups per CU.  Measured at
    // 100 units (two streams of 50): + 0.4 ... 1.2 % on one box, - 0.3 % on another with WIDE_SMALL = 160, - 1 % with 256 - inside the noise: the
    // weight stream per pixel doubles and eats what the extra workgroups bring.  Kept for the bit-identity test and further tuning.
    if (grid < tune_get("WIDE_SMALL", 0)) {
        hipLaunchKernelGGL(conv1x1_wide_kernel<4>, dim3(((M + 127) / 128) * (a.Cout / 256)), dim3(512), 0, st, a, fw);
    } else {
        hipLaunchKernelGGL(conv1x1_wide_kernel<8>, dim3(grid), dim3(512), 0, st, a, fw);
    }
    prof_end(tok, st);
}

// [Cout][K] K-major packed bf16 weights -> MFMA-operand order [Cout/32][K/16][64 lanes][8]: lane l of fragment (ct, ks)
// holds row ct*32 + (l & 31), k = ks*16 + 8*(l >> 5) .. +8
__global__ void fragpack_kernel(const bf16_t* __restrict__ w, int Cout, int K, bf16_t* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;     // one 16-B chunk per thread
    const long nchunk = (long)Cout * K / 8;
    if (i >= nchunk) return;
    const int l = (int)(i & 63);
    const long f = i >> 6;
    const int KS = K / 16;
    const int ct = (int)(f / KS), ks = (int)(f - (long)ct * KS);
    const uint4 v = *reinterpret_cast<const uint4*>(w + (size_t)(ct * 32 + (l & 31)) * K + ks * 16 + 8 * (l >> 5));
    *reinterpret_cast<uint4*>(out + (size_t)i * 8) = v;
}

void launch_fragpack(const void* w, int Cout, int K, void* out, hipStream_t st) {
    const long n = (long)Cout * K / 8;
    hipLaunchKernelGGL(fragpack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, static_cast<const bf16_t*>(w), Cout, K,
                       static_cast<bf16_t*>(out));
}

bool bneck_wide_fusable(const BneckWideArgs& a) {
    // first block of res2 (stride 1): fc = [conv3 | downsample] along K, bc = bias sum, no residual
    if (a.ds) return a.fa && a.fb && a.fc && a.zeros && a.Cmid == 64 && a.Cin == 64 && a.H == 64 && a.W == 64 && tune_get("HALO64S_DS", 1) != 0;
    if (!a.fa || !a.fb || !a.fc || a.Cin != 4 * a.Cmid || a.H != a.W) return false;
    // res5 (Cmid 512, 8x8 frames, two per workgroup) is instantiated and correct but NOT used: 8.7 MB of weights per
    // 128-pixel workgroup and only B/2 workgroups make it slower (240 us) than the three layer kernels (201 us)
    if (a.Cmid == 512 && a.H == 8) { const int f5 = tune_get("FUSE_WIDE5", 0); return f5 == 2 || (f5 == 1 && a.B % 2 == 0); }
    if (a.Cmid == 128 && a.H == 32) return a.zeros != nullptr && tune_get("FUSE_WIDE3", 1) != 0;
    if (a.Cmid == 64 && a.H == 64) return a.zeros != nullptr && tune_get("FUSE_WIDE2", 1) != 0;
    return a.Cmid == 256 && a.H == 16;
}

// res4 identity blocks of a small launch (<= HALF16_MAX frames, default 96) run on half-frame tiles, block by block (a tile's halo
// comes from the other half of the frame: no chaining inside one launch).  Frames/s with and without, by batch (two streams from
// 64 units, so a launch carries half the batch; tools/half16_sweep.sh): B = 32 + 14 %, 64 + 11 %, 100 + 9 %, 128 + 5 %, 160 + 3 %,
// 200 (100 frames per launch) + 0.3 %, 256 (128 per launch) - 2.7 % - the chained stage kernel wins once every CU has a frame.
static bool bneck_half16_wanted(const BneckWideArgs& a) {
    // ... and for launches just above half the chip (129 .. HALF16_HI = 160 frames, i.e. batches of 258 .. 320 in two halves): the two
    // concurrent frame-per-workgroup launches would need a second, mostly empty round of workgroups (B = 272 + 7 %, 300 + 5 %,
    // 340 + 2 %; from 384 the stage kernel wins again)
    if (a.ds || a.Cmid != 256 || a.H != 16 || a.W != 16 || !a.zeros) return false;
    return a.B <= tune_get("HALF16_MAX", 96) || (a.B > 128 && a.B <= tune_get("HALF16_HI", 160));
}

void launch_bneck_wide(const BneckWideArgs& a_in, hipStream_t st) {
    BneckWideArgs a = a_in;
    a.debug = tune_get("BDBG", 0) | (a_in.debug & 16);
    ConvArgs d{};
    d.B = a.B; d.H = a.H; d.W = a.W; d.Ho = a.H; d.Wo = a.W; d.Cin = a.Cin; d.Cout = 4 * a.Cmid; d.KH = 0; d.stride = 1; d.res = a.ds ? nullptr : a.x;
    void* tok = prof_begin(d, 2, st);
    if (a.Cmid == 128 && tune_get("HALO128S", 0)) hipLaunchKernelGGL(bneck_halo128s_kernel, dim3(a.B * 8), dim3(512), 0, st, a);
    else if (a.Cmid == 128 && tune_get("HALO_WLDS", 1)) hipLaunchKernelGGL((bneck_halo_kernel<128, true>), dim3(a.B * 4), dim3(512), 0, st, a);
    else if (a.Cmid == 128) hipLaunchKernelGGL((bneck_halo_kernel<128>), dim3(a.B * 4), dim3(512), 0, st, a);
    else if (a.ds && a.t1out && a.nd == 64) hipLaunchKernelGGL((bneck_halo64s_kernel<true, false, 64>), dim3(a.B * 32), dim3(512), 0, st, a);
    else if (a.ds) hipLaunchKernelGGL((bneck_halo64s_kernel<true, false, 0>), dim3(a.B * 32), dim3(512), 0, st, a);
    else if (a.Cmid == 64 && a.t1in && a.t1out && a.nd == 64) hipLaunchKernelGGL((bneck_halo64s_kernel<false, true, 64>), dim3(a.B * 32), dim3(512), 0, st, a);
    else if (a.Cmid == 64 && a.t1in && a.t1out && a.nd == 128 && a.y_s2) hipLaunchKernelGGL((bneck_halo64s_kernel<false, true, 128, true>), dim3(a.B * 32), dim3(512), 0, st, a);
    else if (a.Cmid == 64 && a.t1in && a.t1out && a.nd == 128) hipLaunchKernelGGL((bneck_halo64s_kernel<false, true, 128>), dim3(a.B * 32), dim3(512), 0, st, a);
    else if (a.Cmid == 64 && tune_get("HALO64S", 1)) hipLaunchKernelGGL((bneck_halo64s_kernel<false, false, 0>), dim3(a.B * 32), dim3(512), 0, st, a);
    else if (a.Cmid == 64) hipLaunchKernelGGL((bneck_halo_kernel<64>), dim3(a.B * 16), dim3(512), 0, st, a);
    else if (a.Cmid == 256 && bneck_half16_wanted(a)) hipLaunchKernelGGL(bneck_half16_kernel, dim3(a.B * 2), dim3(512), 0, st, a);
    else if (a.Cmid == 256) hipLaunchKernelGGL((bneck_wide_kernel<256, 16, 1>), dim3(a.B), dim3(512), 0, st, a);
    else if (tune_get("FUSE_WIDE5", 0) == 2) hipLaunchKernelGGL((bneck_wide_kernel<512, 8, 1>), dim3(a.B), dim3(512), 0, st, a);   // one frame per workgroup
    else hipLaunchKernelGGL((bneck_wide_kernel<512, 8, 2>), dim3(a.B / 2), dim3(512), 0, st, a);
    prof_end(tok, st);
}

bool bneck_stage_fusable(const BneckWideArgs& a) { return !a.ds && a.Cmid == 256 && a.H == 16 && a.W == 16 && tune_get("STAGE_RUN", 1) != 0 && !bneck_half16_wanted(a); }

void launch_bneck_wide_stage(const BneckStageArgs& s_in, hipStream_t st) {
    BneckStageArgs s = s_in;
    const BneckWideArgs& a = s.blk[0];
    ConvArgs d{};
    d.B = a.B * s.n;                   // report row: n blocks x B frames of the same shape
    d.H = a.H; d.W = a.W; d.Ho = a.H; d.Wo = a.W; d.Cin = a.Cin; d.Cout = 4 * a.Cmid; d.KH = 0; d.stride = 1; d.res = a.x;
    void* tok = prof_begin(d, 2, st);
    // Every second workgroup of each XCD starts ~half a block period late (110 k cycles), so that the memory phases (A: x in, C:
    // residual in / y out, 54 % of a block's time, bound by the XCD's path to memory) of one half meet the MFMA-only phase B of the
    // other: 693 -> 667 and 704 -> 682 us for the five chained blocks at B = 256 INCLUDING the 77 us of start skew (round 1 delayed
    // the odd workgroups = whole XCDs, which cannot help a per-XCD limit, and saw nothing).  Only when every CU has a workgroup.
    s.stagger = tune_get("STAGGER", a.B >= 200 ? 110000 : 0);
    hipLaunchKernelGGL((bneck_wide_stage_kernel<256, 16, 1>), dim3(a.B), dim3(512), 0, st, s);
    prof_end(tok, st);
}

}  // namespace ivosw

#ifdef IVOSW_PROBES
// Tuning probe: one wide fused bottleneck launch (weights given K-major packed; fragment-ordered copies are made into
// `frag`, >= 2 * (Cmid*Cin + 9*Cmid*Cmid + Cin*Cmid) + 256 bytes, the last 256 zero) with phase stamps ts [workgroups][8]
// (may be NULL).
extern "C" int ivosw_bneck_wide_probe(const void* x, void* y, const void* wa, const float* ba, const void* wb, const float* bb,
                                      const void* wc, const float* bc, void* frag, int B, int H, int W, int Cin, int Cmid,
                                      unsigned long long* ts, ivosw_stream_t stream) {
    using namespace ivosw;
    IVOSW_REQUIRE(x && y && wa && ba && wb && bb && wc && bc && frag, "null pointer");
    IVOSW_ON_DEVICE_OF(y);
    hipStream_t st = as_stream(stream);
    char* f = static_cast<char*>(frag);
    const size_t n1 = (size_t)Cmid * Cin * 2, n2 = (size_t)Cmid * 9 * Cmid * 2;
    launch_fragpack(wa, Cmid, Cin, f, st);
    launch_fragpack(wb, Cmid, 9 * Cmid, f + n1, st);
    launch_fragpack(wc, Cin, Cmid, f + n1 + n2, st);
    BneckWideArgs a{};
    a.x = x; a.y = y; a.fa = f; a.ba = ba; a.fb = f + n1; a.bb = bb; a.fc = f + n1 + n2; a.bc = bc;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cmid = Cmid; a.ts = ts;
    a.zeros = f + 2 * n1 + n2;                      // the caller provides 256 zeroed bytes behind the three weight copies
    IVOSW_REQUIRE(bneck_wide_fusable(a), "shape is not covered by the wide fused bottleneck kernel");
    launch_bneck_wide(a, st);
    IVOSW_CHECK_LAUNCH();
    return IVOSW_OK;
}
#endif  // IVOSW_PROBES

#ifdef IVOSW_PROBES
// Tuning probe: the big-register-tile contraction of round 5's attainable-roof measurement (gemm_bt.h) behind the C ABI, so that the
// kernel the micro-benchmark times is also built into the library and checked by the GPU tests.
extern "C" int ivosw_gemm_bt_probe(const void* A, const void* B, const float* bias, void* C, int M, int N, int K, int relu,
                                   unsigned long long* ts, ivosw_stream_t stream) {
    using namespace ivosw;
    IVOSW_REQUIRE(A && B && bias && C, "null pointer");
    IVOSW_REQUIRE(M > 0 && N > 0 && K >= 32 && M % 256 == 0 && N % 256 == 0 && K % 32 == 0, "M % 256, N % 256, K % 32 must be 0");
    IVOSW_REQUIRE((long)256 * K * 2 < (1L << 31), "K too large for 32-bit tile offsets");
    IVOSW_ON_DEVICE_OF(C);
    BtArgs a{};
    a.A = static_cast<const bf16_t*>(A); a.B = static_cast<const bf16_t*>(B); a.bias = bias; a.C = static_cast<bf16_t*>(C);
    a.M = M; a.N = N; a.K = K; a.relu = relu; a.ts = ts;
    hipLaunchKernelGGL(gemm_bt_kernel<0>, dim3((M / 256) * (N / 256)), dim3(256), 0, as_stream(stream), a);
    IVOSW_CHECK_LAUNCH();
    return IVOSW_OK;
}
#endif  // IVOSW_PROBES
