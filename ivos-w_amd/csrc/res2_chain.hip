// res2 (ResNet-50 layer1: three bottlenecks at 64 x 64, 64 -> 256 channels) + res3's forwarded conv1 as ONE launch, bf16, with the
// pointwise chain of every bottleneck kept in REGISTERS (round 6; sized in LAB_NOTES round 5 "next lever").
//
// Reference arithmetic: torchvision Bottleneck x 3 as Encoder.forward runs it (models/assessment.py:58, `self.res2`), BN folded:
//     b0: y0 = relu(W3 relu(W2 * relu(W1 p)) + Wd p)     b1, b2: y = relu(W3 relu(W2 * relu(W1 x)) + x)
// plus res3's first 1x1 (models/assessment.py:59) applied to y2 while it is on chip.
//
// Same tiling as res2_stage.hip (a workgroup owns an 8 x 16 OUTPUT tile through the three blocks on shrinking halos; one slot order
// [ 8x16 core | ring of 10x18 | ring of 12x20 ] for all blocks), a different dataflow:
//   * FOUR waves, one per SIMD, 512 registers each.  A wave owns pixel tiles (32 slots) with ALL channels: core tile w and ring tile
//     4 + w (block 0: both; block 1: ring tiles 4, 5 only; block 2: the core tile).
//   * Weights are the A operand of every 32x32x16 MFMA, so a lane holds 16 channels of ONE pixel; after bf16 packing such an accumulator
//     tile IS the B-operand fragment of the next contraction: t2 -> conv3 and y -> the next conv1 never leave the registers (no LDS image,
//     no permlane).  The next layer's weights carry the matching K permutation pi(8 h + e) = (e & 3) + 8 (e >> 2) + 4 h per 16 channels.
//   * The residual is TWO MFMAs per tile against identity fragments (y, packed, is a B operand already); a bias is ONE MFMA per channel
//     tile against a (hi, lo) bf16 split of the fp32 value (shared by a wave's two pixel tiles through the C operand).  A tile's epilogue is
//     8 v_cvt_pk_bf16_f32 + 8 v_pk_max_i16.  Block 0 has no identity: its conv3 is [conv3 | downsample] along K (K = 128, fp32 sum as in
//     the reference), the second half on p fragments every wave takes out of the halo raster at the end of A0 (32 registers through block 0's 3x3).  Summation orders differ from res2_stage.hip: this kernel is
//     NOT bit-identical to it; tests compare the two at the bf16 tolerance.
//   * Only the 3x3 inputs go through LDS (padded 144-byte raster rows as in res2_stage.hip).
//   * EVERY weight fragment (1 KB: lane l = row l & 31, k = 8 (l >> 5) ..) comes through ONE LDS ring per workgroup.  The fragments of a
//     tile - 520, packed in consumption order by res2_chain_pack_kernel - are fetched by LDS-DMA in groups of 16 (four pieces per wave), two
//     groups ahead of the group being opened (ring of 4 x 16 KB; groups of 8 in a ring of 6 or 8: + 2.8 % cycles - twice the barriers); one s_barrier per
//     group, counted s_waitcnt vmcnt: never drained inside a tile.  (res2_stage.hip streams 1.1 MB of
//     fragments per tile into registers through the texture path, every fragment once per wave that needs it; here 520 KB enter the CU once.)
// Micro-benchmark of the block body behind the go decision: tools/ubench/chain_bench.hip, profiles/r06_chain_bench.txt.
//
// LDS (149 312 B): ring [0, 65536) | P (14 x 22 x 128 B, A0 only) / T1_1 (12 x 20 x 144 B) [49152, 88576) | T1_0 (14 x 22 x 144 B) / T1_2
// (10 x 18 x 144 B) [88576, 132928).
// Build: this file is compiled with -mllvm -amdgpu-mfma-vgpr-form (build.py): accumulators in VGPRs, no v_accvgpr_read in the epilogues.
#include <algorithm>
#include <type_traits>

#include "conv.h"
#include "mfma_tile.h"

namespace ivosw {

namespace {
#ifndef RC_GROUPS_N
#define RC_GROUPS_N 4
#endif
#ifndef RC_GF
#define RC_GF 16                                    // fragments per ring group (one s_barrier per group): 8 | 16
#endif
constexpr int RC_GROUPS = RC_GROUPS_N, RC_LEAD = RC_GROUPS - 2, RC_GB = RC_GF * 1024, RC_RING = RC_GROUPS * RC_GB;   // a group is requested RC_LEAD groups ahead
constexpr int RC_NP = RC_GF / 4;                    // pieces (1 KB) per wave and group
constexpr int RC_T1R = 144;
constexpr int RC_P_OFF = RC_RING, RC_T1B_OFF = RC_RING;                   // P | T1_1
constexpr int RC_T1A_OFF = RC_RING + 308 * 128;                           // T1_0 | T1_2
constexpr int RC_LDS = RC_T1A_OFF + 308 * RC_T1R;
static_assert(240 * RC_T1R <= 308 * 128 && RC_LDS <= 163840, "LDS map");
constexpr int RC_NF = 520, RC_NG = (RC_NF + RC_GF - 1) / RC_GF;      // (a last partial group is fetched whole: the stream buffer is RC_NG groups long)
constexpr int S_A0 = 0, S_B0 = 10, S_B1 = 190, S_B2 = 338;               // segment bases (fragment indices)
// offsets inside a block segment; conv3 takes 5 fragments per channel tile {bias, W3 x 4}, block 0 nine {bias sum, W3 x 4, Wd x 4}
constexpr int O_B2 = 0, O_W2 = 2, O_C3 = 74;
#ifndef RC_DEPTH
#define RC_DEPTH 2
#endif
#ifndef RC_DEFER
#define RC_DEFER 1
#endif

template <int I, int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}
template <int N>
__device__ __forceinline__ void lgkm() {
    static_assert(N >= 0, "counted wait");
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N < 15 ? N : 15) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void pin() { __builtin_amdgcn_sched_barrier(0); }
template <int V> using IC = std::integral_constant<int, V>;

// slot -> pixel offset from the tile origin; unused slots map to the origin (their results are dropped)   [as res2_stage.hip]
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

// the weight ring: groups of 8 fragments, RC_GROUPS slots; boundary<J>() runs in front of the first read of the tile's group J.
// PERSIST: a workgroup walks over several tiles and the ring does not stop at a tile boundary - the groups behind a tile's last one are
// the NEXT tile's first ones - and 10 boundaries of block 2 (from J = RC_AUX0) also carry one "aux" piece each: the next tile's p halo
// (P aliases t1_1, which is dead from block 2 on).  Counted waits: a boundary needs ITS group's two
// pieces; younger than those are the pieces of the three following groups (6) plus the aux pieces of the three boundaries before it.
// A tile ends with vmcnt(0) + its global stores (stores count in vmcnt and may complete early: no counted wait may rely on them), so the
// first RC_LEAD boundaries of a following tile wait for nothing - their groups were in flight before the drain.
constexpr int RC_AUX0 = RC_GF == 8 ? 43 : 22, RC_NAUX = 10;       // the first boundary of block 2 (fragment 338) and the ten behind it
template <bool PERSIST>
struct Ring {
    __amdgpu_buffer_rsrc_t rs;
    unsigned char* lds;
    unsigned lds_base, vcur;
    int wave, lane, vpiece, slot, sfill;
    int first, has_next;                       // first tile of this workgroup; a next tile exists
    const bf16_t* nX;                          // next tile: frame base and origin
    int ny0, nx0;
    const bf16_t* zeros;
    template <int G>
    __device__ __forceinline__ int group_soff() {                  // stream offset of (tile-local) group G
        if constexpr (G < RC_NG) return G * RC_GB;
        else if constexpr (PERSIST) return has_next ? (G - RC_NG) * RC_GB : RC_NG * RC_GB;
        else return RC_NG * RC_GB;                                   // behind the last group: out of range (zeros, no memory access)
    }
    template <int G, int Q>
    __device__ __forceinline__ void issue_piece(int s) {            // piece Q (0 / 1) of this wave's two pieces of group G into ring slot s
        unsigned char* dst = lds + s * RC_GB + wave * (RC_NP * 1024) + Q * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)dst, 16, vpiece + Q * 1024, group_soff<G>(), 0, 0);
    }
    template <int G>
    __device__ __forceinline__ void issue_group(int s) { sfor<0, RC_NP>([&](auto qc) { issue_piece<G, decltype(qc)::value>(s); }); }
    // p halo group g (8 raster rows of 14 x 22) / slot-ordered group g (8 compact rows) of the tile at (X, y0, x0)
    __device__ __forceinline__ void p_piece(const bf16_t* X, int y0, int x0, int g, bool live) {
        const int rsub = lane >> 3, cpos = lane & 7;
        const int row = g * 8 + rsub;
        const int hy = row / 22, hx = row - hy * 22;
        const int y = y0 - 3 + hy, x = x0 - 3 + hx;
        const bool ok = live && y >= 0 && y < 64 && x >= 0 && x < 64;
        const bf16_t* src = ok ? X + ((size_t)y * 64 + x) * 64 + (cpos ^ ((row >> 1) & 7)) * 8 : zeros;
        if (row < 308) dma16(src, lds + RC_P_OFF + g * 1024);        // (the last group is half a piece: EXEC masks the rows beyond the raster)
    }
    template <int K>
    __device__ __forceinline__ void aux() {            // aux piece K of this wave (a piece that does not exist repeats the one before)
        int g = wave + 4 * K;
        if (g >= 39) g -= 4;
        p_piece(nX, ny0, nx0, g, has_next != 0);
    }
    // The pieces a boundary owes (two of group J + RC_LEAD, one aux piece in block 2) are NOT issued at the barrier: an LDS-DMA issue
    // holds the wave for 50+ cycles, and behind a barrier the matrix pipe of this SIMD has nothing queued (one wave per SIMD) - they go
    // out behind the reads of the group's fragments 2, 4 and 6, each of which sits right behind an MFMA (RC_DEFER; 0: all at the barrier).
    // Counted waits: when boundary J waits, the pieces of boundaries J - W .. J - 1 are out (W = RC_LEAD - 1 groups in flight).
    template <int J>
    __device__ __forceinline__ void boundary() {
        constexpr int W = RC_LEAD - 1;                                       // groups in flight behind the one being opened
        constexpr int lo = J - W > RC_AUX0 ? J - W : RC_AUX0, hi = J < RC_AUX0 + RC_NAUX ? J : RC_AUX0 + RC_NAUX;
        constexpr int nauxb = PERSIST && hi > lo ? hi - lo : 0;           // aux boundaries among J - W .. J - 1
        if constexpr (PERSIST && J < RC_LEAD) { if (first) wait_vmcnt<RC_NP * W>(); }
        else wait_vmcnt<RC_NP * W + nauxb>();
        pin();
        __builtin_amdgcn_s_barrier();          // everybody's pieces of group J have landed; everybody has consumed group J - 2: its slot is free
        pin();
        vcur = lds_base + slot * RC_GB + lane * 16;
        sfill = slot + RC_LEAD;                // the slot group J + RC_LEAD goes to (= the slot of group J - 2)
        if (sfill >= RC_GROUPS) sfill -= RC_GROUPS;
        slot = slot + 1 == RC_GROUPS ? 0 : slot + 1;
        if constexpr (!RC_DEFER) sfor<0, RC_NP + 1>([&](auto qc) { owed<J, decltype(qc)::value>(); });
        pin();
    }
    template <int J, int Q>
    __device__ __forceinline__ void owed() {
        if constexpr (Q < RC_NP) issue_piece<J + RC_LEAD, Q>(sfill);
        else if constexpr (PERSIST && J >= RC_AUX0 && J < RC_AUX0 + RC_NAUX) aux<J - RC_AUX0>();
    }
    // the tile's last group may be partial (520 fragments, groups of 16): what it still owes goes out behind the last fragment read
    __device__ __forceinline__ void finish_tile() {
        if constexpr (RC_DEFER) {
            constexpr int RL = (RC_NF - 1) % RC_GF;
            sfor<0, RC_NP + 1>([&](auto qc) { constexpr int Q = decltype(qc)::value; if constexpr (2 * Q + 2 > RL) owed<RC_NG - 1, Q>(); });
        }
    }
    template <int F>
    __device__ __forceinline__ u32x4 rd() {
        if constexpr (F % RC_GF == 0) boundary<F / RC_GF>();
        const u32x4 v = lds_read_b128_o<(F % RC_GF) * 1024>(vcur);
        constexpr int R = F % RC_GF;                     // the group's pieces behind its fragment reads 2, 4, ...; the aux piece behind the next even one
        if constexpr (RC_DEFER && R >= 2 && R % 2 == 0 && R / 2 - 1 <= RC_NP) owed<F / RC_GF, R / 2 - 1>();
        return v;
    }
};

// ---------------------------------------------------------------- one bottleneck behind its conv1
// in : t1 raster (width SRCW, origin OFS pixels up-left of the 3x3's own region), y = residual (packed B fragments) of NPT pixel tiles
// out: y (in place), acc = the next conv1's pre-activations: [pixel tile][channel tile] (NM = 2 or 4 channel tiles)
// DS (block 0): no residual; conv3 = [conv3 | downsample], the second K half on the p fragments ps[pixel tile][k-step]
// side(S): called once per k-step S of the next conv1's loop (block 2 stores y2 there); side3(S): once per k-step of the 3x3 (block 0 stores the
// previous tile's t1out there)
template <int NPT, int SRCW, int SEG, int NM, bool DS, typename RingT, typename SideT, typename Side3T>
__device__ __forceinline__ void chain_block(RingT& ring, unsigned (&y)[2][64], const unsigned (&rb)[2], const u32x4 (&idf)[2], const u32x4& ones,
                                            f32x16 (&acc)[NPT * NM], const u32x4 (&ps)[2][4], SideT&& side, Side3T&& side3) {
    constexpr int D = RC_DEPTH;
    constexpr int CS = DS ? 9 : 5, O_B1 = O_C3 + 8 * CS;
    const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned t2[NPT][16];
    // ======================================================== 3x3: t2 = relu(b2 + W2 * t1)
    {
        f32x16 a2[NPT][2];
        constexpr int R = 2 + NPT;                            // reads of a k-step, in the order A0, B0, A1, B1
        u32x4 a[D + 1][2], b[D + 1][NPT];
        auto rdstep = [&](auto sc, auto qc) {
            constexpr int S = decltype(sc)::value, Q = decltype(qc)::value, buf = S % (D + 1);
            constexpr int tap = S >> 2, kk = S & 3, toff = ((tap / 3) * SRCW + (tap % 3)) * RC_T1R + kk * 32;
            if constexpr (Q == 0) a[buf][0] = ring.template rd<SEG + O_W2 + 2 * S>();
            else if constexpr (Q == 2) a[buf][1] = ring.template rd<SEG + O_W2 + 2 * S + 1>();
            else if constexpr (Q == 1) b[buf][0] = lds_read_b128_o<toff>(rb[0]);
            else b[buf][NPT - 1] = lds_read_b128_o<toff>(rb[NPT - 1]);
        };
        const u32x4 bf0 = ring.template rd<SEG + O_B2>(), bf1 = ring.template rd<SEG + O_B2 + 1>();
        sfor<0, D>([&](auto sc) { sfor<0, R>([&](auto qc) { rdstep(sc, qc); }); });
        lgkm<R * D>();
#pragma unroll
        for (int i = 0; i < NPT; ++i) { a2[i][0] = mfma_bf16(bf0, ones, z); a2[i][1] = mfma_bf16(bf1, ones, z); }
        sfor<0, 36>([&](auto sc) {
            constexpr int S = decltype(sc)::value, buf = S % (D + 1);
            constexpr bool more = S + D < 36;
            constexpr int full = R * (S + D < 36 ? S + D : 36);       // reads of all k-steps below min(S + D, 36) are issued
            using SN = IC<S + D>;
            // allowed outstanding = issued - (index of the newest read this MFMA needs) - 1
            lgkm<full - (R * S + 1) - 1>();
            a2[0][0] = mfma_bf16(a[buf][0], b[buf][0], a2[0][0]);
            if constexpr (more) rdstep(SN{}, IC<0>{});
            side3(sc);
            lgkm<full + (more ? 1 : 0) - (R * S + 2) - 1>();
            a2[0][1] = mfma_bf16(a[buf][1], b[buf][0], a2[0][1]);
            if constexpr (more) { rdstep(SN{}, IC<1>{}); if constexpr (NPT == 1) rdstep(SN{}, IC<2>{}); }
            if constexpr (NPT == 2) {
                lgkm<full + (more ? 2 : 0) - (R * S + 3) - 1>();
                a2[1][0] = mfma_bf16(a[buf][0], b[buf][1], a2[1][0]);
                if constexpr (more) rdstep(SN{}, IC<2>{});
                pin();
                a2[1][1] = mfma_bf16(a[buf][1], b[buf][1], a2[1][1]);
                if constexpr (more) rdstep(SN{}, IC<3>{});
            }
            pin();
        });
        // t2's epilogue rides under channel tile 0's bias / residual MFMAs of the conv3 loop below: hand the accumulators over
        // ==================================================== conv3 + residual, channel tile by channel tile
        u32x4 c[2][CS];
        f32x16 a3[2][NPT];          // [m & 1][pixel tile]
        auto rdc = [&](auto mc, auto jc) {
            constexpr int M = decltype(mc)::value, J = decltype(jc)::value;
            c[M & 1][J] = ring.template rd<SEG + O_C3 + CS * M + J>();
        };
        sfor<0, CS>([&](auto jc) { rdc(IC<0>{}, jc); });
        auto epi_t2 = [&](auto ic, auto mc, auto hc) {          // half (8 values -> 4 registers) of t2 tile (pixel tile I, channel tile M)
            constexpr int I = decltype(ic)::value, M = decltype(mc)::value, H = decltype(hc)::value;
            if constexpr (I < NPT) {
#pragma unroll
                for (int q = 0; q < 4; ++q) t2[I][8 * M + 4 * H + q] = relu2_bf16(a2[I][M][8 * H + 2 * q], a2[I][M][8 * H + 2 * q + 1]);
            }
        };
        auto epi_y = [&](auto mc, auto ic, auto hc) {           // half of y tile (M, pixel tile I) from a3[M & 1][I]
            constexpr int M = decltype(mc)::value, I = decltype(ic)::value, H = decltype(hc)::value;
            if constexpr (I < NPT) {
#pragma unroll
                for (int q = 0; q < 4; ++q) y[I][8 * M + 4 * H + q] = relu2_bf16(a3[M & 1][I][8 * H + 2 * q], a3[M & 1][I][8 * H + 2 * q + 1]);
            }
        };
        auto epi_yq = [&](auto mc, auto ic, auto qc) {          // a quarter (4 values -> 2 registers) of y tile (M, pixel tile I)
            constexpr int M = decltype(mc)::value, I = decltype(ic)::value, Q = decltype(qc)::value;
            if constexpr (I < NPT) {
                y[I][8 * M + 2 * Q] = relu2_bf16(a3[M & 1][I][4 * Q], a3[M & 1][I][4 * Q + 1]);
                y[I][8 * M + 2 * Q + 1] = relu2_bf16(a3[M & 1][I][4 * Q + 2], a3[M & 1][I][4 * Q + 3]);
            }
        };
        sfor<0, 8>([&](auto mc) {
            constexpr int M = decltype(mc)::value, cb = M & 1;
            using MP = IC<M - 1>; using MN = IC<M + 1>;
            // filler K of this channel tile: t2's epilogue under tile 0, tile M - 1's y epilogue otherwise
            // (tile M - 1's epilogue in QUARTERS - 2 v_cvt_pk + 2 v_pk_max - behind MFMAs 2 .. 9 of this tile: a wave alone on its SIMD hides
            // about five issues per MFMA, sixteen in one gap do not hide)
            auto fill = [&](auto kc) {
                constexpr int K = decltype(kc)::value;
                if constexpr (M == 0) { if constexpr (K < 8) epi_t2(IC<((K >> 1) & 1)>{}, IC<(K >> 2)>{}, IC<(K & 1)>{}); }
                else if constexpr (K >= 2 && K < 2 + 4 * NPT) epi_yq(MP{}, IC<((K - 2) >> 2)>{}, IC<((K - 2) & 3)>{});
            };
            lgkm<CS - 1>();                              // this tile's bias fragment has landed (its weight fragments may not)
            a3[cb][0] = mfma_bf16(c[cb][0], ones, z); fill(IC<0>{}); pin();
            if constexpr (NPT == 2) { a3[cb][1] = mfma_bf16(c[cb][0], ones, z); }
            fill(IC<1>{});
            if constexpr (M < 7) rdc(MN{}, IC<0>{});
            pin();
            if constexpr (DS) {
                lgkm<(M < 7 ? 1 : 0)>();                 // all of this tile's fragments (the first read of the next tile may be out)
                sfor<0, 4>([&](auto kc) {                // the downsample half first: it does not wait for t2's epilogue
                    constexpr int KS = decltype(kc)::value;
                    a3[cb][0] = mfma_bf16(c[cb][5 + KS], ps[0][KS], a3[cb][0]); fill(IC<2 + 2 * KS>{}); pin();
                    if constexpr (NPT == 2) { a3[cb][1] = mfma_bf16(c[cb][5 + KS], ps[1][KS], a3[cb][1]); }
                    fill(IC<3 + 2 * KS>{});
                    if constexpr (M < 7) rdc(MN{}, IC<1 + KS>{});
                    pin();
                });
                sfor<0, 4>([&](auto kc) {
                    constexpr int KS = decltype(kc)::value;
                    a3[cb][0] = mfma_bf16(c[cb][1 + KS], *reinterpret_cast<u32x4*>(&t2[0][4 * KS]), a3[cb][0]); pin();
                    if constexpr (NPT == 2) { a3[cb][1] = mfma_bf16(c[cb][1 + KS], *reinterpret_cast<u32x4*>(&t2[1][4 * KS]), a3[cb][1]); }
                    if constexpr (M < 7) rdc(MN{}, IC<5 + KS>{});
                    pin();
                });
            } else {
                a3[cb][0] = mfma_bf16(idf[0], *reinterpret_cast<u32x4*>(&y[0][8 * M]), a3[cb][0]); fill(IC<2>{}); pin();
                if constexpr (NPT == 2) { a3[cb][1] = mfma_bf16(idf[0], *reinterpret_cast<u32x4*>(&y[1][8 * M]), a3[cb][1]); }
                fill(IC<3>{});
                if constexpr (M < 7) rdc(MN{}, IC<1>{});
                pin();
                a3[cb][0] = mfma_bf16(idf[1], *reinterpret_cast<u32x4*>(&y[0][8 * M + 4]), a3[cb][0]); fill(IC<4>{}); pin();
                if constexpr (NPT == 2) { a3[cb][1] = mfma_bf16(idf[1], *reinterpret_cast<u32x4*>(&y[1][8 * M + 4]), a3[cb][1]); }
                fill(IC<5>{});
                if constexpr (M < 7) rdc(MN{}, IC<2>{});
                pin();
                if constexpr (M == 0) { fill(IC<6>{}); fill(IC<7>{}); }
                lgkm<(M < 7 ? 3 : 0)>();                 // all of this tile's fragments (the three reads of the next tile may be out)
                sfor<0, 4>([&](auto kc) {
                    constexpr int KS = decltype(kc)::value;
                    a3[cb][0] = mfma_bf16(c[cb][1 + KS], *reinterpret_cast<u32x4*>(&t2[0][4 * KS]), a3[cb][0]);
                    if constexpr (M > 0) fill(IC<6 + 2 * KS>{});
                    pin();
                    if constexpr (NPT == 2) { a3[cb][1] = mfma_bf16(c[cb][1 + KS], *reinterpret_cast<u32x4*>(&t2[1][4 * KS]), a3[cb][1]); }
                    if constexpr (M > 0) fill(IC<7 + 2 * KS>{});
                    if constexpr (M < 7 && KS < 2) rdc(MN{}, IC<3 + KS>{});
                    pin();
                });
            }
        });
        // ==================================================== the next conv1: acc[i * NM + m] = b1 + W1' y   (channel tile 7's epilogue under its bias MFMAs)
        {
            u32x4 w[D + 1][NM];
            auto rdw = [&](auto sc, auto qc) {
                constexpr int S = decltype(sc)::value, Q = decltype(qc)::value;
                w[S % (D + 1)][Q] = ring.template rd<SEG + O_B1 + NM + NM * S + Q>();
            };
            u32x4 bfr[NM];
            sfor<0, NM>([&](auto mc) { bfr[decltype(mc)::value] = ring.template rd<SEG + O_B1 + decltype(mc)::value>(); });
            sfor<0, D>([&](auto sc) { sfor<0, NM>([&](auto qc) { rdw(sc, qc); }); });
            lgkm<NM * D>();
            sfor<0, NM>([&](auto mc) {
                constexpr int M = decltype(mc)::value;
                acc[M] = mfma_bf16(bfr[M], ones, z);
                if constexpr (NPT == 2) acc[NM + M] = mfma_bf16(bfr[M], ones, z);
                if constexpr (M < 2) { epi_y(IC<7>{}, IC<M>{}, IC<0>{}); epi_y(IC<7>{}, IC<M>{}, IC<1>{}); }
                pin();
            });
            sfor<0, 16>([&](auto sc) {
                constexpr int S = decltype(sc)::value, buf = S % (D + 1);
                constexpr bool more = S + D < 16;
                constexpr int full = NM * (S + D < 16 ? S + D : 16);
                sfor<0, NM>([&](auto mc) {
                    constexpr int M = decltype(mc)::value;
                    lgkm<full + (more ? M : 0) - (NM * S + M) - 1>();
                    acc[M] = mfma_bf16(w[buf][M], *reinterpret_cast<u32x4*>(&y[0][4 * S]), acc[M]);
                    if constexpr (more) rdw(IC<S + D>{}, mc);
                    if constexpr (M == 1) side(sc);
                    pin();
                    if constexpr (NPT == 2) { acc[NM + M] = mfma_bf16(w[buf][M], *reinterpret_cast<u32x4*>(&y[1][4 * S]), acc[NM + M]); pin(); }
                });
            });
        }
    }
}

// ReLU, bf16, frame mask, 8-byte pieces of 4 consecutive channels of the lane's pixel into a t1 raster (addr: the pixel's row + 8 * lhalf)
__device__ __forceinline__ void store_t1(const f32x16& d, unsigned addr, unsigned mask, int m) {
    if (addr == 0xffffffffu) return;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        u32x2 pk;
        pk.x = relu2_bf16(d[4 * g], d[4 * g + 1]) & mask;
        pk.y = relu2_bf16(d[4 * g + 2], d[4 * g + 3]) & mask;
        lds_write_b64(addr + (m * 4 + g) * 16, pk);
    }
}
// 8 packed registers of one channel tile (lane = pixel, channels 4 h + {0..3, 8..11, 16..19, 24..27}) -> two 16-byte pieces per lane:
// lower lanes channels [0, 8) and [16, 24), upper lanes [8, 16) and [24, 32) of the tile
__device__ __forceinline__ void tile_to_rows(const unsigned (&q)[8], u32x4& lo, u32x4& hi) {
    const auto s0 = __builtin_amdgcn_permlane32_swap(q[0], q[2], false, false);
    const auto s1 = __builtin_amdgcn_permlane32_swap(q[1], q[3], false, false);
    const auto s2 = __builtin_amdgcn_permlane32_swap(q[4], q[6], false, false);
    const auto s3 = __builtin_amdgcn_permlane32_swap(q[5], q[7], false, false);
    lo = u32x4{s0[0], s1[0], s0[1], s1[1]};
    hi = u32x4{s2[0], s3[0], s2[1], s3[1]};
}
}  // namespace

template <bool YS2, bool PERSIST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void res2_chain_kernel(Res2ChainArgs kargs) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[RC_LDS];
    const Res2ChainArgs& p = kargs;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhalf = lane >> 5;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const bf16_t* zeros = static_cast<const bf16_t*>(p.zeros);
    const int ntiles = p.B * 32, G = gridDim.x;
    const int L = xcd_remap(blockIdx.x, G);          // consecutive logical ids on one XCD: concurrent neighbours share their halos through that L2
    if (L >= ntiles) return;
    const int nk = PERSIST ? (ntiles - L + G - 1) / G : 1;
    auto tile_of = [&](int k) { const int idx = L + k * G; return p.rev ? ntiles - 1 - idx : idx; };
    int b, y0, x0;
    const bf16_t* X;
    auto set_tile = [&](int t) {
        b = t >> 5; y0 = ((t & 31) >> 2) * 8; x0 = (t & 3) * 16;
        X = static_cast<const bf16_t*>(p.x) + (size_t)b * 64 * 64 * 64;
    };
    set_tile(tile_of(0));
    // phase stamps (probe only; PERSIST: of the workgroup's second tile).  s_memtime is a scalar-memory instruction: it shares lgkmcnt with
    // the LDS reads and returns out of order, so it is waited for at once - never outstanding beside a counted fragment wait
    unsigned long long stamps[7];
    const int kstamp = nk > 1 ? 1 : 0;
    int kcur = 0;
    auto stamp = [&](int i) {
        if (p.ts && kcur == kstamp) { stamps[i] = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    };

    Ring<PERSIST> ring;
    ring.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wstream), 0, RC_NG * RC_GB, 0x00020000);
    ring.lds = lds; ring.lds_base = lds_base; ring.wave = wave; ring.lane = lane; ring.vpiece = wave * (RC_NP * 1024) + lane * 16; ring.slot = 0; ring.sfill = 0; ring.vcur = 0;
    ring.first = 1; ring.has_next = 0; ring.nX = X; ring.ny0 = 0; ring.nx0 = 0; ring.zeros = zeros;
    // ---------------- the first tile's p halo (14 x 22 raster, 39 groups of 8 rows of 128 B, chunks XOR (row >> 1) & 7) by LDS-DMA; then the
    // ring's first groups
    if (kstamp == 0) stamp(0);
    for (int g = wave; g < 39; g += 4) ring.p_piece(X, y0, x0, g, true);
    sfor<0, RC_LEAD>([&](auto gc) { ring.template issue_group<decltype(gc)::value>(decltype(gc)::value); });

    // constant fragments: identity halves (A operand), ones (B operand of the bias MFMAs)
    u32x4 idf[2], ones;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e0 = 2 * q, e1 = 2 * q + 1;
            const unsigned v0 = (lrow == 16 * s + (e0 & 3) + 8 * (e0 >> 2) + 4 * lhalf) ? 0x3f80u : 0u;
            const unsigned v1 = (lrow == 16 * s + (e1 & 3) + 8 * (e1 >> 2) + 4 * lhalf) ? 0x3f80u : 0u;
            idf[s][q] = v0 | (v1 << 16);
        }
    ones = u32x4{lhalf ? 0u : 0x3f803f80u, 0u, 0u, 0u};
    const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    // the wave's two pixel tiles: core tile `wave` and ring tile 4 + wave; pixel offsets from the tile origin
    int pdy[2], pdx[2];
    pdy[0] = 2 * wave + (lrow >> 4); pdx[0] = lrow & 15;
    slot_pos((4 + wave) * 32 + lrow, pdy[1], pdx[1]);
    const bool used1 = slot_used((4 + wave) * 32 + lrow);
    auto inframe = [&](int dy, int dx) { const int yy = y0 + dy, xx = x0 + dx; return (yy >= 0 && yy < 64 && xx >= 0 && xx < 64) ? 0xffffffffu : 0u; };

    unsigned y[2][64];          // [pixel tile][k-step t: 4 t .. 4 t + 3]: channels 16 t + pi(8 h + e)
    u32x4 ps[2][4];             // p at the wave's slots: block 0's downsample operand, taken out of the halo raster at the end of A0
    auto noside = [](auto) {};
    // a tile's t1out (8 x 16 bytes per lane) leaves under the NEXT tile's first 3x3 loop, one store per k-step 4 .. 11: issued at the end of
    // the tile they were a tail of ~ 1.4 k cycles nothing overlapped (a CU's write path takes ~ 10 B/clk); the last tile's are flushed behind the loop
    u32x4 pend[8];
    bf16_t* pend_ptr = nullptr;
    auto t1side = [&](auto sc) {
        constexpr int S = decltype(sc)::value;
        if constexpr (S >= 4 && S < 12) {
            if (pend_ptr) *reinterpret_cast<u32x4*>(pend_ptr + 32 * ((S - 4) >> 1) + 16 * ((S - 4) & 1)) = pend[S - 4];
        }
    };
  for (int k = 0; k < nk; ++k) {
    kcur = k;
    if (PERSIST) {
        ring.first = k == 0;
        ring.has_next = k + 1 < nk;
        if (k + 1 < nk) {
            const int t = tile_of(k + 1);
            ring.nX = static_cast<const bf16_t*>(p.x) + (size_t)(t >> 5) * 64 * 64 * 64;
            ring.ny0 = ((t & 31) >> 2) * 8; ring.nx0 = (t & 3) * 16;
        }
        if (k == kstamp && k > 0) stamp(0);
    }
    // ================================================================ A0: t1_0 = relu(b1 + W1 p) on the 14 x 22 raster
    {
        const bool three = wave < 2;                     // raster pixel tiles wave, wave + 4, wave + 8 (< 10)
        unsigned prow[3], pkey[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int hr = (wave + 4 * i) * 32 + lrow;
            prow[i] = lds_base + RC_P_OFF + hr * 128;
            pkey[i] = (hr >> 1) & 7;
        }
        const u32x4 bf0 = ring.template rd<S_A0>(), bf1 = ring.template rd<S_A0 + 1>();        // (the first boundary: the halo is older than the ring's pieces)
        if (p.ts) { lgkm<0>(); stamp(1); }
        u32x4 wa[4][2];
        sfor<0, 4>([&](auto kc) { constexpr int KS = decltype(kc)::value; wa[KS][0] = ring.template rd<S_A0 + 2 + 2 * KS>(); wa[KS][1] = ring.template rd<S_A0 + 3 + 2 * KS>(); });
        f32x16 a1[3][2];
        lgkm<8>();
#pragma unroll
        for (int i = 0; i < 3; ++i) { a1[i][0] = mfma_bf16(bf0, ones, z); a1[i][1] = mfma_bf16(bf1, ones, z); }
        u32x4 pf[2][3];
        auto rdp = [&](int ks, int buf) {
#pragma unroll
            for (int i = 0; i < 3; ++i) pf[buf][i] = lds_read_b128(prow[i] + (((2 * ks + lhalf) ^ pkey[i]) << 4));
        };
        rdp(0, 0);
        sfor<0, 4>([&](auto kc) {
            constexpr int KS = decltype(kc)::value;
            if constexpr (KS < 3) { rdp(KS + 1, (KS + 1) & 1); lgkm<3>(); } else lgkm<0>();
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (i == 2 && !three) break;
                a1[i][0] = mfma_bf16(wa[KS][0], pf[KS & 1][i], a1[i][0]);
                a1[i][1] = mfma_bf16(wa[KS][1], pf[KS & 1][i], a1[i][1]);
            }
        });
        // p at the wave's slots, for block 0 (the raster dies at the barrier below: t1_1 takes its place)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int hr = (pdy[i] + 3) * 22 + pdx[i] + 3;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) ps[i][ks] = lds_read_b128(lds_base + RC_P_OFF + hr * 128 + (((2 * ks + lhalf) ^ ((hr >> 1) & 7)) << 4));
        }
        // conv1's epilogue: raster stores
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (i == 2 && !three) break;
            const int hr = (wave + 4 * i) * 32 + lrow;
            const int hy = hr / 22, hx = hr - hy * 22;
            const unsigned mask = inframe(hy - 3, hx - 3);
            const unsigned addr = hr < 308 ? lds_base + RC_T1A_OFF + hr * RC_T1R + 8 * lhalf : 0xffffffffu;
            store_t1(a1[i][0], addr, mask, 0);
            store_t1(a1[i][1], addr, mask, 1);
        }
        lds_wait();
        __builtin_amdgcn_s_barrier();                // t1_0 complete, P dead
        pin();
        stamp(2);
    }
    // ================================================================ block 0 (12 x 20 region: all eight tiles)
    {
        unsigned rb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) rb[i] = lds_base + RC_T1A_OFF + ((pdy[i] + 2) * 22 + pdx[i] + 2) * RC_T1R + lhalf * 16;
        f32x16 acc[4];
        chain_block<2, 22, S_B0, 2, true>(ring, y, rb, idf, ones, acc, ps, noside, t1side);
        const unsigned a0 = lds_base + RC_T1B_OFF + ((pdy[0] + 2) * 20 + pdx[0] + 2) * RC_T1R + 8 * lhalf;
        const unsigned a1 = used1 ? lds_base + RC_T1B_OFF + ((pdy[1] + 2) * 20 + pdx[1] + 2) * RC_T1R + 8 * lhalf : 0xffffffffu;
        const unsigned m1 = inframe(pdy[1], pdx[1]);
        store_t1(acc[0], a0, 0xffffffffu, 0); store_t1(acc[1], a0, 0xffffffffu, 1);
        store_t1(acc[2], a1, m1, 0); store_t1(acc[3], a1, m1, 1);
        lds_wait();
        __builtin_amdgcn_s_barrier();                // t1_1 complete, t1_0 dead
        pin();
        stamp(3);
    }
    // ================================================================ block 1 (10 x 18 region: tiles 0 .. 5; waves 2, 3 own one)
    {
        unsigned rb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) rb[i] = lds_base + RC_T1B_OFF + ((pdy[i] + 1) * 20 + pdx[i] + 1) * RC_T1R + lhalf * 16;
        const unsigned a0 = lds_base + RC_T1A_OFF + ((pdy[0] + 1) * 18 + pdx[0] + 1) * RC_T1R + 8 * lhalf;
        if (wave < 2) {
            f32x16 acc[4];
            chain_block<2, 20, S_B1, 2, false>(ring, y, rb, idf, ones, acc, ps, noside, noside);
            const unsigned a1 = used1 ? lds_base + RC_T1A_OFF + ((pdy[1] + 1) * 18 + pdx[1] + 1) * RC_T1R + 8 * lhalf : 0xffffffffu;
            const unsigned m1 = inframe(pdy[1], pdx[1]);
            store_t1(acc[0], a0, 0xffffffffu, 0); store_t1(acc[1], a0, 0xffffffffu, 1);
            store_t1(acc[2], a1, m1, 0); store_t1(acc[3], a1, m1, 1);
        } else {
            f32x16 acc[2];
            chain_block<1, 20, S_B1, 2, false>(ring, y, rb, idf, ones, acc, ps, noside, noside);
            store_t1(acc[0], a0, 0xffffffffu, 0); store_t1(acc[1], a0, 0xffffffffu, 1);
        }
        lds_wait();
        __builtin_amdgcn_s_barrier();                // t1_2 complete, t1_1 dead: the next tile's p halo may land in its place
        pin();
        stamp(4);
    }
    // ================================================================ block 2 (the 8 x 16 tile) + res3's conv1 (256 -> 128)
    {
        unsigned rb[2];
        rb[0] = rb[1] = lds_base + RC_T1A_OFF + (pdy[0] * 18 + pdx[0]) * RC_T1R + lhalf * 16;
        f32x16 acc[4];
        // y2 (256 channels of the lane's pixel; YS2: the even pixels only, compactly) leaves under the MFMAs of res3's conv1, one 16-byte
        // piece per lane and k-step: a CU's write path takes ~ 10 B/clk, the tile's 48 KB of output would otherwise be a tail of ~ 4 k cycles
        const bool mine = !YS2 || (((pdy[0] | pdx[0]) & 1) == 0);
        bf16_t* YO = YS2 ? static_cast<bf16_t*>(p.y) + (((size_t)b * 32 + ((y0 + pdy[0]) >> 1)) * 32 + ((x0 + pdx[0]) >> 1)) * 256 + 8 * lhalf
                         : static_cast<bf16_t*>(p.y) + (((size_t)b * 64 + y0 + pdy[0]) * 64 + x0 + pdx[0]) * 256 + 8 * lhalf;
        u32x4 ylo, yhi;
        auto y2side = [&](auto sc) {
            constexpr int S = decltype(sc)::value, m = S >> 1;
            if constexpr ((S & 1) == 0) {
                unsigned q[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) q[i] = y[0][8 * m + i];
                tile_to_rows(q, ylo, yhi);
                if (mine) *reinterpret_cast<u32x4*>(YO + 32 * m) = ylo;
            } else {
                if (mine) *reinterpret_cast<u32x4*>(YO + 32 * m + 16) = yhi;
            }
        };
        chain_block<1, 18, S_B2, 4, false>(ring, y, rb, idf, ones, acc, ps, y2side, noside);
        ring.finish_tile();
        // everything this wave has in flight lands: the phantom groups behind the last one (or the next tile's first groups and p halo), y2
        wait_vmcnt<0>();
        stamp(5);
        // t1out: 128 channels of the lane's pixel, packed now, stored under the next tile's block 0 (t1side)
        pend_ptr = static_cast<bf16_t*>(p.t1out) + ((size_t)b * 64 * 64 + (size_t)(y0 + pdy[0]) * 64 + x0 + pdx[0]) * 128 + 8 * lhalf;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            unsigned q[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) q[i] = relu2_bf16(acc[m][2 * i], acc[m][2 * i + 1]);
            tile_to_rows(q, pend[2 * m], pend[2 * m + 1]);
        }
        if (k + 1 == nk) {
#pragma unroll
            for (int j = 0; j < 8; ++j) *reinterpret_cast<u32x4*>(pend_ptr + 32 * (j >> 1) + 16 * (j & 1)) = pend[j];
        }
    }
    if (p.ts && k == kstamp) {
        if (!PERSIST) wait_vmcnt<0>();
        stamp(6);
        if (tid == 0) {
#pragma unroll
            for (int i = 0; i < 7; ++i) p.ts[(size_t)blockIdx.x * 8 + i] = stamps[i];
        }
    }
    if (PERSIST && k + 1 < nk) set_tile(tile_of(k + 1));
  }
}

// ---------------------------------------------------------------- the fragment stream (pack time)
// one block per fragment: lane l holds A[row0 + (l & 31)][col0 + kmap(8 (l >> 5) + e)], e = 0 .. 7; a bias fragment holds (hi, lo) of
// the fp32 bias at k = 0, 1 of the lower lanes
__global__ void res2_chain_pack_kernel(Res2ChainPackArgs a) {
    const int f = blockIdx.x, lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    const bf16_t* W = nullptr;
    const float* Bv = nullptr;
    int K = 0, row0 = 0, col0 = 0;
    bool perm = false;
    const float* Bv2 = nullptr;
    if (f < S_B0) {
        if (f < 2) Bv = a.b1[0] + 32 * f;
        else { const int j = f - 2; W = a.w1[0]; K = 64; row0 = 32 * (j & 1); col0 = 16 * (j >> 1); }
    } else {
        const int blk = f < S_B1 ? 0 : f < S_B2 ? 1 : 2;
        const int j = f - (blk == 0 ? S_B0 : blk == 1 ? S_B1 : S_B2);
        const int CS = blk == 0 ? 9 : 5, OB1 = O_C3 + 8 * CS;
        if (j < O_W2) Bv = a.b2[blk] + 32 * j;
        else if (j < O_C3) { const int q = j - O_W2; W = a.w2[blk]; K = 576; row0 = 32 * (q & 1); col0 = 16 * (q >> 1); }
        else if (j < OB1) {
            const int q = j - O_C3, m = q / CS, r = q % CS;
            if (r == 0) { Bv = a.b3[blk] + 32 * m; if (blk == 0) Bv2 = a.bd + 32 * m; }
            else if (r < 5) { W = a.w3[blk]; K = 64; row0 = 32 * m; col0 = 16 * (r - 1); perm = true; }
            else { W = a.wd; K = 64; row0 = 32 * m; col0 = 16 * (r - 5); }
        }
        else if (blk < 2) { const int q = j - OB1; if (q < 2) Bv = a.b1[blk + 1] + 32 * q; else { W = a.w1[blk + 1]; K = 256; row0 = 32 * ((q - 2) & 1); col0 = 16 * ((q - 2) >> 1); perm = true; } }
        else { const int q = j - OB1; if (q < 4) Bv = a.b1[3] + 32 * q; else { W = a.w1[3]; K = 256; row0 = 32 * ((q - 4) & 3); col0 = 16 * ((q - 4) >> 2); perm = true; } }
    }
    bf16_t v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 8 * h + e;
        if (Bv) {
            const float bv = Bv[i] + (Bv2 ? Bv2[i] : 0.f);
            const bf16_t hi = f32_to_bf16(bv);
            v[e] = k == 0 ? hi : k == 1 ? f32_to_bf16(bv - bf16_to_f32(hi)) : (bf16_t)0;
        } else {
            const int kk = perm ? (e & 3) + 8 * (e >> 2) + 4 * h : k;
            v[e] = W[(size_t)(row0 + i) * K + col0 + kk];
        }
    }
    uint4 o;
    o.x = v[0] | ((unsigned)v[1] << 16); o.y = v[2] | ((unsigned)v[3] << 16); o.z = v[4] | ((unsigned)v[5] << 16); o.w = v[6] | ((unsigned)v[7] << 16);
    *reinterpret_cast<uint4*>(static_cast<bf16_t*>(a.out) + ((size_t)f * 64 + lane) * 8) = o;
}

size_t res2_chain_stream_bytes() { return (size_t)RC_NG * RC_GB; }

void launch_res2_chain_pack(const Res2ChainPackArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(res2_chain_pack_kernel, dim3(RC_NF), dim3(64), 0, st, a);
}

bool res2_chain_ok(const Res2ChainArgs& a) { return a.x && a.y && a.t1out && a.wstream && a.zeros && a.B > 0; }

void launch_res2_chain(const Res2ChainArgs& a, hipStream_t st) {
    // persistent form (tunable R2C_PERSIST, default on): one workgroup per CU walks over its tiles (8 x 16 pixels each) with the weight ring
    // running across the tile boundaries and the next tile's p halo prefetched during block 2; R2C_GRID workgroups (0 = one per CU)
    static int ncu = 0;
    if (!ncu) { int dev = 0; hipDeviceProp_t pr; (void)hipGetDevice(&dev); ncu = hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256; }
    const int ntiles = a.B * 32;
    const bool persist = tune_get("R2C_PERSIST", 1) != 0;
    int gmax = tune_get("R2C_GRID", 0);
    if (gmax <= 0) gmax = ncu;
    const int grid = persist ? std::min(ntiles, gmax) : ntiles;
    ConvArgs d{};
    d.B = a.B; d.H = 64; d.W = 64; d.Ho = 64; d.Wo = 64; d.Cin = 64; d.Cout = 256; d.KH = -2; d.KW = -2; d.stride = 1;   // KH = -2: the res2 stage row of the layer report
    void* tok = prof_begin(d, 2, st);
    if (persist) {
        if (a.y_s2) hipLaunchKernelGGL((res2_chain_kernel<true, true>), dim3(grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((res2_chain_kernel<false, true>), dim3(grid), dim3(256), 0, st, a);
    } else {
        if (a.y_s2) hipLaunchKernelGGL((res2_chain_kernel<true, false>), dim3(grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((res2_chain_kernel<false, false>), dim3(grid), dim3(256), 0, st, a);
    }
    prof_end(tok, st);
}

}  // namespace ivosw
