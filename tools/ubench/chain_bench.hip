// the register-chained bottleneck body as a micro-benchmark.
//
// One identity bottleneck of res2 (3x3 on a t1 raster -> conv3 + residual -> ReLU -> the NEXT block's conv1 -> t1' raster) on a
// 16 x 16 pixel region, with the dataflow LAB_NOTES round 5 sized on paper:
//   * 4 waves, ONE per SIMD (512 registers each); a wave owns two 32-pixel tiles with ALL channels;
//   * the pointwise chain stays in registers: after bf16 packing, the accumulator tile of a 32x32x16 MFMA (weights as the A
//     operand: a lane holds 16 channels of ONE pixel) IS the B-operand fragment of the next contraction - no permlane, no LDS - because
//     the next layer's weights are packed with the matching K permutation pi(8 h + e) = (e & 3) + 8 (e >> 2) + 4 h per 16 channels;
//   * residual = 2 MFMAs per tile against identity fragments (the packed y registers are B operands already), bias = 1 MFMA per tile
//     against a (hi, lo) bf16 split of the fp32 bias: the epilogue of a tile is 8 v_cvt_pk + 8 v_pk_max and nothing else;
//   * only the 3x3's input goes through LDS (padded 144-byte raster rows, as in res2_stage.hip);
//   * EVERY weight fragment comes through ONE LDS ring per workgroup: the host packs the fragments of a block in consumption order
//     (1 KB each, lane-linear), groups of 8 are fetched by LDS-DMA (2 pieces per wave and group) three groups ahead of their use, one
//     s_barrier per group (vmcnt(6) counted: never drained).
// Reports cycles per block pass (256 pixels), cycles per MFMA, and checks pass 1 against a host reference of the same bf16 roundings.
// Compare with today's res2_stage kernel (profiles/r04_res2_groups_probe.txt, block 1 = B1 + C1 + D1: 16.6 k cycles for 6 pixel tiles).
// Build: hipcc -w --offload-arch=gfx950 -O3 -std=c++17 -o chain_bench chain_bench.hip
#include "../../ivos-w_amd/csrc/mfma_tile.h"
#include <math.h>
#include <stdlib.h>
#include <type_traits>
#include <vector>
using namespace ivosw;
namespace ivosw { void set_error(const char*, ...) {} }

constexpr int CB_GROUPS = 6, CB_GROUP_BYTES = 8192;
constexpr int CB_RING = CB_GROUPS * CB_GROUP_BYTES;
constexpr int CB_T1R = 144, CB_RW = 18;
constexpr int CB_RAST = CB_RW * CB_RW * CB_T1R;
constexpr int CB_LDS = CB_RING + 2 * CB_RAST;
// fragment stream of one block pass
constexpr int F_B2 = 0, F_W2 = 2, F_C3 = 74, F_B1 = 114, F_W1 = 116, F_END = 148, CB_NG = 19;
#ifndef CB_DEPTH
#define CB_DEPTH 2
#endif

struct ChainArgs {
    const void* wstream;        // CB_NG groups of 8 fragments of 1 KB
    const bf16_t* in_t1;        // [18 * 18][64]
    const bf16_t* in_y;         // [256 slots][256]
    bf16_t* out_t1;             // [workgroup][256][64]
    bf16_t* out_y;              // [workgroup][256][256]
    int passes;
    unsigned long long* ts;     // [workgroup][2]
};

template <int I, int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}
template <int N>
__device__ __forceinline__ void lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N < 15 ? N : 15) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void pin() { __builtin_amdgcn_sched_barrier(0); }

template <int ABL>
__device__ __forceinline__ f32x16 mm(u32x4 a, u32x4 b, f32x16 c) {
    if constexpr (ABL & 2) {
        asm volatile("" ::"v"(a), "v"(b));
        return c;
    } else return mfma_bf16(a, b, c);
}

// ABL: 1 = no LDS-DMA in the loop, 2 = no MFMA, 4 = no epilogue VALU (accumulators kept alive), 8 = no raster reads
template <int ABL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void chain_block_kernel(ChainArgs p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[CB_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhalf = lane >> 5;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    constexpr int D = CB_DEPTH;

    // ---------------- inputs (plain loads; everything is waited for before the first LDS-DMA goes out)
    for (int i = tid; i < CB_RW * CB_RW * 8; i += 256) {
        const int px = i >> 3, c = i & 7;
        *reinterpret_cast<uint4*>(lds + CB_RING + px * CB_T1R + c * 16) = *reinterpret_cast<const uint4*>(p.in_t1 + px * 64 + c * 8);
        *reinterpret_cast<uint4*>(lds + CB_RING + CB_RAST + px * CB_T1R + c * 16) = make_uint4(0, 0, 0, 0);
    }
    unsigned y[2][64];          // [pixel tile][k-step t: 4 t .. 4 t + 3]: channels 16 t + pi(8 h + e)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const bf16_t* src = p.in_y + (size_t)((2 * wave + i) * 32 + lrow) * 256;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const uint2 lo = *reinterpret_cast<const uint2*>(src + 16 * t + 4 * lhalf);
            const uint2 hi = *reinterpret_cast<const uint2*>(src + 16 * t + 8 + 4 * lhalf);
            y[i][4 * t] = lo.x; y[i][4 * t + 1] = lo.y; y[i][4 * t + 2] = hi.x; y[i][4 * t + 3] = hi.y;
        }
    }
    // constant fragments: identity halves (A operand), ones (B operand of the bias MFMA)
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
    __syncthreads();
    wait_vmcnt<0>();

    // ---------------- the weight ring
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wstream), 0, CB_NG * CB_GROUP_BYTES, 0x00020000);
    const int vpiece = wave * 2048 + lane * 16;
    auto issue_group = [&](int g, int slot) {       // this wave's two pieces of group g into ring slot `slot`
        if (ABL & 1) return;
        unsigned char* dst = lds + slot * CB_GROUP_BYTES + wave * 2048;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)dst, 16, vpiece, g * CB_GROUP_BYTES, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)(dst + 1024), 16, vpiece + 1024, g * CB_GROUP_BYTES, 0, 0);
    };
    int gi = 0, slot = 0;                            // the group the NEXT boundary opens, and its slot
    unsigned vcur = 0;
    auto boundary = [&]() {
        if (!(ABL & 1)) wait_vmcnt<6>();
        pin();
        __builtin_amdgcn_s_barrier();
        pin();
        int g4 = gi + 4, s4 = slot + 4;
        if (g4 >= CB_NG) g4 -= CB_NG;
        if (s4 >= CB_GROUPS) s4 -= CB_GROUPS;
        issue_group(g4, s4);
        vcur = lds_base + slot * CB_GROUP_BYTES + lane * 16;
        gi = gi + 1 == CB_NG ? 0 : gi + 1;
        slot = slot + 1 == CB_GROUPS ? 0 : slot + 1;
        pin();
    };
    auto rdA = [&](auto fc) -> u32x4 {
        constexpr int F = decltype(fc)::value;
        if constexpr (F % 8 == 0) boundary();
        return lds_read_b128_o<(F % 8) * 1024>(vcur);
    };
    if (ABL & 1) {
        // static ring: every slot holds group (slot) once
        for (int g = 0; g < CB_GROUPS; ++g) {
            unsigned char* dst = lds + g * CB_GROUP_BYTES + wave * 2048;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)dst, 16, vpiece, g * CB_GROUP_BYTES, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)(dst + 1024), 16, vpiece + 1024, g * CB_GROUP_BYTES, 0, 0);
        }
        wait_vmcnt<0>();
        __syncthreads();
    } else {
        issue_group(0, 0); issue_group(1, 1); issue_group(2, 2); issue_group(3, 3);
    }

    unsigned long long t0 = 0;
    if (p.ts) { t0 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

    // pixel of slot (2 wave + i) * 32 + lrow in the 16 x 16 region: row 2 (2 wave + i) + (lrow >> 4), column lrow & 15
    for (int pass = 0; pass < p.passes; ++pass) {
        const unsigned rin = lds_base + CB_RING + (pass & 1) * CB_RAST, rout = lds_base + CB_RING + ((pass & 1) ^ 1) * CB_RAST;
        unsigned rb[2], wb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int py = 2 * (2 * wave + i) + (lrow >> 4), px = lrow & 15;
            rb[i] = rin + (py * CB_RW + px) * CB_T1R + lhalf * 16;                   // tap (0, 0) = the pixel up-left
            wb[i] = rout + ((py + 1) * CB_RW + px + 1) * CB_T1R + lhalf * 8;
        }
        // ======================================================== 3x3: t2 = relu(b2 + W2 * t1)
        f32x16 acc[2][2];       // [pixel tile][channel tile]
        unsigned t2[2][16];
        {
            u32x4 a[D + 1][2], b[D + 1][2];
            auto rdstep = [&](auto sc, auto qc) {          // read q (A0, B0, A1, B1) of k-step s
                constexpr int S = decltype(sc)::value, Q = decltype(qc)::value, buf = S % (D + 1);
                constexpr int tap = S >> 2, kk = S & 3, toff = ((tap / 3) * CB_RW + (tap % 3)) * CB_T1R + kk * 32;
                if constexpr (Q == 0) a[buf][0] = rdA(std::integral_constant<int, F_W2 + 2 * S>{});
                else if constexpr (Q == 2) a[buf][1] = rdA(std::integral_constant<int, F_W2 + 2 * S + 1>{});
                else if constexpr (Q == 1) b[buf][0] = (ABL & 8) ? lds_read_b128_o<0>(rb[0]) : lds_read_b128_o<toff>(rb[0]);
                else b[buf][1] = (ABL & 8) ? lds_read_b128_o<0>(rb[1]) : lds_read_b128_o<toff>(rb[1]);
            };
            const u32x4 bf0 = rdA(std::integral_constant<int, F_B2>{}), bf1 = rdA(std::integral_constant<int, F_B2 + 1>{});
            sfor<0, D>([&](auto sc) { sfor<0, 4>([&](auto qc) { rdstep(sc, qc); }); });
            lgkm<4 * D>();
            const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            acc[0][0] = mm<ABL>(bf0, ones, z); acc[1][0] = mm<ABL>(bf0, ones, z);
            acc[0][1] = mm<ABL>(bf1, ones, z); acc[1][1] = mm<ABL>(bf1, ones, z);
            sfor<0, 36>([&](auto sc) {
                constexpr int S = decltype(sc)::value, buf = S % (D + 1);
                constexpr bool more = S + D < 36;
                constexpr int full = 4 * (S + D < 36 ? S + D : 36);          // reads of all steps below min(S + D, 36) are issued
                // allowed outstanding = issued - (needed index + 1)
                lgkm<full - (4 * S + 2)>();
                acc[0][0] = mm<ABL>(a[buf][0], b[buf][0], acc[0][0]);
                if constexpr (more) rdstep(std::integral_constant<int, S + D>{}, std::integral_constant<int, 0>{});
                lgkm<full + (more ? 1 : 0) - (4 * S + 3)>();
                acc[0][1] = mm<ABL>(a[buf][1], b[buf][0], acc[0][1]);
                if constexpr (more) rdstep(std::integral_constant<int, S + D>{}, std::integral_constant<int, 1>{});
                lgkm<full + (more ? 2 : 0) - (4 * S + 4)>();
                acc[1][0] = mm<ABL>(a[buf][0], b[buf][1], acc[1][0]);
                if constexpr (more) rdstep(std::integral_constant<int, S + D>{}, std::integral_constant<int, 2>{});
                pin();
                acc[1][1] = mm<ABL>(a[buf][1], b[buf][1], acc[1][1]);
                if constexpr (more) rdstep(std::integral_constant<int, S + D>{}, std::integral_constant<int, 3>{});
                pin();
            });
        }
        // ======================================================== conv3 + residual, M tile by M tile; t2's epilogue rides under tile 0's
        // bias / residual MFMAs, the epilogue of tile m under the MFMAs of tile m + 1
        {
            u32x4 c[2][5];
            f32x16 a3[2][2];    // [m & 1][pixel tile]
            auto rdc = [&](auto mc, auto jc) {
                constexpr int M = decltype(mc)::value, J = decltype(jc)::value;
                c[M & 1][J] = rdA(std::integral_constant<int, F_C3 + 5 * M + J>{});
            };
            sfor<0, 5>([&](auto jc) { rdc(std::integral_constant<int, 0>{}, jc); });
            auto epi_t2 = [&](auto ic, auto mc, auto hc) {          // half (8 values -> 4 registers) of t2 tile (pixel tile I, channel tile M)
                constexpr int I = decltype(ic)::value, M = decltype(mc)::value, H = decltype(hc)::value;
                if (ABL & 4) { t2[I][8 * M + 4 * H] = __float_as_uint(acc[I][M][8 * H]); t2[I][8 * M + 4 * H + 1] = t2[I][8 * M + 4 * H + 2] = t2[I][8 * M + 4 * H + 3] = 0x3f803f80u; return; }
#pragma unroll
                for (int q = 0; q < 4; ++q) t2[I][8 * M + 4 * H + q] = relu2_bf16(acc[I][M][8 * H + 2 * q], acc[I][M][8 * H + 2 * q + 1]);
            };
            auto epi_y = [&](auto mc, auto ic, auto hc) {           // half of y tile (M, pixel tile I) from a3[M & 1][I]
                constexpr int M = decltype(mc)::value, I = decltype(ic)::value, H = decltype(hc)::value;
                if (ABL & 4) { y[I][8 * M + 4 * H] = __float_as_uint(a3[M & 1][I][8 * H]); return; }
#pragma unroll
                for (int q = 0; q < 4; ++q) y[I][8 * M + 4 * H + q] = relu2_bf16(a3[M & 1][I][8 * H + 2 * q], a3[M & 1][I][8 * H + 2 * q + 1]);
            };
            const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            sfor<0, 8>([&](auto mc) {
                constexpr int M = decltype(mc)::value, cb = M & 1;
                using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
                using MP = std::integral_constant<int, M - 1>; using MN = std::integral_constant<int, M + 1>;
                // filler k (0 .. 7) of this tile: t2's epilogue under tile 0, tile M - 1's y epilogue otherwise
                auto fill = [&](auto kc) {
                    constexpr int K = decltype(kc)::value;
                    if constexpr (M == 0) {
                        if constexpr (K < 8) epi_t2(std::integral_constant<int, (K >> 1) & 1>{}, std::integral_constant<int, (K >> 2)>{}, std::integral_constant<int, K & 1>{});
                    } else if constexpr (K >= 2 && K < 6) epi_y(MP{}, std::integral_constant<int, ((K - 2) >> 1)>{}, std::integral_constant<int, (K - 2) & 1>{});
                };
                lgkm<4>();                                   // this tile's bias fragment has landed (its four weight fragments may not)
                a3[cb][0] = mm<ABL>(c[cb][0], ones, z); fill(I0{}); pin();
                a3[cb][1] = mm<ABL>(c[cb][0], ones, z); fill(I1{});
                if constexpr (M < 7) rdc(MN{}, std::integral_constant<int, 0>{});
                pin();
                a3[cb][0] = mm<ABL>(idf[0], *reinterpret_cast<u32x4*>(&y[0][8 * M]), a3[cb][0]); fill(std::integral_constant<int, 2>{}); pin();
                a3[cb][1] = mm<ABL>(idf[0], *reinterpret_cast<u32x4*>(&y[1][8 * M]), a3[cb][1]); fill(std::integral_constant<int, 3>{});
                if constexpr (M < 7) rdc(MN{}, std::integral_constant<int, 1>{});
                pin();
                a3[cb][0] = mm<ABL>(idf[1], *reinterpret_cast<u32x4*>(&y[0][8 * M + 4]), a3[cb][0]); fill(std::integral_constant<int, 4>{}); pin();
                a3[cb][1] = mm<ABL>(idf[1], *reinterpret_cast<u32x4*>(&y[1][8 * M + 4]), a3[cb][1]); fill(std::integral_constant<int, 5>{});
                if constexpr (M < 7) rdc(MN{}, std::integral_constant<int, 2>{});
                pin();
                if constexpr (M == 0) { fill(std::integral_constant<int, 6>{}); fill(std::integral_constant<int, 7>{}); }
                lgkm<(M < 7 ? 3 : 0)>();                     // all of this tile's fragments (the three reads of the next tile may be out)
                sfor<0, 4>([&](auto kc) {
                    constexpr int KS = decltype(kc)::value;
                    a3[cb][0] = mm<ABL>(c[cb][1 + KS], *reinterpret_cast<u32x4*>(&t2[0][4 * KS]), a3[cb][0]); pin();
                    a3[cb][1] = mm<ABL>(c[cb][1 + KS], *reinterpret_cast<u32x4*>(&t2[1][4 * KS]), a3[cb][1]);
                    if constexpr (M < 7 && KS < 2) rdc(MN{}, std::integral_constant<int, 3 + KS>{});
                    pin();
                });
            });
            // the last tile's epilogue rides under conv1''s bias MFMAs below
            // ==================================================== the next block's conv1: t1' = relu(b1 + W1' y) -> raster
            {
                u32x4 w[D + 1][2];
                auto rdw = [&](auto sc, auto qc) {
                    constexpr int S = decltype(sc)::value, Q = decltype(qc)::value;
                    w[S % (D + 1)][Q] = rdA(std::integral_constant<int, F_W1 + 2 * S + Q>{});
                };
                const u32x4 bf0 = rdA(std::integral_constant<int, F_B1>{}), bf1 = rdA(std::integral_constant<int, F_B1 + 1>{});
                sfor<0, D>([&](auto sc) { rdw(sc, std::integral_constant<int, 0>{}); rdw(sc, std::integral_constant<int, 1>{}); });
                lgkm<2 * D>();
                acc[0][0] = mm<ABL>(bf0, ones, z);
                epi_y(std::integral_constant<int, 7>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}); pin();
                acc[1][0] = mm<ABL>(bf0, ones, z);
                epi_y(std::integral_constant<int, 7>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}); pin();
                acc[0][1] = mm<ABL>(bf1, ones, z);
                epi_y(std::integral_constant<int, 7>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}); pin();
                acc[1][1] = mm<ABL>(bf1, ones, z);
                epi_y(std::integral_constant<int, 7>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}); pin();
                sfor<0, 16>([&](auto sc) {
                    constexpr int S = decltype(sc)::value, buf = S % (D + 1);
                    constexpr bool more = S + D < 16;
                    constexpr int full = 2 * (S + D < 16 ? S + D : 16);
                    lgkm<full - (2 * S + 1)>();
                    acc[0][0] = mm<ABL>(w[buf][0], *reinterpret_cast<u32x4*>(&y[0][4 * S]), acc[0][0]);
                    if constexpr (more) rdw(std::integral_constant<int, S + D>{}, std::integral_constant<int, 0>{});
                    pin();
                    acc[1][0] = mm<ABL>(w[buf][0], *reinterpret_cast<u32x4*>(&y[1][4 * S]), acc[1][0]);
                    if constexpr (more) rdw(std::integral_constant<int, S + D>{}, std::integral_constant<int, 1>{});
                    lgkm<full + (more ? 2 : 0) - (2 * S + 2)>();
                    acc[0][1] = mm<ABL>(w[buf][1], *reinterpret_cast<u32x4*>(&y[0][4 * S]), acc[0][1]); pin();
                    acc[1][1] = mm<ABL>(w[buf][1], *reinterpret_cast<u32x4*>(&y[1][4 * S]), acc[1][1]); pin();
                });
            }
        }
        // t1' epilogue: ReLU, bf16, 8-byte pieces of 4 consecutive channels into the other raster
        if (!(ABL & 4)) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        u32x2 pk;
                        pk.x = relu2_bf16(acc[i][m][4 * g], acc[i][m][4 * g + 1]);
                        pk.y = relu2_bf16(acc[i][m][4 * g + 2], acc[i][m][4 * g + 3]);
                        lds_write_b64(wb[i] + (m * 4 + g) * 16, pk);
                    }
        } else {
            asm volatile("" ::"v"(acc[0][0][0]), "v"(acc[0][1][0]), "v"(acc[1][0][0]), "v"(acc[1][1][0]));
        }
        lds_wait();
        __builtin_amdgcn_s_barrier();
        pin();
    }
    wait_vmcnt<0>();
    if (p.ts && tid == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        p.ts[2 * blockIdx.x] = t0; p.ts[2 * blockIdx.x + 1] = t1;
    }
    // ---------------- dump: y (natural channel order) and the raster the last pass wrote
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        bf16_t* dst = p.out_y + ((size_t)blockIdx.x * 256 + (2 * wave + i) * 32 + lrow) * 256;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            *reinterpret_cast<uint2*>(dst + 16 * t + 4 * lhalf) = make_uint2(y[i][4 * t], y[i][4 * t + 1]);
            *reinterpret_cast<uint2*>(dst + 16 * t + 8 + 4 * lhalf) = make_uint2(y[i][4 * t + 2], y[i][4 * t + 3]);
        }
    }
    {
        const unsigned char* r = lds + CB_RING + (p.passes & 1) * CB_RAST;
        for (int i = tid; i < 256 * 8; i += 256) {
            const int s = i >> 3, c = i & 7, py = s >> 4, px = s & 15;
            *reinterpret_cast<uint4*>(p.out_t1 + ((size_t)blockIdx.x * 256 + s) * 64 + c * 8) =
                *reinterpret_cast<const uint4*>(r + ((py + 1) * CB_RW + px + 1) * CB_T1R + c * 16);
        }
    }
}

// ================================================================= host
static inline uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static inline float bf2f(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
static uint64_t rng_s = 0x9e3779b97f4a7c15ull;
static inline uint32_t rnd() { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return (uint32_t)(rng_s >> 32); }
static inline float grand() { float s = 0.f; for (int i = 0; i < 12; ++i) s += (rnd() >> 8) * (1.0f / 16777216.0f); return s - 6.0f; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static inline int pi_k(int k) { const int h = k >> 3, e = k & 7; return (e & 3) + 8 * (e >> 2) + 4 * h; }

struct Weights {
    std::vector<uint16_t> w2;   // [64][9][64]
    std::vector<uint16_t> w3;   // [256][64]
    std::vector<uint16_t> w1;   // [64][256]
    std::vector<float> b2, b3, b1;
};

// fragment (1 KB): lane (i = lane & 31, h = lane >> 5) holds A[i][8 h + e], e = 0 .. 7
template <typename F>
static void put_frag(std::vector<uint16_t>& s, int f, F&& elem) {
    for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 8; ++e) s[(size_t)f * 512 + lane * 8 + e] = elem(lane & 31, 8 * (lane >> 5) + e);
}
static void put_bias(std::vector<uint16_t>& s, int f, const float* b) {
    put_frag(s, f, [&](int i, int k) -> uint16_t {
        const uint16_t hi = f2bf(b[i]);
        if (k == 0) return hi;
        if (k == 1) return f2bf(b[i] - bf2f(hi));
        return 0;
    });
}

int main(int argc, char** argv) {
    const int passes = argc > 1 ? atoi(argv[1]) : 200;
    const int grid = argc > 2 ? atoi(argv[2]) : 256;
    Weights W;
    W.w2.resize(64 * 9 * 64); W.w3.resize(256 * 64); W.w1.resize(64 * 256);
    W.b2.resize(64); W.b3.resize(256); W.b1.resize(64);
    for (auto& v : W.w2) v = f2bf(grand() * sqrtf(2.0f / 576));
    for (auto& v : W.w3) v = f2bf(grand() * sqrtf(1.0f / 64));
    for (auto& v : W.w1) v = f2bf(grand() * sqrtf(2.0f / 256) * 0.7f);
    for (auto& v : W.b2) v = grand() * 0.1f;
    for (auto& v : W.b3) v = grand() * 0.1f;
    for (auto& v : W.b1) v = grand() * 0.1f;
    std::vector<uint16_t> t1(CB_RW * CB_RW * 64), yin(256 * 256);
    for (auto& v : t1) { const float g = grand(); v = f2bf(g > 0 ? g : 0.f); }
    for (auto& v : yin) { const float g = grand(); v = f2bf(g > 0 ? g : 0.f); }

    // the stream
    std::vector<uint16_t> st((size_t)CB_NG * 8 * 512, 0);
    for (int m = 0; m < 2; ++m) put_bias(st, F_B2 + m, W.b2.data() + 32 * m);
    for (int ks = 0; ks < 36; ++ks)
        for (int m = 0; m < 2; ++m)
            put_frag(st, F_W2 + 2 * ks + m, [&](int i, int k) { return W.w2[((size_t)(32 * m + i) * 9 + (ks >> 2)) * 64 + 16 * (ks & 3) + k]; });
    for (int m = 0; m < 8; ++m) {
        put_bias(st, F_C3 + 5 * m, W.b3.data() + 32 * m);
        for (int ks = 0; ks < 4; ++ks) put_frag(st, F_C3 + 5 * m + 1 + ks, [&](int i, int k) { return W.w3[(size_t)(32 * m + i) * 64 + 16 * ks + pi_k(k)]; });
    }
    for (int m = 0; m < 2; ++m) put_bias(st, F_B1 + m, W.b1.data() + 32 * m);
    for (int ks = 0; ks < 16; ++ks)
        for (int m = 0; m < 2; ++m) put_frag(st, F_W1 + 2 * ks + m, [&](int i, int k) { return W.w1[(size_t)(32 * m + i) * 256 + 16 * ks + pi_k(k)]; });

    void *dst_, *dt1, *dy, *dot1, *doy, *dts;
    CK(hipMalloc(&dst_, st.size() * 2)); CK(hipMalloc(&dt1, t1.size() * 2)); CK(hipMalloc(&dy, yin.size() * 2));
    CK(hipMalloc(&dot1, (size_t)grid * 256 * 64 * 2)); CK(hipMalloc(&doy, (size_t)grid * 256 * 256 * 2)); CK(hipMalloc(&dts, (size_t)grid * 16));
    CK(hipMemcpy(dst_, st.data(), st.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dt1, t1.data(), t1.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dy, yin.data(), yin.size() * 2, hipMemcpyHostToDevice));
    ChainArgs a{dst_, (const bf16_t*)dt1, (const bf16_t*)dy, (bf16_t*)dot1, (bf16_t*)doy, 1, (unsigned long long*)dts};

    // ---- one pass against the host reference (bias split hi + lo as on the device; fp32 accumulation in a different order)
    int bad = 0;
    {
        hipLaunchKernelGGL(chain_block_kernel<0>, dim3(grid), dim3(256), 0, 0, a);
        CK(hipDeviceSynchronize());
        std::vector<uint16_t> oy((size_t)grid * 256 * 256), ot((size_t)grid * 256 * 64);
        CK(hipMemcpy(oy.data(), doy, oy.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ot.data(), dot1, ot.size() * 2, hipMemcpyDeviceToHost));
        std::vector<float> t2(256 * 64), yy(256 * 256), tn(256 * 64);
        for (int s = 0; s < 256; ++s) {
            const int py = s >> 4, px = s & 15;
            for (int o = 0; o < 64; ++o) {
                double acc = W.b2[o];
                for (int tap = 0; tap < 9; ++tap)
                    for (int c = 0; c < 64; ++c)
                        acc += (double)bf2f(W.w2[((size_t)o * 9 + tap) * 64 + c]) * bf2f(t1[((py + tap / 3) * CB_RW + px + tap % 3) * 64 + c]);
                t2[s * 64 + o] = bf2f(f2bf(acc > 0 ? (float)acc : 0.f));
            }
            for (int o = 0; o < 256; ++o) {
                double acc = W.b3[o] + bf2f(yin[s * 256 + o]);
                for (int c = 0; c < 64; ++c) acc += (double)bf2f(W.w3[(size_t)o * 64 + c]) * t2[s * 64 + c];
                yy[s * 256 + o] = bf2f(f2bf(acc > 0 ? (float)acc : 0.f));
            }
            for (int o = 0; o < 64; ++o) {
                double acc = W.b1[o];
                for (int c = 0; c < 256; ++c) acc += (double)bf2f(W.w1[(size_t)o * 256 + c]) * yy[s * 256 + c];
                tn[s * 64 + o] = bf2f(f2bf(acc > 0 ? (float)acc : 0.f));
            }
        }
        double wy = 0, wt = 0;
        const int wgs[3] = {0, grid / 2, grid - 1};
        for (int wi = 0; wi < 3; ++wi) {
            const int wg = wgs[wi];
            for (int i = 0; i < 256 * 256; ++i) {
                const double e = fabs(bf2f(oy[(size_t)wg * 65536 + i]) - yy[i]) / (fabs(yy[i]) + 1.0);
                if (e > wy) wy = e;
                if (e > 2e-2) { if (bad < 8) printf("  y MISMATCH wg %d slot %d ch %d got %f want %f\n", wg, i >> 8, i & 255, bf2f(oy[(size_t)wg * 65536 + i]), yy[i]); ++bad; }
            }
            for (int i = 0; i < 256 * 64; ++i) {
                const double e = fabs(bf2f(ot[(size_t)wg * 16384 + i]) - tn[i]) / (fabs(tn[i]) + 1.0);
                if (e > wt) wt = e;
                if (e > 2e-2) { if (bad < 16) printf("  t1' MISMATCH wg %d slot %d ch %d got %f want %f\n", wg, i >> 6, i & 63, bf2f(ot[(size_t)wg * 16384 + i]), tn[i]); ++bad; }
            }
        }
        printf("check (1 pass, 3 workgroups): worst rel err y %.2e  t1' %.2e, %d bad\n", wy, wt, bad);
    }
    // ---- timing
    auto timeit = [&](auto abl, const char* name) {
        constexpr int A = decltype(abl)::value;
        ChainArgs b = a; b.passes = passes;
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(chain_block_kernel<A>, dim3(grid), dim3(256), 0, 0, b);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        const int reps = 5;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(chain_block_kernel<A>, dim3(grid), dim3(256), 0, 0, b);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> ts((size_t)grid * 2);
        CK(hipMemcpy(ts.data(), dts, ts.size() * 8, hipMemcpyDeviceToHost));
        double cyc = 0;
        for (int g = 0; g < grid; ++g) cyc += (double)(ts[2 * g + 1] - ts[2 * g]);
        cyc /= grid;
        const double us = ms * 1000.0 / reps;
        const double nmfma = 4 + 144 + 8 * 14 + 4 + 64;          // per wave and pass
        const double flop = 2.0 * 256 * (64.0 * 576 + 256 * 64 + 64 * 256) * passes * grid;   // algorithmic (no bias / identity MFMAs)
        printf("  %-28s %9.1f us  %8.0f cyc per pass  %6.2f cyc per MFMA  clock %4.0f MHz  %7.1f algorithmic TFLOP/s\n", name, us, cyc / passes, cyc / passes / nmfma,
               cyc / us, flop / us * 1e-6);
    };
    printf("chain block: %d passes x %d workgroups, 4 waves x 2 pixel tiles, ring depth %d, fragment lead %d k-steps\n", passes, grid, CB_GROUPS, CB_DEPTH);
    for (int r = 0; r < 2; ++r) {
        timeit(std::integral_constant<int, 0>{}, "full");
        timeit(std::integral_constant<int, 1>{}, "no DMA");
        timeit(std::integral_constant<int, 4>{}, "no epilogue VALU");
        timeit(std::integral_constant<int, 8>{}, "raster reads at one address");
        timeit(std::integral_constant<int, 2>{}, "no MFMA");
        timeit(std::integral_constant<int, 1 | 4 | 8>{}, "no DMA / epilogue / raster");
    }
    printf(bad ? "FAILED\n" : "ALL CHECKS PASSED\n");
    return bad != 0;
}
