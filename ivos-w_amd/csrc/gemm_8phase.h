// The 256 x 256 "8-phase" contraction (round 6): C[m][n] = act(sum_k A[m][k] * B[n][k] + bias[n]), bf16 in,
// fp32 accumulate, bf16 out.  Geometry and schedule follow the plain-HIP template the CDNA4 guide documents
// (cdna_hip_programming.md, "The 256^2 8-phase template"): the example source is not in this image, so the kernel below is
// re-derived from that section's table and rules.
//
//   * 512 threads = 8 waves as 2 (M) x 4 (N); a wave owns 128 x 64 of the tile = 8 x 4 tiles of v_mfma_f32_16x16x32_bf16, i.e.
//     128 accumulator registers; two waves per SIMD.
//   * BK = 64.  A K-tile of an operand is two HALF-TILES of 128 rows x 64 k = 16 KB; half h holds, for every wave, the rows of its
//     quadrant h (A: rows wr * 128 + h * 64 + [0, 64); B: columns wc * 64 + h * 32 + [0, 32)), so that one phase's fragment reads
//     touch exactly one half-tile per operand and a half-tile is dead as soon as its phase is over.  LDS = 2 buffers x (2 + 2)
//     half-tiles = 128 KB.  A half-tile is 16 sub-tiles of 16 rows x 32 k (1 KB, what ONE LDS-DMA wave-instruction writes), with
//     the 16-byte chunks of rows 8 .. 15 XOR-ed by 2 ("st_16x32": byte ^= ((byte >> 9) & 1) << 5): the four 16-lane groups of a
//     ds_read_b128 then hit 16 distinct slots of the 256-byte bank row.  The swizzle sits on the DMA's SOURCE address and on the
//     read address (the DMA destination is lane-linear).
//   * A K-tile is four phases of 16 MFMAs (one quadrant of the wave's output x K = 64); an iteration is two K-tiles = 8 phases.
//       phase 1 / 5: read B half 0 (4 x ds_read_b128) then A half 0 (8); lgkmcnt(8) in front of the phase's first barrier retires the B
//                    reads, so B half 0 of this buffer may be re-staged ONE phase later
//       phase 2 / 6: read B half 1 (4)           phase 3 / 7: read A half 1 (8)           phase 4 / 8: none (B half 0 is still in registers)
//     every phase: reads | stage ONE half-tile (2 x LDS-DMA per wave) | s_barrier | lgkmcnt(0) | 16 MFMA under s_setprio 1 | s_barrier.
//     Staging order (buffer.operand-half, K-tile): ph1 odd.A1(t+1) | ph2 even.B0(t+2) | ph3 even.A0 | ph4 even.B1 | ph5 even.A1 |
//     ph6 odd.B0(t+3) | ph7 odd.A0 | ph8 odd.B1: each half-tile is re-staged >= 2 phases after its last read (1 for B half 0, see
//     above), and s_waitcnt vmcnt(6) at phases 4 and 8 - three half-tiles stay in flight - retires everything staged up to phases
//     1 / 5, i.e. the whole odd / even buffer, ONE phase before its first read.  vmcnt is never 0 in the loop.
//   * The two wave rows run staggered by one barrier (wr == 1 passes one extra s_barrier in front of the loop and wr == 0 one
//     behind it): while one wave of a SIMD issues its 16 MFMAs its partner issues reads and DMA.
//   * K-tiles that do not exist (the run-ahead behind the last one) are staged out of range of the buffer descriptor: no memory
//     access, zeros, the counted waits never change.  K % 128 == 0.
//   * Tower use: an optional second pixel source supplies the upper K-tiles (a first block's conv3 | downsample: the block input sampled
//     at stride 2), an optional residual is added in the epilogue, and the weight rows of a half-tile are staged in a permuted order so
//     that a lane's four result tiles are 16 consecutive output channels of ONE pixel (two 16-byte stores; 8-byte stores cost 30 % of the
//     layer at K = 768, profiles/r06_gemm_8phase_v1.txt).
#pragma once
#include "mfma_tile.h"

namespace ivosw {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct G8Args {
    const bf16_t* A;      // [M][K1] K-major (pixels)
    const bf16_t* B;      // [N][K] K-major (weights), K = K1 + K2
    const float* bias;    // [N]
    bf16_t* C;            // [M][ldc]
    const bf16_t* R;      // optional residual [M][ldc], added before the ReLU (RES)
    int M, N, K, ldc;     // N % 256 == 0, K % 128 == 0, K1 % 64 == 0; rows beyond M are not stored (their operand rows read zeros)
    int relu;
    unsigned long long* ts;   // optional [workgroups][4] stamps: s_memtime start / after the K loop / end, s_memrealtime span
    // K-extension (a first block's conv3 | downsample as one contraction): K-tiles from K1 / 64 on come from A2, a tensor
    // [frames][H2][W2][K2] sampled at (oy * stride2, ox * stride2) for output pixel (oy, ox) of a Wo x Ho frame; K1 = K and A2 = null: none
    const bf16_t* A2;
    int K1, K2, Ho, Wo, H2, W2, stride2;
    int rev;              // tiles in descending order (tunable SNAKE)
};

constexpr int G8_HALF = 16384;            // one half-tile
constexpr int G8_BUF = 4 * G8_HALF;       // A0 | A1 | B0 | B1
constexpr int G8_LDS = 2 * G8_BUF;

__device__ __forceinline__ f32x4_t mfma16(u32x4 a, u32x4 b, f32x4_t c) {
    union { u32x4 u; bf16x8 v; } ua, ub;
    ua.u = a; ub.u = b;
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.v, ub.v, c, 0, 0, 0);
}

// ABL bits (micro-benchmark only): 1 = no LDS-DMA in the loop, 2 = no fragment reads in the loop, 4 = no global stores,
// 8 = no epilogue at all, 16 = no s_setprio, 32 = no stagger of the wave rows, 64 = linear LDS image (no swizzle)
template <int ABL, bool RES = false>
__global__ __launch_bounds__(512, 2) void gemm_8phase_kernel(G8Args p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[G8_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int nbn = p.N / 256;
    const int L0 = xcd_remap(blockIdx.x, gridDim.x), L = p.rev ? (int)gridDim.x - 1 - L0 : L0;
    const int tile_n = L % nbn, tile_m = L / nbn;
    const int m0 = tile_m * 256, n0 = tile_n * 256;
    const int NK = p.K / 64;
    unsigned long long t0 = 0, t1 = 0, r0 = 0;
    if (p.ts) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }

    // ---- staging: wave w fills row group w (16 rows) of a half-tile, both k halves (two 1-KB sub-tiles)
    // A: local row lr of half h is tile row (lr >> 6) * 128 + h * 64 + (lr & 63).  B: local row lr = (wc, nt, 4 cq + r) of half h is
    // column wc * 64 + 16 cq + 8 h + 4 nt + r, so that a lane's four result tiles (h, nt) hold 16 CONSECUTIVE output channels of its pixel
    const int K1 = p.K1, NK1 = K1 / 64;
    const long arows = min(256, p.M - m0);
    const unsigned abytes = (unsigned)(arows * K1 * 2), bbytes = 256u * p.K * 2;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A) + (size_t)m0 * K1, 0, abytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.B) + (size_t)n0 * p.K, 0, bbytes, 0x00020000);
    // the second pixel source: whole tensor in range of one descriptor (< 4 GB), rows addressed per lane
    const unsigned a2bytes = p.A2 ? (unsigned)min((size_t)0xfffffff0u, (size_t)((p.M + p.Ho * p.Wo - 1) / (p.Ho * p.Wo)) * p.H2 * p.W2 * p.K2 * 2) : 0u;
    const __amdgpu_buffer_rsrc_t ra2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A2 ? p.A2 : p.A), 0, a2bytes, 0x00020000);
    int voA, voB, voA2 = 0, hstepA2 = 0;
    {
        const int prow = lane >> 2, pch = lane & 3;
        const int lch = (ABL & 64) ? pch : (pch ^ ((prow >> 3) << 1));
        const int lr = wave * 16 + prow;                      // row of the half-tile
        const int trow = (lr >> 6) * 128 + (lr & 63);         // tile row (half 0)
        voA = trow * K1 * 2 + lch * 16;
        const int wi = lr & 31;
        voB = ((lr >> 5) * 64 + 16 * ((wi & 15) >> 2) + 4 * (wi >> 4) + (wi & 3)) * p.K * 2 + lch * 16;
        if (p.A2) {
            const int hw = p.Ho * p.Wo;
            auto off2 = [&](int m) { const int f = m / hw, q = m - f * hw, oy = q / p.Wo, ox = q - oy * p.Wo;
                                     return (unsigned)((((size_t)f * p.H2 + oy * p.stride2) * p.W2 + ox * p.stride2) * p.K2 * 2); };
            const int m = m0 + trow;
            voA2 = (int)(m < p.M ? off2(m) : 0xfffffff0u) + lch * 16;     // rows beyond M: out of range, zeros
            hstepA2 = (int)(off2(m0 + 64) - off2(m0));                     // uniform: 64 rows further is a whole number of raster rows / frames
        }
    }
    const int hstepA = 64 * K1 * 2, hstepB = 8 * p.K * 2;
    // stage half-tile (op: 0 = A, 1 = B; h) of K-tile kt into buffer kt & 1
    // (the instruction's immediate offset is added to the global address AND to the LDS address M0 points at: the second piece's
    // destination is therefore given 64 bytes low)
    auto stage = [&](int kt, int op, int h) {
        unsigned char* dst = lds + (kt & 1) * G8_BUF + (op * 2 + h) * G8_HALF + wave * 2048;
        if (op) {
            const int so = kt < NK ? kt * 128 + h * hstepB : (int)bbytes;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lptr_t)dst, 16, voB, so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lptr_t)(dst + 1024 - 64), 16, voB, so, 64, 0);
        } else if (kt < NK1) {
            const int so = kt * 128 + h * hstepA;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)dst, 16, voA, so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(dst + 1024 - 64), 16, voA, so, 64, 0);
        } else {
            const int so = kt < NK ? (kt - NK1) * 128 + h * hstepA2 : (int)0x7ffffff0;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra2, (lptr_t)dst, 16, voA2, so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra2, (lptr_t)(dst + 1024 - 64), 16, voA2, so, 64, 0);
        }
    };

    // ---- fragment reads: lane = (row & 15, 16-byte chunk lane >> 4) of a sub-tile
    unsigned fA[2], fB[2];
    {
        const int row = lane & 15, ch = lane >> 4;
        const unsigned off = row * 64 + (((ABL & 64) ? ch : (ch ^ ((row >> 3) << 1))) << 4);
        fA[0] = lds_base + wr * 8192 + off;
        fB[0] = lds_base + 2 * G8_HALF + wc * 4096 + off;
        fA[1] = fA[0] + G8_BUF;
        fB[1] = fB[0] + G8_BUF;
    }
    u32x4 af[4][2], bfr[2][2][2];   // [m tile][k half], [n half][n tile][k half]
    f32x4_t acc[2][2][4][2];        // [m half][n half][m tile][n tile]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int d = 0; d < 2; ++d) acc[a][b][c][d] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    auto readA = [&](int buf, int mh) {
        if (ABL & 2) return;
        const unsigned a = fA[buf] + mh * G8_HALF;
        af[0][0] = lds_read_b128_o<0>(a);    af[0][1] = lds_read_b128_o<1024>(a);
        af[1][0] = lds_read_b128_o<2048>(a); af[1][1] = lds_read_b128_o<3072>(a);
        af[2][0] = lds_read_b128_o<4096>(a); af[2][1] = lds_read_b128_o<5120>(a);
        af[3][0] = lds_read_b128_o<6144>(a); af[3][1] = lds_read_b128_o<7168>(a);
    };
    auto readB = [&](int buf, int nh) {
        if (ABL & 2) return;
        const unsigned a = fB[buf] + nh * G8_HALF;
        bfr[nh][0][0] = lds_read_b128_o<0>(a);    bfr[nh][0][1] = lds_read_b128_o<1024>(a);
        bfr[nh][1][0] = lds_read_b128_o<2048>(a); bfr[nh][1][1] = lds_read_b128_o<3072>(a);
    };
    // 16 MFMAs: quadrant (mh, nh).  Operand order (weights, pixels): a result tile is D[n][m], a lane holds 4 consecutive
    // output channels of one pixel
    auto quad = [&](int mh, int nh) {
        if (!(ABL & 16)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[mh][nh][mt][nt] = mfma16(bfr[nh][nt][kh], af[mt][kh], acc[mh][nh][mt][nt]);
        if (!(ABL & 16)) __builtin_amdgcn_s_setprio(0);
    };
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto lgkm0 = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    constexpr bool dma = !(ABL & 1);

    // ---- prologue: K-tile 0 complete, K-tile 1 without its A half 1 (phase 1 stages it)
    stage(0, 1, 0); stage(0, 0, 0); stage(0, 1, 1); stage(0, 0, 1);
    stage(1, 1, 0); stage(1, 0, 0); stage(1, 1, 1);
    if (ABL & 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                af[i][k] = u32x4{0x3f803f80u + lane, 0x3f003f80u, 0xbf803f80u, 0x3f80bf80u + i};
                bfr[i >> 1][i & 1][k] = u32x4{0x3f803f00u, 0x3f803f80u + lane * 3, 0x3e803f80u + k, 0x3f803f80u};
            }
    }
    wait_vmcnt<6>();
    bar();
    if (!(ABL & 32) && wr == 1) bar();

    for (int kt = 0; kt < NK; kt += 2) {
        // ---------------- K-tile kt (even buffer)
        readB(0, 0); __builtin_amdgcn_sched_barrier(0); readA(0, 0);
        if (dma) stage(kt + 1, 0, 1);
        if (!(ABL & 2)) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        bar(); lgkm0(); quad(0, 0); bar();

        readB(0, 1);
        if (dma) stage(kt + 2, 1, 0);
        bar(); lgkm0(); quad(0, 1); bar();

        readA(0, 1);
        if (dma) stage(kt + 2, 0, 0);
        bar(); lgkm0(); quad(1, 1); bar();

        if (dma) { stage(kt + 2, 1, 1); wait_vmcnt<6>(); }
        bar(); quad(1, 0); bar();

        // ---------------- K-tile kt + 1 (odd buffer)
        readB(1, 0); __builtin_amdgcn_sched_barrier(0); readA(1, 0);
        if (dma) stage(kt + 2, 0, 1);
        if (!(ABL & 2)) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        bar(); lgkm0(); quad(0, 0); bar();

        readB(1, 1);
        if (dma) stage(kt + 3, 1, 0);
        bar(); lgkm0(); quad(0, 1); bar();

        readA(1, 1);
        if (dma) stage(kt + 3, 0, 0);
        bar(); lgkm0(); quad(1, 1); bar();

        if (dma) { stage(kt + 3, 1, 1); wait_vmcnt<6>(); }
        bar(); quad(1, 0); bar();
    }
    if (!(ABL & 32) && wr == 0) bar();
    wait_vmcnt<0>();                          // the phantom half-tiles behind the last K-tile are out of the queue
    if (p.ts) t1 = __builtin_amdgcn_s_memtime();

    // ---- epilogue: + bias (+ residual), ReLU, bf16; lane = pixel (lane & 15) x 4 consecutive channels 4 (lane >> 4) + r of a tile
    if (ABL & 8) {
        float s = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int d = 0; d < 2; ++d) s += acc[a][b][c][d][0] + acc[a][b][c][d][1] + acc[a][b][c][d][2] + acc[a][b][c][d][3];
        if (s == 1.2345e33f) p.C[lane] = 1;
    } else {
        // a lane holds 16 consecutive channels (16 cq + 8 nh + 4 nt + r) of pixel px in its four tiles (nh, nt): two 16-byte stores per
        // pixel tile, the four lanes of a pixel cover a whole 128-byte line
        const int cq = lane >> 4, px = lane & 15;
        const int n = n0 + wc * 64 + 16 * cq;
        float4 bv[2][2];
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) bv[nh][nt] = *reinterpret_cast<const float4*>(p.bias + n + 8 * nh + 4 * nt);
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int m = m0 + wr * 128 + mh * 64 + mt * 16 + px;
                if (m >= p.M) continue;
#pragma unroll
                for (int nh = 0; nh < 2; ++nh) {
                    float v[8];
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const f32x4_t a = acc[mh][nh][mt][nt];
                        v[4 * nt] = a[0] + bv[nh][nt].x; v[4 * nt + 1] = a[1] + bv[nh][nt].y; v[4 * nt + 2] = a[2] + bv[nh][nt].z; v[4 * nt + 3] = a[3] + bv[nh][nt].w;
                    }
                    if (RES) {
                        const u32x4 r = *reinterpret_cast<const u32x4*>(p.R + (size_t)m * p.ldc + n + 8 * nh);
#pragma unroll
                        for (int j = 0; j < 4; ++j) { v[2 * j] += __uint_as_float(r[j] << 16); v[2 * j + 1] += __uint_as_float(r[j] & 0xffff0000u); }
                    }
                    const u32x4 o = {act2_bf16(v[0], v[1], p.relu != 0), act2_bf16(v[2], v[3], p.relu != 0), act2_bf16(v[4], v[5], p.relu != 0), act2_bf16(v[6], v[7], p.relu != 0)};
                    if (!(ABL & 4)) *reinterpret_cast<u32x4*>(p.C + (size_t)m * p.ldc + n + 8 * nh) = o;
                    else if (o[0] == 0x12345678u && o[1] == 0x9abcdef0u) *reinterpret_cast<u32x4*>(p.C + (size_t)m * p.ldc + n + 8 * nh) = o;
                }
            }
    }
    if (p.ts && tid == 0) {
        unsigned long long* t = p.ts + (size_t)blockIdx.x * 4;
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        t[0] = t0; t[1] = t1; t[2] = t2; t[3] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

}  // namespace ivosw
