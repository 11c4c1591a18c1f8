// The big-register-tile contraction of round 5: C[m][n] = act(sum_k A[m][k] * B[n][k] + bias[n]), bf16 in, fp32 accumulate.
//
// ONE wave per SIMD: a 256-thread workgroup owns 256 rows (pixels) x 256 columns (output channels); wave (wm, wn) of the
// 2 x 2 arrangement owns 128 x 128 = a 4 x 4 block of 32x32 MFMA tiles, i.e. 256 accumulator registers, which the 512-entry
// unified file of a SIMD holds only at one wave per SIMD (they live in the AGPR half; __launch_bounds__(256) +
// amdgpu_waves_per_eu(1, 1) is what makes hipcc select the AGPR forms).  BOTH operands come from LDS: per 32-deep K-tile a
// 16-KB row image of A and one of B ([256 rows][64 B], 16-B chunks XOR-swizzled by (row >> 2) & 3: conflict-free for the
// ds_read_b128 lane groups), filled by LDS-DMA (buffer_load ... lds, 8 1-KiB pieces per wave and K-tile) into a FOUR-stage
// ring, and a k-step of 16 MFMAs takes 8 ds_read_b128 (0.5 fragment reads per MFMA against 1.0 + a per-wave weight stream
// through the texture cache in the 8-wave kernels of rounds 1 - 4).
//
// Schedule (pinned with sched_barrier: a wave alone on its SIMD hides ~5 issues per MFMA, and only between the MFMAs): a
// K-tile is two k-steps of 16 MFMAs; behind MFMA i < 8 of a k-step sits one fragment read of the NEXT k-step, behind every
// fourth MFMA one DMA piece - the texture addresser moves 64 B per clock, so the 64 KB per 64 MFMAs of a workgroup must
// leave the waves evenly (the first version issued a K-tile's 16 pieces per wave in one k-step: that k-step took the 1 k
// cycles the addresser needs and the loop ran at 40 - 47 cycles per MFMA instead of 33).  ONE barrier per K-tile, between its
// two k-steps: every wave then has the tile's last fragments in registers, so the barrier publishes tile kt + 1 (each wave
// waited for its own pieces: counted vmcnt, two younger tiles stay in flight) and frees tile kt's stage for tile kt + 4.
#pragma once
#include <type_traits>

#include "mfma_tile.h"

namespace ivosw {

struct BtArgs {
    const bf16_t* A;      // [M][K] K-major
    const bf16_t* B;      // [N][K] K-major
    const float* bias;    // [N]
    bf16_t* C;            // [M][N]
    int M, N, K;          // M % 256 == 0, N % 256 == 0, K % 32 == 0
    int relu;
    unsigned long long* ts;   // optional [workgroups][4] stamps: s_memtime start / after K loop / end, s_memrealtime span
};

constexpr int BT_ROWB = 64;        // bytes per image row per K-tile (32 bf16)
constexpr int BT_OPB = 256 * BT_ROWB;   // one operand image: 16 KB
constexpr int BT_STAGE = 2 * BT_OPB;    // A | B
constexpr int BT_NS = 4;
constexpr int BT_LDS = BT_NS * BT_STAGE;

typedef __attribute__((address_space(3))) void* bt_lptr_t;

// ABL bits (micro-benchmark only): 1 = no LDS-DMA in the loop (the ring keeps its first tiles), 2 = no fragment reads in the
// loop, 4 = no stores, 8 = no epilogue at all (one value per lane keeps the accumulators alive), 16 = the four waves issue
// their DMA pieces behind different MFMAs of a group of four, 32 = every piece of the loop re-reads the tile's first KiB (issue
// cost without the bandwidth; wrong results)
template <int ABL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_bt_kernel(BtArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[BT_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane & 31, lhalf = lane >> 5;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int nbn = p.N / 256;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = L % nbn, tile_m = L / nbn;
    const int m0 = tile_m * 256, n0 = tile_n * 256;
    const int NK = p.K / 32;
    unsigned long long t0 = 0, t1 = 0, r0 = 0;
    if (p.ts) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }

    // ---- LDS-DMA sources: wave w fills rows [64 w, 64 w + 64) of both images, 4 pieces of 16 rows each
    // descriptors of exactly this tile's rows: a piece of a K-tile that does not exist (kt >= NK: the ring runs three and a half
    // tiles ahead, and the loop body is the same for every tile) is issued with soffset = num_records, i.e. out of range - no
    // memory access, zeros into a stage nobody reads - so that the counted waits never change; rows beyond M read zeros as well
    const int abytes = min(256, p.M - m0) * p.K * 2, bbytes = 256 * p.K * 2;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A) + (size_t)m0 * p.K, 0, abytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.B) + (size_t)n0 * p.K, 0, bbytes, 0x00020000);
    int vo[4];
    {
        const int rsub = lane >> 2, cpos = lane & 3;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int row = wave * 64 + g * 16 + rsub;
            vo[g] = row * p.K * 2 + ((cpos ^ ((row >> 2) & 3)) << 4);
        }
    }
    // piece q of K-tile kt: q >> 1 = which 16-row group of the wave's 64 rows, q & 1 = operand
    auto issue_piece = [&](int kt, int q) {
        unsigned char* st = lds + (kt & (BT_NS - 1)) * BT_STAGE + wave * 4096 + (q >> 1) * 1024;
        if (q & 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (bt_lptr_t)(st + BT_OPB), 16, vo[q >> 1], kt < NK ? kt * 64 : bbytes, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (bt_lptr_t)st, 16, vo[q >> 1], kt < NK ? kt * 64 : abytes, 0, 0);
    };
    auto issue_loop = [&](int kt, int q) {
        if (ABL & 32) {
            unsigned char* st = lds + (kt & (BT_NS - 1)) * BT_STAGE + wave * 4096 + (q >> 1) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (bt_lptr_t)(st + (q & 1) * BT_OPB), 16, lane * 16, 0, 0, 0);
        } else issue_piece(kt, q);
    };
    // which MFMA of a group of four carries the group's DMA piece
    const int slot = (ABL & 16) ? wave : 3;

    // the bias is requested first, the ring's prologue goes out behind it, and only then the accumulators are initialised (the
    // compiler's own counted wait for the bias sits there): one memory round trip in front of the first MFMA instead of two
    f32x16 acc[4][4];   // [channel tile j][pixel tile i]
    float4 bq[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) bq[j][g] = *reinterpret_cast<const float4*>(p.bias + n0 + wn * 128 + j * 32 + 8 * g + 4 * lhalf);

    // fragment addresses: row (lane & 31) of the wave's first tile, chunk (2 ks + lane half) ^ key; the other tiles are
    // immediates (32 rows = 2048 B apart, same swizzle key)
    unsigned fa[2], fb[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const unsigned x = ((2 * ks + lhalf) ^ ((lrow >> 2) & 3)) << 4;
        fa[ks] = lds_base + (wm * 128 + lrow) * BT_ROWB + x;
        fb[ks] = lds_base + BT_OPB + (wn * 128 + lrow) * BT_ROWB + x;
    }
    u32x4 pa[2][4], pb[2][4];
    // fragment read r (0..3: pixel tiles, 4..7: channel tiles) of k-step ks into register set `set`
    auto rd1 = [&](int set, int ks, unsigned so, int r) {
        if (ABL & 2) return;
        switch (r) {
            case 0: pa[set][0] = lds_read_b128_o<0>(fa[ks] + so); break;
            case 1: pa[set][1] = lds_read_b128_o<2048>(fa[ks] + so); break;
            case 2: pa[set][2] = lds_read_b128_o<4096>(fa[ks] + so); break;
            case 3: pa[set][3] = lds_read_b128_o<6144>(fa[ks] + so); break;
            case 4: pb[set][0] = lds_read_b128_o<0>(fb[ks] + so); break;
            case 5: pb[set][1] = lds_read_b128_o<2048>(fb[ks] + so); break;
            case 6: pb[set][2] = lds_read_b128_o<4096>(fb[ks] + so); break;
            default: pb[set][3] = lds_read_b128_o<6144>(fb[ks] + so); break;
        }
    };
    // one k-step: 16 MFMAs on register set `set`, with one filler slot behind each
    auto kstep = [&](int set, auto&& fill) {
#pragma unroll
        for (int idx = 0; idx < 16; ++idx) {
            const int j = idx >> 2, i = idx & 3;
            acc[j][i] = mfma_bf16(pb[set][j], pa[set][i], acc[j][i]);
            fill(idx);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // prologue: tiles 0 .. 2 and the first half of tile 3; then tile 0 is waited for and published
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int q = 0; q < 8; ++q) issue_piece(t, q);
#pragma unroll
    for (int q = 0; q < 4; ++q) issue_piece(3, q);
    // accumulators start at the bias: row r of a 32x32 result tile is channel (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                acc[j][i][4 * g] = bq[j][g].x; acc[j][i][4 * g + 1] = bq[j][g].y; acc[j][i][4 * g + 2] = bq[j][g].z; acc[j][i][4 * g + 3] = bq[j][g].w;
            }
    wait_vmcnt<20>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int r = 0; r < 8; ++r) rd1(0, 0, 0, r);
    if (ABL & 2) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                pa[s][i] = u32x4{0x3f803f80u + lane, 0x3f003f80u, 0xbf803f80u, 0x3f80bf80u + i};
                pb[s][i] = u32x4{0x3f803f00u, 0x3f803f80u + lane * 3, 0x3e803f80u + s, 0x3f803f80u};
            }
    }
    for (int kt = 0; kt < NK; ++kt) {
        constexpr bool dma = !(ABL & 1);
        const unsigned so = (kt & (BT_NS - 1)) * BT_STAGE, sn = ((kt + 1) & (BT_NS - 1)) * BT_STAGE;
        lds_wait();
        kstep(0, [&](int idx) {
            if (idx < 8) rd1(1, 1, so, idx);
            if (dma && (idx & 3) == slot) issue_loop(kt + 3, 4 + (idx >> 2));    // second half of tile kt + 3
        });
        lds_wait();
        wait_vmcnt<dma ? 16 : 0>();           // tile kt + 1 has landed; kt + 2, kt + 3 may be in flight
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        kstep(1, [&](int idx) {
            if (idx < 8) rd1(0, 0, sn, idx);
            if (dma && (idx & 3) == slot) issue_loop(kt + 4, idx >> 2);         // first half of tile kt + 4
        });
    }
    lds_wait();
    wait_vmcnt<0>();                          // the phantom pieces behind the last tile are out of the queue
    if (p.ts) t1 = __builtin_amdgcn_s_memtime();

    // ---- epilogue: ReLU, bf16; lanes l and l + 32 hold the two halves of 8 consecutive channels of one pixel, a
    // v_permlane32_swap per dword turns two 8-byte pieces into one 16-byte store per lane
    if (ABL & 8) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[j][i][r];
        if (s == 1.2345e33f) p.C[lane] = 1;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nb = n0 + wn * 128 + j * 32;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + wm * 128 + i * 32 + lrow;
                bf16_t* o = p.C + (size_t)m * p.N + nb + 8 * lhalf;
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    const int g = 2 * gp;
                    const f32x16& a = acc[j][i];
                    unsigned x0 = act2_bf16(a[4 * g], a[4 * g + 1], p.relu != 0);
                    unsigned x1 = act2_bf16(a[4 * g + 2], a[4 * g + 3], p.relu != 0);
                    unsigned y0 = act2_bf16(a[4 * g + 4], a[4 * g + 5], p.relu != 0);
                    unsigned y1 = act2_bf16(a[4 * g + 6], a[4 * g + 7], p.relu != 0);
                    // lower lanes: [own g | upper's g] = channels 8g .. 8g+7; upper lanes: [lower's g+1 | own g+1] = 8(g+1) ..
                    auto s0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
                    auto s1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
                    const u32x4 v = {s0[0], s1[0], s0[1], s1[1]};
                    if (!(ABL & 4)) *reinterpret_cast<u32x4*>(o + 16 * gp) = v;
                    else if (v[0] == 0x12345678u && v[3] == 0x9abcdef0u) *reinterpret_cast<u32x4*>(o + 16 * gp) = v;
                }
            }
        }
    }
    if (p.ts && tid == 0) {
        unsigned long long* t = p.ts + (size_t)blockIdx.x * 4;
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        t[0] = t0; t[1] = t1; t[2] = t2; t[3] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}


// ---------------------------------------------------------------- the same contraction, PERSISTENT (round 5, experiment)
// One workgroup per CU walks over the tiles blockIdx.x, + gridDim.x, ...  and the LDS-DMA ring does NOT stop at a tile boundary: the
// pieces the one-tile kernel issues out of range behind its last K-tile (the ring runs 3.5 K-tiles ahead) are the first K-tiles of the
// workgroup's NEXT tile here, so a tile starts with its operands on chip instead of with an HBM / L2 round trip.  Why: with MFMA work
// only, a one-tile workgroup at K = 768 spends 42 cycles per MFMA against 33.4 at K = 8 192 - ~ 6.5 k cycles of start-up per tile, four
// tiles per CU and launch.  Counted waits stay valid across the epilogue: its stores and the next tile's bias loads only ADD to
// vmcnt (loads retire in order among loads), so `vmcnt(16)` can only release later, never earlier.  K >= 128 (NK >= 4: the run-ahead must
// stay inside the next tile).
// MEASURED SLOWER (profiles/r05_gemm_tile_bench.txt, last section): 130 - 135 us against 114 - 116 for 65 536 x 1 024 x 768, 117 - 121 against 96 - 100
// for 16 384 x 2 048 x 1 536 - the epilogue's 32 stores per wave sit in front of the next tile's counted waits (vmcnt retires them before it
// releases), which costs what the saved start-up bought and more.  A template so that only the micro-benchmark instantiates it.
template <int UNUSED = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_bt_persist_kernel(BtArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[BT_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane & 31, lhalf = lane >> 5;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int nbn = p.N / 256, ntiles = (p.M / 256) * nbn, G = gridDim.x;
    const int NK = p.K / 32;
    unsigned long long t0 = 0, t1 = 0, r0 = 0;
    if (p.ts) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }

    int vo[4];
    {
        const int rsub = lane >> 2, cpos = lane & 3;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int row = wave * 64 + g * 16 + rsub;
            vo[g] = row * p.K * 2 + ((cpos ^ ((row >> 2) & 3)) << 4);
        }
    }
    const int tbytes = 256 * p.K * 2;
    // descriptors of a tile's A rows / B rows; a tile that does not exist gets empty descriptors: every piece of it is out of range
    auto srd_a = [&](int t) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A) + (size_t)(t < ntiles ? (t / nbn) * 256 : 0) * p.K, 0, t < ntiles ? tbytes : 0, 0x00020000);
    };
    auto srd_b = [&](int t) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.B) + (size_t)(t < ntiles ? (t % nbn) * 256 : 0) * p.K, 0, t < ntiles ? tbytes : 0, 0x00020000);
    };
    int t = xcd_remap(blockIdx.x, G);
    __amdgpu_buffer_rsrc_t ra = srd_a(t), rb = srd_b(t), ra2 = srd_a(t + G), rb2 = srd_b(t + G);
    int base = 0;                                    // ring position (in K-tiles) of the current tile's K-tile 0
    auto issue_piece = [&](int kt, int q) {          // kt >= NK: K-tile kt - NK of the next tile
        unsigned char* st = lds + ((base + kt) & (BT_NS - 1)) * BT_STAGE + wave * 4096 + (q >> 1) * 1024;
        const bool nx = kt >= NK;
        const int so = (nx ? kt - NK : kt) * 64;
        if (q & 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(nx ? rb2 : rb, (bt_lptr_t)(st + BT_OPB), 16, vo[q >> 1], so, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(nx ? ra2 : ra, (bt_lptr_t)st, 16, vo[q >> 1], so, 0, 0);
    };
    unsigned fa[2], fb[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const unsigned x = ((2 * ks + lhalf) ^ ((lrow >> 2) & 3)) << 4;
        fa[ks] = lds_base + (wm * 128 + lrow) * BT_ROWB + x;
        fb[ks] = lds_base + BT_OPB + (wn * 128 + lrow) * BT_ROWB + x;
    }
    u32x4 pa[2][4], pb[2][4];
    auto rd1 = [&](int set, int ks, unsigned so, int r) {
        switch (r) {
            case 0: pa[set][0] = lds_read_b128_o<0>(fa[ks] + so); break;
            case 1: pa[set][1] = lds_read_b128_o<2048>(fa[ks] + so); break;
            case 2: pa[set][2] = lds_read_b128_o<4096>(fa[ks] + so); break;
            case 3: pa[set][3] = lds_read_b128_o<6144>(fa[ks] + so); break;
            case 4: pb[set][0] = lds_read_b128_o<0>(fb[ks] + so); break;
            case 5: pb[set][1] = lds_read_b128_o<2048>(fb[ks] + so); break;
            case 6: pb[set][2] = lds_read_b128_o<4096>(fb[ks] + so); break;
            default: pb[set][3] = lds_read_b128_o<6144>(fb[ks] + so); break;
        }
    };
    f32x16 acc[4][4];
    auto kstep = [&](int set, auto&& fill) {
#pragma unroll
        for (int idx = 0; idx < 16; ++idx) {
            const int j = idx >> 2, i = idx & 3;
            acc[j][i] = mfma_bf16(pb[set][j], pa[set][i], acc[j][i]);
            fill(idx);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    float4 bq[4][4];
    auto load_bias = [&](int tt) {
        const int n0 = (tt < ntiles ? tt % nbn : 0) * 256;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) bq[j][g] = *reinterpret_cast<const float4*>(p.bias + n0 + wn * 128 + j * 32 + 8 * g + 4 * lhalf);
    };
    load_bias(t);
    // prologue of the FIRST tile only
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int q = 0; q < 8; ++q) issue_piece(k, q);
#pragma unroll
    for (int q = 0; q < 4; ++q) issue_piece(3, q);
    wait_vmcnt<20>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int r = 0; r < 8; ++r) rd1(0, 0, 0, r);

    for (; t < ntiles; t += G) {
        const int m0 = (t / nbn) * 256, n0 = (t % nbn) * 256;
        // (zero start + bias in the epilogue: initialising the AGPR tiles from the bias registers inside this loop made hipcc route all 256
        // values through scratch)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
        float4 bc[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) bc[j][g] = bq[j][g];
        for (int kt = 0; kt < NK; ++kt) {
            const unsigned so = ((base + kt) & (BT_NS - 1)) * BT_STAGE, sn = ((base + kt + 1) & (BT_NS - 1)) * BT_STAGE;
            lds_wait();
            kstep(0, [&](int idx) {
                if (idx < 8) rd1(1, 1, so, idx);
                if ((idx & 3) == 3) issue_piece(kt + 3, 4 + (idx >> 2));
            });
            lds_wait();
            wait_vmcnt<16>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            kstep(1, [&](int idx) {
                if (idx < 8) rd1(0, 0, sn, idx);         // (the last K-tile reads K-tile 0 of the NEXT tile: already published)
                if ((idx & 3) == 3) issue_piece(kt + 4, idx >> 2);
            });
        }
        if (t + G < ntiles) load_bias(t + G);            // in flight under the epilogue
        // epilogue of this tile (the next tile's first fragments sit in register set 0 meanwhile)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nb = n0 + wn * 128 + j * 32;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + wm * 128 + i * 32 + lrow;
                bf16_t* o = p.C + (size_t)m * p.N + nb + 8 * lhalf;
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    const int g = 2 * gp;
                    const f32x16& a = acc[j][i];
                    unsigned x0 = act2_bf16(a[4 * g] + bc[j][g].x, a[4 * g + 1] + bc[j][g].y, p.relu != 0);
                    unsigned x1 = act2_bf16(a[4 * g + 2] + bc[j][g].z, a[4 * g + 3] + bc[j][g].w, p.relu != 0);
                    unsigned y0 = act2_bf16(a[4 * g + 4] + bc[j][g + 1].x, a[4 * g + 5] + bc[j][g + 1].y, p.relu != 0);
                    unsigned y1 = act2_bf16(a[4 * g + 6] + bc[j][g + 1].z, a[4 * g + 7] + bc[j][g + 1].w, p.relu != 0);
                    auto s0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
                    auto s1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
                    const u32x4 v = {s0[0], s1[0], s0[1], s1[1]};
                    *reinterpret_cast<u32x4*>(o + 16 * gp) = v;
                }
            }
        }
        base += NK;
        ra = ra2; rb = rb2;
        ra2 = srd_a(t + 2 * G); rb2 = srd_b(t + 2 * G);
    }
    lds_wait();
    wait_vmcnt<0>();
    if (p.ts && tid == 0) {
        unsigned long long* tsp = p.ts + (size_t)blockIdx.x * 4;
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        tsp[0] = t0; tsp[1] = t2; tsp[2] = t2; tsp[3] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    (void)t1;
}

}  // namespace ivosw
