// Whole-bottleneck fusion for the HBM-bound stages of the assessment tower (bf16 throughput mode).
//
// Reference arithmetic: one torchvision ResNet-50 v1.5 Bottleneck as Encoder.forward runs it
// (models/assessment.py:58-61):  out = relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1(x)))))))) + x)
// with the BN scale folded into the packed weights and the BN shift as a bias (see pack_conv_kernel).
//
// Why: layer by layer, res2/res3 move 16 "C-channel planes" per block through HBM (x in, t1 out/in, t2 out/in,
// residual in, y out) and sit at 3-4 TB/s with the matrix cores idle.  Fused, one workgroup owns a 16x16-pixel
// tile of one frame and only x (with a 1-pixel halo) comes in and y goes out: 8 planes, half the bytes, and the
// intermediates never leave the CU:
//
//   phase A  t1 = relu(Wa * x + ba) on the 18x18 halo (zero outside the image).  K = CIN streams HBM/L2 -> LDS by
//            LDS-DMA in 64-channel K-tiles through a 3-stage ring that borrows t1's own space, so three of the
//            four K-tiles are in flight from the first cycle; result kept in LDS as bf16 [324 px][64 ch].
//   phase B  t2 = relu(Wb (*) t1 + bb): 9 taps, B operand = shifted windows of t1, all nine 8-KB weight taps
//            resident (fetched under phase A).  Each wave owns a 2x2 block of 32x32 tiles on HALF of every tap's
//            K (4 MFMAs per fragment fetch, no barrier inside the 18-step loop); the two halves meet through LDS.
//   phase C  y  = relu(Wc * t2 + bc + x) in two channel halves: accumulators (initialised with the bias) ->
//            per-wave LDS staging -> full 128-B line stores; the residual tile is fetched into registers before
//            phase B so its HBM latency hides under the taps.
//
// All three contractions run "transposed" (MFMA A operand = weights, B operand = pixels) so that a lane ends up
// holding 4 consecutive channels of ONE pixel per accumulator quad: t1/t2 are written with ds_write_b64 and the
// final tile with ds_write_b128, instead of 2-byte scatters.
//
// LDS map (163 840 B, one workgroup of 8 waves per CU); a ring stage is [Wa K-tile 8 KB][A: 41 x 1 KB row groups]:
//   S0 [0, 50176)        K-tiles 0,3 -> t1 [0, 41472) -> exchange scratch / store staging of waves 0..5
//   S1 [50176, 100352)   K-tile 1    -> Wb taps 0..5 -> t2 [50176, 82944) | scratch/staging of waves 6,7
//   S2 [100352, 150528)  K-tile 2    -> Wc [100352, 133120) | Wb taps 6,7 | tap 8 runs to 157696
//   [157696, 158720) DMA dummy, [160768, 163840) the bias vectors
// Downsample block (DS: first block of res2, Cin 64, y = relu(Wc t2 + bc + Wd x + bd)): one K-tile in phase A, every
// weight tap in flight from the first cycle, and phase C gets a second K-tile: the x centre rows are re-fetched
// (L2-hot) next to Wd into the space t1 and the scratch vacated.
#ifdef IVOSW_PROBES
#include "../../include/ivosw_probe.h"
#endif
#include <stdlib.h>

#include "conv.h"
#include "mfma_tile.h"

namespace ivosw {

namespace {

constexpr int BT = 16;                  // output tile edge (pixels)
constexpr int HT = BT + 2;              // halo tile edge
constexpr int HR = HT * HT;             // 324 halo pixels
constexpr int NGA = 41;                 // 1-KB DMA row groups per A K-tile (328 rows)
constexpr int WK_BYTES = 64 * ROWB;     // one 64-row weight K-tile: 8 KB
constexpr int STG = WK_BYTES + NGA * 1024;      // 50176
constexpr int S0 = 0, S1 = STG, S2 = 2 * STG;
constexpr int T1_OFF = S0;
constexpr int T2_OFF = S1;
constexpr int WC_OFF = S2;
constexpr int WB_HI = WC_OFF + 256 * ROWB;      // 133120: taps 6..8
constexpr int DUMMY_OFF = WB_HI + 3 * WK_BYTES; // 157696
constexpr int BIAS_OFF = 160768;                // ba[64] | bb[64] | bc[256] | bd[256] fp32 | 512 B pad (3 x 1-KB DMA)
constexpr int FUSED_LDS = 163840;
static_assert(S1 + 6 * WK_BYTES <= S2 && DUMMY_OFF + 1024 <= BIAS_OFF && BIAS_OFF + 3072 <= FUSED_LDS, "LDS map");
// downsample variant, phase C: x centre rows (256 x 128 B) and the two 128-row halves of Wd
constexpr int XC_OFF = 0, WD_OFF0 = 32768, WD_OFF1 = WB_HI;
static_assert(S2 + WK_BYTES + 352 * ROWB <= FUSED_LDS, "fragment reads of the padded pixel tile stay inside LDS");

__host__ __device__ constexpr int wb_slot(int t) { return t < 6 ? S1 + t * WK_BYTES : WB_HI + (t - 6) * WK_BYTES; }
__device__ __forceinline__ int wave_scratch(int wave) { return wave < 6 ? wave * 8192 : T2_OFF + 256 * ROWB + (wave - 6) * 8192; }

// ablation: MFMA off (operands kept live so nothing upstream is dead code)
__device__ __forceinline__ f32x16 mfma_dbg(u32x4 a, u32x4 b, f32x16 c, int dbg) {
    if (ABL(dbg, 8)) {
        asm volatile("" ::"v"(a), "v"(b));
        return c;
    }
    return mfma_bf16(a, b, c);
}

__device__ __forceinline__ void lds_write_b128(unsigned addr, u32x4 v) {
    asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

}  // namespace

// bottleneck of mid width 64: x [B,H,W,CIN] -> y [B,H,W,256].  DS = false: identity block (CIN == 256, residual = x);
// DS = true: downsample block (CIN == 64, residual = Wd x + bd, stride 1)
template <int CIN, bool DS>
__global__ __launch_bounds__(512) void bneck64_kernel(BneckArgs p) {
    static_assert((CIN == 256 && !DS) || (CIN == 64 && DS), "res2: identity blocks (4 K-tiles over a 3-stage ring) or the first block (1 K-tile)");
    constexpr int NKA = CIN / 64;
    __shared__ __attribute__((aligned(16))) unsigned char lds[FUSED_LDS];  // the ONLY LDS object

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int tiles_x = p.W / BT, tiles = tiles_x * (p.H / BT);
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int b = L / tiles, tl = L - b * tiles;
    const int y0 = (tl / tiles_x) * BT, x0 = (tl % tiles_x) * BT;

    const bf16_t* X = static_cast<const bf16_t*>(p.x);
    const bf16_t* zeros = static_cast<const bf16_t*>(p.zeros);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;

    auto stamp = [&](int k) {
        if (p.ts && tid == 0) p.ts[(size_t)blockIdx.x * 16 + k] = __builtin_amdgcn_s_memtime();
    };
    // Stagger: the first wave of workgroups (one per CU) would march through load-bound phase A and the compute-bound
    // phases B/C in lockstep, leaving HBM idle half of the time; delaying every other one by half a tile period puts
    // neighbouring CUs in opposite phases, and the equal tile times keep them there.
    if (p.stagger > 0 && blockIdx.x < 256 && ((blockIdx.x >> 3) & 1)) {
        for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    }
    stamp(0);

    // ---------------------------------------------------------------- DMA duties of this lane
    // A K-tile: row groups wave, wave+8, .., wave+32 (40 groups) + group 40 by wave 0; the other waves' sixth DMA
    // copies 1 KB of zeros to a dummy page so that every wave issues the same number (the vmcnt counts are literals)
    const int rsub = lane >> 3, cpos = lane & 7;
    const bf16_t* abase[6];
    unsigned okmask = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int g = wave + 8 * i;
        const int hr = g * 8 + rsub;
        const int hy = hr / HT, hx = hr - hy * HT;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        const bool ok = g < NGA && hr < HR && y >= 0 && y < p.H && x >= 0 && x < p.W;
        abase[i] = ok ? X + (((long)b * p.H + y) * p.W + x) * CIN + (cpos ^ ((hr >> 1) & 7)) * 8 : zeros;
        okmask |= ok ? (1u << i) : 0u;
    }
    const int wrow = wave * 8 + rsub;               // weight row of a 64-row K-tile this lane fetches
    const int wchunk = (cpos ^ ((wrow >> 1) & 7)) * 8;
    const bf16_t* wa_src = static_cast<const bf16_t*>(p.wa) + (long)wrow * CIN + wchunk;
    const bf16_t* wb_src = static_cast<const bf16_t*>(p.wb) + (long)wrow * (9 * 64) + wchunk;

    auto issue_a = [&](int kt, int st) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int g = wave + 8 * i;
            const bf16_t* src = abase[i] + (((okmask >> i) & 1u) && !ABL(p.debug, 4) ? kt * 64 : 0);
            dma16(src, lds + (g < NGA ? st + WK_BYTES + g * 1024 : DUMMY_OFF));
        }
        dma16(wa_src + kt * 64, lds + st + wave * 1024);
    };
    auto issue_wb = [&](int tap) { dma16(wb_src + tap * 64, lds + wb_slot(tap) + wave * 1024); };
    auto issue_wc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int g = wave * 4 + i;
            const int row = g * 8 + rsub;
            dma16(static_cast<const bf16_t*>(p.wc) + (long)row * 64 + (cpos ^ ((row >> 1) & 7)) * 8, lds + WC_OFF + g * 1024);
        }
    };

    // biases -> LDS (oldest DMAs of wave 0: every later counted wait covers them)
    if (wave == 0) {
        const float* zf = reinterpret_cast<const float*>(zeros);
        const float* s0 = lane < 16 ? p.ba + lane * 4 : lane < 32 ? p.bb + (lane - 16) * 4 : p.bc + (lane - 32) * 4;
        const float* s1 = lane < 32 ? p.bc + 128 + lane * 4 : DS ? p.bd + (lane - 32) * 4 : zf;
        const float* s2 = (DS && lane < 32) ? p.bd + 128 + lane * 4 : zf;
        dma16(s0, lds + BIAS_OFF);
        dma16(s1, lds + BIAS_OFF + 1024);
        dma16(s2, lds + BIAS_OFF + 2048);
    }
    issue_a(0, S0);
    if constexpr (!DS) {
        issue_a(1, S1);
        issue_a(2, S2);
    } else {
#pragma unroll
        for (int t = 0; t < 6; ++t) issue_wb(t);
        issue_wc();
#pragma unroll
        for (int t = 6; t < 9; ++t) issue_wb(t);
    }

    // Store-pass geometry (phase C): lane = (pixel sub-row prr, 8-channel group u); pass (cp, q) covers pixel tile 2wm+q,
    // channels (4wn+2cp)*32 .. +64, i.e. exactly K-tile 2wn+cp of x: the residual is picked out of the ring stage of
    // that K-tile while it is resident (16 B x 16 per lane, bf16), so x is read from memory once.
    const int wm = wave >> 1, wn = wave & 1;
    const int u = lane & 7, prr = lane >> 3;
    char* Yb = static_cast<char*>(p.y);
    const unsigned rowb = (unsigned)p.W * 512u;
    const unsigned lbase = (unsigned)(((b * p.H + y0) * p.W + x0 + prr) * 512) + (unsigned)((4 * wn * 32 + 8 * u) * 2);
    auto goff = [&](int cp, int q, int it) {   // byte offsets fit 32 bits (bneck_fusable)
        return lbase + (unsigned)((2 * wm + q) * 2 + (it >> 1)) * rowb + (unsigned)((it & 1) * 8 * 512 + cp * 128);
    };
    u32x4 rr[4][4];
    unsigned rrow[2][4];                             // LDS byte offset (within a stage's A part) of this lane's residual chunks
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int px = (2 * wm + q) * 32 + it * 8 + prr;
            const int hr = ((px >> 4) + 1) * HT + (px & 15) + 1;
            rrow[q][it] = hr * ROWB + ((u ^ ((hr >> 1) & 7)) << 4);
        }

    // ================================================================ phase A: t1 = relu(Wa x + ba) on the halo
    {
        const int ct = wave & 1, pq = wave >> 1;    // channel tile; pixel tiles pq, pq+4, pq+8 (the last only if pq < 3)
        const bool has3 = pq < 3;
        f32x16 acc[3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKA; ++kt) {
            // DMA issue order per wave.  identity: T0 T1 T2 | T3 (after tile 0) | Wb0-5 (after tile 1) | Wc x4, Wb6-8
            // (after tile 2).  DS: T0, Wb0-5, Wc x4, Wb6-8 all up front.
            constexpr int younger_id[4] = {14, 14, 13, 13};
            constexpr int stage[4] = {S0, S1, S2, S0};
            const int st = stage[kt];
            wait_vmcnt_n(DS ? 13 : younger_id[kt]);
            __builtin_amdgcn_s_barrier();            // K-tile kt landed for every wave
            asm volatile("" ::: "memory");
            stamp(1 + kt);
            const unsigned w_base = lds_base + st, a_base = w_base + WK_BYTES;
            u32x4 wf[4], pf[4][3];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int ch = 2 * ks + lhalf;
                wf[ks] = lds_read_b128(w_base + swz(ct * 32 + lrow, ch));
                pf[ks][0] = lds_read_b128(a_base + swz(pq * 32 + lrow, ch));
                pf[ks][1] = lds_read_b128(a_base + swz((pq + 4) * 32 + lrow, ch));
                if (has3) pf[ks][2] = lds_read_b128(a_base + swz((pq + 8) * 32 + lrow, ch));
            }
            if constexpr (!DS) {
                if (wn == (kt >> 1)) {               // this K-tile holds the residual of my store passes (cp = kt & 1)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int it = 0; it < 4; ++it) rr[(kt & 1) * 2 + q][it] = lds_read_b128(a_base + rrow[q][it]);
                }
            }
            lds_wait();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                acc[0] = mfma_dbg(wf[ks], pf[ks][0], acc[0], p.debug);
                acc[1] = mfma_dbg(wf[ks], pf[ks][1], acc[1], p.debug);
                if (has3) acc[2] = mfma_dbg(wf[ks], pf[ks][2], acc[2], p.debug);
            }
            if (kt < 2) stamp(11 + 2 * kt);
            __builtin_amdgcn_s_barrier();            // every wave is done reading this stage
            asm volatile("" ::: "memory");
            if constexpr (!DS) {
                if (kt == 0) issue_a(3, S0);
                if (kt == 1) {
#pragma unroll
                    for (int t = 0; t < 6; ++t) issue_wb(t);
                }
                if (kt == 2) {
                    issue_wc();
#pragma unroll
                    for (int t = 6; t < 9; ++t) issue_wb(t);
                }
            }
            if (kt < 2) stamp(12 + 2 * kt);
        }
        stamp(5);
        // epilogue A: + bias, ReLU, zero outside the image (the 3x3 pads t1, not x), bf16 -> t1 (S0 is free)
        u32x4 bq[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bq[g] = lds_read_b128(lds_base + BIAS_OFF + (ct * 32 + 8 * g + 4 * lhalf) * 4);
        lds_wait();
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (i == 2 && !has3) break;
            const int hr = (pq + 4 * i) * 32 + lrow;
            const int hy = hr / HT, hx = hr - hy * HT;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const bool in = y >= 0 && y < p.H && x >= 0 && x < p.W;
            if (hr < HR) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fmaxf(acc[i][4 * g + j] + __uint_as_float(bq[g][j]), 0.f);
                    u32x2 pk;
                    pk.x = in ? pack2_bf16(v[0], v[1]) : 0u;
                    pk.y = in ? pack2_bf16(v[2], v[3]) : 0u;
                    lds_write_b64(lds_base + T1_OFF + hr * ROWB + (((ct * 4 + g) ^ ((hr >> 1) & 7)) << 4) + 8 * lhalf, pk);
                }
            }
        }
    }

    wait_vmcnt_n(0);                                 // every DMA has landed
    lds_wait();
    __builtin_amdgcn_s_barrier();                    // t1 complete, Wb / Wc / biases visible to all waves
    asm volatile("" ::: "memory");
    stamp(6);

    // ================================================================ phase B: t2 = relu(Wb (*) t1 + bb)
    {
        const int pp = wave & 3, kg = wave >> 2;     // pixel tiles 2pp, 2pp+1 x both channel tiles; K-steps 2kg, 2kg+1 of every tap
        f32x16 acc[2][2];                            // [channel tile][pixel tile]
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][q][r] = 0.f;
        int hr0[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int px = (2 * pp + q) * 32 + lrow;  // output pixel: row px>>4, col px&15
            hr0[q] = (px >> 4) * HT + (px & 15);
        }
        u32x4 wf[2][2], pf[2][2];
        auto frag_read = [&](int s, int buf) {       // step s: tap s>>1, K-step 2kg + (s&1)
            const int t = s >> 1;
            const int ch = 2 * (2 * kg + (s & 1)) + lhalf;
            const unsigned w_base = lds_base + wb_slot(t);
            wf[buf][0] = lds_read_b128(w_base + swz(lrow, ch));
            wf[buf][1] = lds_read_b128(w_base + swz(32 + lrow, ch));
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int hr = hr0[q] + (t / 3) * HT + (t % 3);
                pf[buf][q] = lds_read_b128(lds_base + T1_OFF + hr * ROWB + ((ch ^ ((hr >> 1) & 7)) << 4));
            }
        };
        frag_read(0, 0);
#pragma unroll
        for (int s = 0; s < 18; ++s) {
            lds_wait();
            if (s < 17) frag_read(s + 1, (s + 1) & 1);
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[c][q] = mfma_dbg(wf[s & 1][c], pf[s & 1][q], acc[c][q], p.debug);
        }
        stamp(7);
        __builtin_amdgcn_s_barrier();                // every wave is done with t1 and the taps: their space is scratch now
        asm volatile("" ::: "memory");
        // the K halves meet: wave (pp, kg) finishes channel tile kg and hands its partial of tile 1-kg to wave (pp, 1-kg)
        const unsigned mine = lds_base + wave_scratch(wave), theirs = lds_base + wave_scratch(wave ^ 4);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x16& a = kg ? acc[0][q] : acc[1][q];
                u32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = __float_as_uint(a[4 * g + j]);
                lds_write_b128(mine + ((q * 4 + g) * 64 + lane) * 16, v);
            }
        lds_wait();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        u32x4 part[2][4], bq[4];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g) part[q][g] = lds_read_b128(theirs + ((q * 4 + g) * 64 + lane) * 16);
#pragma unroll
        for (int g = 0; g < 4; ++g) bq[g] = lds_read_b128(lds_base + BIAS_OFF + 256 + (kg * 32 + 8 * g + 4 * lhalf) * 4);
        lds_wait();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int px = (2 * pp + q) * 32 + lrow;
            const f32x16& a = kg ? acc[1][q] : acc[0][q];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(a[4 * g + j] + __uint_as_float(part[q][g][j]) + __uint_as_float(bq[g][j]), 0.f);
                u32x2 pk;
                pk.x = pack2_bf16(v[0], v[1]);
                pk.y = pack2_bf16(v[2], v[3]);
                lds_write_b64(lds_base + T2_OFF + px * ROWB + (((kg * 4 + g) ^ ((px >> 1) & 7)) << 4) + 8 * lhalf, pk);
            }
        }
        lds_wait();
        __builtin_amdgcn_s_barrier();                // t2 complete
        asm volatile("" ::: "memory");
        stamp(8);
    }

    if constexpr (DS) {
        // ============================================================ phase C (DS): y = relu(Wc t2 + Wd x + bc + bd)
        // second K-tile: x centre rows (re-fetched, L2-hot) and Wd land where t1 / the scratch were (free since the
        // barrier above) while the first K-tile runs
        {
            const bf16_t* Wd = static_cast<const bf16_t*>(p.wd);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int g = wave * 4 + i;
                const int row = g * 8 + rsub;                       // centre pixel row / Wd output channel
                const int sw = (cpos ^ ((row >> 1) & 7)) * 8;
                dma16(X + (((long)b * p.H + y0 + (row >> 4)) * p.W + x0 + (row & 15)) * CIN + sw, lds + XC_OFF + g * 1024);
                dma16(Wd + (long)row * 64 + sw, lds + (g < 16 ? WD_OFF0 + g * 1024 : WD_OFF1 + (g - 16) * 1024));
            }
        }
        f32x16 acc[4][2];
        {
            u32x4 bq[4][4], bd[4][4];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bq[c][g] = lds_read_b128(lds_base + BIAS_OFF + 512 + ((4 * wn + c) * 32 + 8 * g + 4 * lhalf) * 4);
                    bd[c][g] = lds_read_b128(lds_base + BIAS_OFF + 1536 + ((4 * wn + c) * 32 + 8 * g + 4 * lhalf) * 4);
                }
            lds_wait();
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[c][q][r] = __uint_as_float(bq[c][r >> 2][r & 3]) + __uint_as_float(bd[c][r >> 2][r & 3]);
        }
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            if (kt == 1) {
                wait_vmcnt_n(0);
                __builtin_amdgcn_s_barrier();        // x centre rows + Wd landed for every wave
                asm volatile("" ::: "memory");
            }
            const unsigned w_base = lds_base + (kt == 0 ? WC_OFF + 4 * wn * 32 * ROWB : (wn ? WD_OFF1 : WD_OFF0));
            const unsigned p_base = lds_base + (kt == 0 ? T2_OFF : XC_OFF);
            u32x4 wf[2][4], pf[2][2];
            auto frag_read = [&](int ks, int buf) {
                const int ch = 2 * ks + lhalf;
#pragma unroll
                for (int c = 0; c < 4; ++c) wf[buf][c] = lds_read_b128(w_base + swz(c * 32 + lrow, ch));
#pragma unroll
                for (int q = 0; q < 2; ++q) pf[buf][q] = lds_read_b128(p_base + swz((2 * wm + q) * 32 + lrow, ch));
            };
            frag_read(0, 0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                lds_wait();
                if (ks < 3) frag_read(ks + 1, (ks + 1) & 1);
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int q = 0; q < 2; ++q) acc[c][q] = mfma_dbg(wf[ks & 1][c], pf[ks & 1][q], acc[c][q], p.debug);
            }
        }
        stamp(9);
        __builtin_amdgcn_s_barrier();                // every wave is done with t2 / x / Wc / Wd: their space is store staging
        asm volatile("" ::: "memory");
        float* stg = reinterpret_cast<float*>(lds + wave_scratch(wave));
#pragma unroll
        for (int cp = 0; cp < 2; ++cp)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int slot = (8 * c + 2 * g + lhalf) ^ (lrow & 15);
                        const f32x16& a = acc[2 * cp + c][q];
                        *reinterpret_cast<float4*>(stg + lrow * 64 + slot * 4) = make_float4(a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]);
                    }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int pr = it * 8 + prr;
                    const float4 v0 = *reinterpret_cast<const float4*>(stg + pr * 64 + (((2 * u) ^ (pr & 15)) << 2));
                    const float4 v1 = *reinterpret_cast<const float4*>(stg + pr * 64 + (((2 * u + 1) ^ (pr & 15)) << 2));
                    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                    unsigned pk[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) pk[k] = relu2_bf16(v[2 * k], v[2 * k + 1]);
                    if (!ABL(p.debug, 2) || pk[0] == 0x12345678u) {
                        u32x4 ov = {pk[0], pk[1], pk[2], pk[3]};
                        if (p.nt) __builtin_nontemporal_store(ov, reinterpret_cast<u32x4*>(Yb + goff(cp, q, it)));
                        else *reinterpret_cast<u32x4*>(Yb + goff(cp, q, it)) = ov;
                    }
                }
            }
        stamp(10);
        return;
    }
    // ================================================================ phase C: y = relu(Wc t2 + bc + x), two channel halves
    {
        float* stg = reinterpret_cast<float*>(lds + wave_scratch(wave));
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
            // accumulators start at the bias: channel of acc[c][q][r] = (4wn+2cp+c)*32 + 8*(r>>2) + 4*lhalf + (r&3)
            f32x16 acc[2][2];
            {
                u32x4 bq[2][4];
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        bq[c][g] = lds_read_b128(lds_base + BIAS_OFF + 512 + ((4 * wn + 2 * cp + c) * 32 + 8 * g + 4 * lhalf) * 4);
                lds_wait();
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[c][q][r] = __uint_as_float(bq[c][r >> 2][r & 3]);
            }
            {
                u32x4 wf[2][2], pf[2][2];
                auto frag_read = [&](int ks, int buf) {
                    const int ch = 2 * ks + lhalf;
#pragma unroll
                    for (int c = 0; c < 2; ++c) wf[buf][c] = lds_read_b128(lds_base + WC_OFF + swz((4 * wn + 2 * cp + c) * 32 + lrow, ch));
#pragma unroll
                    for (int q = 0; q < 2; ++q) pf[buf][q] = lds_read_b128(lds_base + T2_OFF + swz((2 * wm + q) * 32 + lrow, ch));
                };
                frag_read(0, 0);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    lds_wait();
                    if (ks < 3) frag_read(ks + 1, (ks + 1) & 1);
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int q = 0; q < 2; ++q) acc[c][q] = mfma_dbg(wf[ks & 1][c], pf[ks & 1][q], acc[c][q], p.debug);
                }
            }
            if (cp == 1) stamp(9);
            // store pass: per-wave staging (no workgroup barrier): [32 px][64 ch] fp32, 16-B slots XOR-swizzled by the
            // pixel row so the 8-lane write groups and the 16-lane read groups are conflict-free
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int slot = (8 * c + 2 * g + lhalf) ^ (lrow & 15);
                        const f32x16& a = acc[c][q];
                        *reinterpret_cast<float4*>(stg + lrow * 64 + slot * 4) = make_float4(a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]);
                    }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int pr = it * 8 + prr;
                    const float4 v0 = *reinterpret_cast<const float4*>(stg + pr * 64 + (((2 * u) ^ (pr & 15)) << 2));
                    const float4 v1 = *reinterpret_cast<const float4*>(stg + pr * 64 + (((2 * u + 1) ^ (pr & 15)) << 2));
                    const u32x4 r4 = rr[cp * 2 + q][it];
                    const unsigned w4[4] = {ABL(p.debug, 1) ? 0u : r4[0], r4[1], r4[2], r4[3]};
                    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                    unsigned pk[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float lo = fmaxf(v[2 * k] + __uint_as_float(w4[k] << 16), 0.f);
                        const float hi = fmaxf(v[2 * k + 1] + __uint_as_float(w4[k] & 0xffff0000u), 0.f);
                        pk[k] = pack2_bf16(lo, hi);
                    }
                    if (!ABL(p.debug, 2) || pk[0] == 0x12345678u) {
                        u32x4 ov = {pk[0], pk[1], pk[2], pk[3]};
                        if (p.nt) __builtin_nontemporal_store(ov, reinterpret_cast<u32x4*>(Yb + goff(cp, q, it)));
                        else *reinterpret_cast<u32x4*>(Yb + goff(cp, q, it)) = ov;
                    }
                }
            }
        }
        stamp(10);
    }
}

bool bneck_fusable(const BneckArgs& a) {
    const bool shape = (a.Cin == 256 && !a.wd) || (a.Cin == 64 && a.wd && a.bd);
    return shape && a.Cmid == 64 && a.H % BT == 0 && a.W % BT == 0 && a.H == a.W &&
           (size_t)a.B * a.H * a.W * 512 < ((size_t)1 << 32);      // 32-bit byte offsets in the store pass
}

void launch_bneck(const BneckArgs& a_in, hipStream_t st) {
    BneckArgs a = a_in;
    a.debug = tune_get("BDBG", 0);
    a.stagger = tune_get("STAGGER", 0);
    a.nt = (tune_get("NT", 3) >> 2) & 1;   // bit 2: measured neutral-to-worse here (the next block re-reads y through L2)
    ConvArgs d{};
    d.B = a.B; d.H = a.H; d.W = a.W; d.Ho = a.H; d.Wo = a.W; d.Cin = a.Cin; d.Cout = 4 * a.Cmid; d.KH = 0; d.stride = 1; d.res = a.wd ? nullptr : a.x;
    void* tok = prof_begin(d, 2, st);
    const int grid = a.B * (a.H / BT) * (a.W / BT);
    if (a.wd) hipLaunchKernelGGL((bneck64_kernel<64, true>), dim3(grid), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((bneck64_kernel<256, false>), dim3(grid), dim3(512), 0, st, a);
    prof_end(tok, st);
}

}  // namespace ivosw

#ifdef IVOSW_PROBES
// Tuning probe (not part of the reference surface): one fused bottleneck launch on caller-provided tensors with
// s_memtime stamps at the phase boundaries of every workgroup: ts [B*(H/16)*(W/16)][16] uint64 (device).
extern "C" int ivosw_bneck_probe(const void* x, void* y, const void* wa, const float* ba, const void* wb, const float* bb,
                                 const void* wc, const float* bc, const void* wd, const float* bd, const void* zeros, int B, int H, int W, int Cin, int Cmid,
                                 unsigned long long* ts, ivosw_stream_t stream) {
    using namespace ivosw;
    IVOSW_REQUIRE(x && y && wa && ba && wb && bb && wc && bc && zeros, "null pointer");
    IVOSW_ON_DEVICE_OF(y);
    BneckArgs a{};
    a.x = x; a.y = y; a.wa = wa; a.ba = ba; a.wb = wb; a.bb = bb; a.wc = wc; a.bc = bc; a.wd = wd; a.bd = bd; a.zeros = zeros;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cmid = Cmid; a.ts = ts;
    IVOSW_REQUIRE(bneck_fusable(a), "shape is not covered by the fused bottleneck kernels");
    launch_bneck(a, as_stream(stream));
    IVOSW_CHECK_LAUNCH();
    return IVOSW_OK;
}
#endif  // IVOSW_PROBES
