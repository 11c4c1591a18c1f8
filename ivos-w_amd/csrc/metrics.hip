// DAVIS region similarity J (Jaccard) and contour accuracy F (boundary F-measure) on the device — SURVEY §8(f) row 3,
// the step right before the hot path in every interaction loop (oracle state, rewards, J&F logging).
//
// Reference call site: utils/misc.py:118-162 (sequence_metric) -> davisinteractive.metrics.batched_jaccard /
// batched_f_measure (davisinteractive==1.0.4, requirements.txt:14; the package is NOT under /root/reference, so the
// algorithm below follows its published definition — the DAVIS evaluation code — and parity is unpinned, see
// oracle/jf_oracle.py).  Per frame and object id:
//   J: |gt & pred| / |gt | pred|, 1 when the union is empty
//   F: boundary maps b = (seg ^ east) | (seg ^ south) | (seg ^ south-east) with the last row / column special-cased;
//      each boundary is dilated by a disk of radius bound_pix = ceil(0.008 * |(H, W)|) (8 at 480p); precision = matched
//      pred-boundary pixels / pred-boundary pixels, recall likewise for gt; F = 2PR / (P + R)
//
// This is integer / bit work bound by HBM (two label maps in) and latency: the kernels produce the six INTEGER counts per
// (frame, object); the float64 ratios are formed on the host with the reference's own expressions, so results are
// bit-identical whenever the counts are.
//   jf_boundary_row_kernel  one lane = 16 pixels of a row (16-byte loads contiguous across the wave), four rows per wave:
//                       object masks by a SWAR byte compare -> boundary words (bit-packed, ws) + intersection / union /
//                       boundary-pixel counts (DPP wave reduction, one atomic per counter and block)
//   jf_match_kernel     one thread = one boundary word of map A: skipped when empty (boundaries are sparse), otherwise
//                       ORs the (2r+1) row-smears of map B around it (disk = per-row half widths) and counts A & dil(B)
#include "common.h"
#include <stdlib.h>

namespace ivosw {

namespace {
constexpr int JF_MAX_OBJ = 32;
constexpr int JF_MAX_R = 32;

struct JfArgs {
    const uint8_t* gt;
    const uint8_t* pred;
    int N, H, W, WW;              // WW = words per bitmap row
    int n_obj;
    uint8_t ids[JF_MAX_OBJ];
    int r;
    uint8_t hw[JF_MAX_R + 1];     // half width of the disk at |dy|: floor(sqrt(r^2 - dy^2))
    uint32_t* bg;                 // gt boundary bitmaps   [n_obj][N][H][WW]
    uint32_t* bp;                 // pred boundary bitmaps
    unsigned long long* counts;   // [N][n_obj][6]: inter, union, n_fg (pred boundary), n_gt, fg_match, gt_match
};

}  // namespace

// Boundary kernel: a wave owns one row segment of up to 1024 pixels, lane l the 16 pixels [16 l, 16 l + 16) — the label
// loads of a lane are 16-byte loads that are CONTIGUOUS across the wave (every byte used once).  East / south-east
// neighbours come from the next lane; a lane pair forms one bitmap word.  History (100 frames x 3 objects at 480p):
// one thread per 32-pixel word with 36-byte windows at a 32-byte lane stride 111 us -> this layout 94 us (the loads were
// never the limit: the kernel is VALU-bound) -> two counters per 32-bit word through the reduction 57 us -> DPP instead
// of ds_bpermute shuffles 53 us -> four rows per wave with the south row's masks reused 47 us.
//
// DPP cross-lane moves (one VALU instruction each; __shfl_down is a ds_bpermute plus address arithmetic, and this kernel
// is VALU-bound).  wave_shl:1 = every lane reads lane + 1 (0 past the end); the reduction is the row_shr 1/2/4/8 scan
// followed by row_bcast15 / row_bcast31, total in lane 63.
__device__ __forceinline__ uint32_t lane_next(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t wave_total_in_lane63(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8 -> lane 15 of each row holds the row sum
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast15 into rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast31 into rows 2, 3
    return v;
}

__device__ __forceinline__ uint32_t mask16(const uint32_t (&w)[4], uint32_t id4) {
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t v = w[k] ^ id4;
        const uint32_t t = ~((((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) | 0x7F7F7F7Fu);    // 0x80 in every zero byte of v
        m |= ((((t >> 7) * 0x00204081u) >> 21) & 0xFu) << (4 * k);
    }
    return m;
}

__device__ __forceinline__ void load16(const uint8_t* base, size_t off, size_t total, bool on, uint32_t (&w)[4]) {
    w[0] = w[1] = w[2] = w[3] = 0u;
    if (!on) return;
    if (off + 16 <= total) {
        __builtin_memcpy(w, base + off, 16);          // bytes past the row end belong to the next row: masked by the caller
    } else {
        for (int k = 0; k < 16; ++k)
            if (off + k < total) w[k >> 2] |= (uint32_t)base[off + k] << (8 * (k & 3));
    }
}

// A wave walks JF_RPW consecutive rows: the south row of one step is the own row of the next, so JF_RPW + 1 row loads and
// mask computations serve JF_RPW rows, and the counts are reduced once per object and wave (the kernel is VALU-bound).
constexpr int JF_RPW = 4;
__global__ __launch_bounds__(256) void jf_boundary_row_kernel(JfArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.z, seg = blockIdx.y;
    const int y0 = (blockIdx.x * 4 + wave) * JF_RPW;
    const int x0 = seg * 1024 + lane * 16;
    const bool col_live = x0 < a.W;
    const size_t total = (size_t)a.N * a.H * a.W;
    uint32_t g[JF_RPW + 1][4], p[JF_RPW + 1][4];
    uint8_t tg[JF_RPW + 1], tp[JF_RPW + 1];          // the pixel right of this lane's 16 at the end of a 1024-pixel segment
    const bool tail = (lane == 63) && col_live && x0 + 16 < a.W;
#pragma unroll
    for (int r = 0; r <= JF_RPW; ++r) {
        const int y = y0 + r;
        const bool on = col_live && y < a.H;
        const size_t off = ((size_t)n * a.H + (y < a.H ? y : 0)) * a.W + x0;
        load16(a.gt, off, total, on, g[r]);
        load16(a.pred, off, total, on, p[r]);
        tg[r] = (tail && on) ? a.gt[off + 16] : 0;
        tp[r] = (tail && on) ? a.pred[off + 16] : 0;
    }
    const int nvalid = col_live ? min(16, a.W - x0) : 0;
    const uint32_t vm = (1u << nvalid) - 1u;
    const int lastbit = a.W - 1 - x0;
    const uint32_t lastcol = (lastbit >= 0 && lastbit < 16) ? (1u << lastbit) : 0u;
    __shared__ unsigned long long red[4][4];
    for (int o = 0; o < a.n_obj; ++o) {
        const uint8_t id = a.ids[o];
        const uint32_t id4 = id * 0x01010101u;
        // bits 0..15 = this lane's pixels, bit 16 = the east neighbour of pixel 15 (rows >= H load zeros: no match for id > 0,
        // and the valid mask clears id == 0 matches of rows that do not exist through `rows` below)
        uint32_t mg[JF_RPW + 1], mp[JF_RPW + 1];
#pragma unroll
        for (int r = 0; r <= JF_RPW; ++r) {
            const bool row_ok = y0 + r < a.H;
            const uint32_t m0 = row_ok ? (mask16(g[r], id4) & vm) : 0u, m1 = row_ok ? (mask16(p[r], id4) & vm) : 0u;
            const uint32_t e0 = lane == 63 ? ((tail && row_ok && tg[r] == id) ? 1u : 0u) : (lane_next(m0) & 1u);
            const uint32_t e1 = lane == 63 ? ((tail && row_ok && tp[r] == id) ? 1u : 0u) : (lane_next(m1) & 1u);
            mg[r] = m0 | (e0 << 16);
            mp[r] = m1 | (e1 << 16);
        }
        uint32_t c01 = 0, c23 = 0;
#pragma unroll
        for (int r = 0; r < JF_RPW; ++r) {
            const int y = y0 + r;
            const bool lastrow = (y == a.H - 1);
            auto bmap = [&](uint32_t m0, uint32_t m1) {
                const uint32_t seg16 = m0 & 0xffffu, e = (m0 >> 1) & 0xffffu, s = m1 & 0xffffu, se = (m1 >> 1) & 0xffffu;
                uint32_t b;
                if (lastrow) b = (seg16 ^ e) & ~lastcol;                                   // b[-1, :] = seg ^ e ; b[-1, -1] = 0
                else b = (((seg16 ^ e) | (seg16 ^ s) | (seg16 ^ se)) & ~lastcol) | ((seg16 ^ s) & lastcol);   // b[:, -1] = seg ^ s
                return b & vm;
            };
            const bool row_ok = y < a.H;
            const uint32_t bgw = row_ok ? bmap(mg[r], mg[r + 1]) : 0u, bpw = row_ok ? bmap(mp[r], mp[r + 1]) : 0u;
            // lanes 2k, 2k+1 -> bitmap word k of the segment
            const uint32_t hg = lane_next(bgw), hp = lane_next(bpw);
            if (row_ok && col_live && !(lane & 1)) {
                const size_t widx = (((size_t)o * a.N + n) * a.H + y) * a.WW + (x0 >> 5);
                a.bg[widx] = bgw | (hg << 16);
                a.bp[widx] = bpw | (hp << 16);
            }
            const uint32_t sg = mg[r] & 0xffffu, sp = mp[r] & 0xffffu;
            // each count is <= 16 per lane and row, <= 4096 per wave: two counters share one 32-bit word through the reduction
            c01 += __popc(sg & sp) | (__popc(sg | sp) << 16);
            c23 += __popc(bpw) | (__popc(bgw) << 16);
        }
        c01 = wave_total_in_lane63(c01);
        c23 = wave_total_in_lane63(c23);
        if (lane == 63) { red[wave][0] = c01 & 0xffffu; red[wave][1] = c01 >> 16; red[wave][2] = c23 & 0xffffu; red[wave][3] = c23 >> 16; }
        __syncthreads();
        if (threadIdx.x < 4) {
            const unsigned long long v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
            if (v) atomicAdd(a.counts + ((size_t)n * a.n_obj + o) * 6 + threadIdx.x, v);
        }
        __syncthreads();
    }
}

// grid (blocks over H*WW, N * n_obj, 2): z = 0 counts pred-boundary pixels inside dil(gt boundary) (fg_match),
// z = 1 gt-boundary pixels inside dil(pred boundary) (gt_match)
__global__ __launch_bounds__(256) void jf_match_kernel(JfArgs a) {
    const int on = blockIdx.y;                    // o * N + n
    const int o = on / a.N, n = on - o * a.N;
    const int item = blockIdx.x * 256 + threadIdx.x;
    const int y = item / a.WW, j = item - y * a.WW;
    const uint32_t* own = (blockIdx.z == 0 ? a.bp : a.bg) + (size_t)on * a.H * a.WW;
    const uint32_t* oth = (blockIdx.z == 0 ? a.bg : a.bp) + (size_t)on * a.H * a.WW;
    unsigned long long c = 0;
    if (y < a.H) {
        const uint32_t mine = own[(size_t)y * a.WW + j];
        if (mine) {
            uint32_t dil = 0;
            // rows in groups of 8: the 24 loads of a group are issued together (one dependent L2 round trip per group instead
            // of one per row)
            for (int d0 = -a.r; d0 <= a.r; d0 += 8) {
                uint32_t cu[8], pv[8], nx[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int dy = d0 + k, yy = y + dy;
                    const bool ok = dy <= a.r && yy >= 0 && yy < a.H;
                    const uint32_t* row = oth + (size_t)(ok ? yy : y) * a.WW;
                    cu[k] = ok ? row[j] : 0u;
                    pv[k] = (ok && j > 0) ? row[j - 1] : 0u;
                    nx[k] = (ok && j + 1 < a.WW) ? row[j + 1] : 0u;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (!(cu[k] | pv[k] | nx[k])) continue;
                    const int dy = d0 + k;
                    const int w = a.hw[dy < 0 ? -dy : dy];
                    // a set bit at x covers x-w .. x+w: smear towards higher x inside [prev | cur], towards lower x inside [cur | next]
                    uint64_t sl = (uint64_t)pv[k] | ((uint64_t)cu[k] << 32), sr = (uint64_t)cu[k] | ((uint64_t)nx[k] << 32);
                    for (int covered = 0; covered < w;) {
                        const int sh = min(covered + 1, w - covered);
                        sl |= sl << sh;
                        sr |= sr >> sh;
                        covered += sh;
                    }
                    dil |= (uint32_t)(sl >> 32) | (uint32_t)sr;
                }
            }
            c = __popc(mine & dil);
        }
    }
    // boundaries are sparse: most waves have nothing to add and skip the reduction; a word contributes <= 32
    if (__ballot(c != 0) != 0ull) {
        const uint32_t t = wave_total_in_lane63((uint32_t)c);
        if ((threadIdx.x & 63) == 63 && t) atomicAdd(a.counts + ((size_t)n * a.n_obj + o) * 6 + 4 + blockIdx.z, (unsigned long long)t);
    }
}

}  // namespace ivosw

extern "C" size_t ivosw_jf_ws_bytes(int N, int H, int W, int n_obj) {
    if (N <= 0 || H <= 0 || W <= 0 || n_obj <= 0) return 0;
    const size_t ww = (size_t)(W + 31) / 32;
    return 2 * ivosw::align_up((size_t)n_obj * N * H * ww * sizeof(uint32_t), 256);
}

extern "C" int ivosw_jf_counts(const uint8_t* gt, const uint8_t* pred, int N, int H, int W, const uint8_t* obj_ids,
                               int n_obj, int bound_pix, int64_t* counts, void* ws, size_t ws_bytes,
                               ivosw_stream_t stream) {
    using namespace ivosw;
    IVOSW_REQUIRE(gt && pred && obj_ids && counts && ws, "null pointer");
    IVOSW_ON_DEVICE_OF(counts);
    IVOSW_REQUIRE(N > 0 && H > 0 && W > 0, "N, H, W must be positive");
    IVOSW_REQUIRE(n_obj > 0 && n_obj <= JF_MAX_OBJ, "1..32 object ids");
    IVOSW_REQUIRE((long)N * n_obj <= 65535, "N * n_obj must fit one grid dimension (65535): split the sequence");
    IVOSW_REQUIRE(W <= 65535 * 16, "image too wide");
    IVOSW_REQUIRE(bound_pix >= 0 && bound_pix <= JF_MAX_R, "boundary tolerance 0..32 pixels");
    if (ws_bytes < ivosw_jf_ws_bytes(N, H, W, n_obj)) {
        set_error("ivosw_jf_counts: workspace %zu < %zu", ws_bytes, ivosw_jf_ws_bytes(N, H, W, n_obj));
        return IVOSW_ERR_WS;
    }
    hipStream_t st = as_stream(stream);
    JfArgs a{};
    a.gt = gt; a.pred = pred; a.N = N; a.H = H; a.W = W; a.WW = (W + 31) / 32; a.n_obj = n_obj; a.r = bound_pix;
    for (int o = 0; o < n_obj; ++o) a.ids[o] = obj_ids[o];
    for (int d = 0; d <= bound_pix; ++d) {        // skimage.morphology.disk(r): x^2 + y^2 <= r^2 (integer arithmetic)
        int w = 0;
        while ((w + 1) * (w + 1) + d * d <= bound_pix * bound_pix) ++w;
        a.hw[d] = (uint8_t)w;
    }
    Arena ar(ws);
    a.bg = ar.take<uint32_t>((size_t)n_obj * N * H * a.WW);
    a.bp = ar.take<uint32_t>((size_t)n_obj * N * H * a.WW);
    a.counts = reinterpret_cast<unsigned long long*>(counts);
    (void)hipMemsetAsync(counts, 0, (size_t)N * n_obj * 6 * sizeof(int64_t), st);
    const unsigned nblk = (unsigned)(((size_t)H * a.WW + 255) / 256);
    hipLaunchKernelGGL(jf_boundary_row_kernel, dim3((H + 4 * JF_RPW - 1) / (4 * JF_RPW), (W + 1023) / 1024, N), dim3(256), 0, st, a);
    hipLaunchKernelGGL(jf_match_kernel, dim3(nblk, N * n_obj, 2), dim3(256), 0, st, a);
    IVOSW_CHECK_LAUNCH();
    return IVOSW_OK;
}
