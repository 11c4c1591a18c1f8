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
//   jf_boundary_kernel  one thread = 32 pixels of one row: object masks of rows y, y+1 -> boundary words (bit-packed,
//                       ws) + intersection / union / boundary-pixel counts (block reduce, one atomic per counter)
//   jf_match_kernel     one thread = one boundary word of map A: skipped when empty (boundaries are sparse), otherwise
//                       ORs the (2r+1) row-smears of map B around it (disk = per-row half widths) and counts A & dil(B)
#include "common.h"

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

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
}  // namespace

// 33 label bytes (this word's 32 pixels + the east neighbour of the last one) as 9 dwords.  Interior words take 9
// unaligned dword loads (amdhsa runs with unaligned access mode; rows are not 4-byte aligned for W = 854); the last
// word(s) of a row, which would read past the row end, assemble the bytes one by one and zero-fill beyond the image.
__device__ __forceinline__ void load_labels(const uint8_t* row, int x0, int W, bool present, uint32_t (&w)[9]) {
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = 0u;
    if (!present) return;
    if (x0 + 36 <= W) {
#pragma unroll
        for (int k = 0; k < 9; ++k) __builtin_memcpy(&w[k], row + x0 + 4 * k, 4);
    } else {
        for (int k = 0; k < 33; ++k)
            if (x0 + k < W) w[k >> 2] |= (uint32_t)row[x0 + k] << (8 * (k & 3));
    }
}

// bit k = (label k == id), k = 0..32: SWAR zero-byte test on word ^ id, the four byte flags gathered by a multiply
__device__ __forceinline__ uint64_t label_mask(const uint32_t (&w)[9], uint32_t id4, uint8_t id) {
    uint64_t m = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t v = w[k] ^ id4;
        const uint32_t t = ~((((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) | 0x7F7F7F7Fu);    // 0x80 in every byte of v that is zero
        const uint32_t nib = (((t >> 7) * 0x00204081u) >> 21) & 0xFu;
        m |= (uint64_t)nib << (4 * k);
    }
    m |= (uint64_t)((w[8] & 0xFFu) == id) << 32;
    return m;
}

// Measured (100 frames, 3 objects, 480p): 111 us, of which the label loads are ~90 (with the loads ablated both kernels
// together take 72 of 160 us).  Tried and dropped: staging the rows through LDS with lane-contiguous dword loads (210 us
// for both kernels), four rows per thread with the south row reused (172 us: fewer loads but 4x fewer threads and 150
// VGPRs — the loads are latency-, not count-bound).
__global__ __launch_bounds__(256) void jf_boundary_kernel(JfArgs a) {
    const int n = blockIdx.y;
    const int item = blockIdx.x * 256 + threadIdx.x;
    const int y = item / a.WW, j = item - y * a.WW;
    const bool live = y < a.H;
    const int x0 = j * 32;
    // labels of this thread's 33 pixels in rows y and y+1 of both maps
    uint32_t g0[9], g1[9], p0[9], p1[9];
    {
        const size_t ro = ((size_t)n * a.H + (live ? y : 0)) * a.W;
        const bool has_s = live && y + 1 < a.H;
        load_labels(a.gt + ro, x0, a.W, live, g0);
        load_labels(a.pred + ro, x0, a.W, live, p0);
        load_labels(a.gt + ro + a.W, x0, a.W, has_s, g1);
        load_labels(a.pred + ro + a.W, x0, a.W, has_s, p1);
    }
    const int nvalid = live ? min(32, a.W - x0) : 0;                       // pixels of this word inside the image
    const uint32_t vm = nvalid >= 32 ? 0xffffffffu : ((1u << nvalid) - 1u);
    const uint64_t vm33 = nvalid >= 32 ? ~0ull : ((1ull << nvalid) - 1ull);  // zero-filled bytes must not match id 0
    const int lastbit = a.W - 1 - x0;                                       // bit of the last column, if in this word
    const uint32_t lastcol = (lastbit >= 0 && lastbit < 32) ? (1u << lastbit) : 0u;
    const bool lastrow = (y == a.H - 1);
    __shared__ unsigned long long red[4][4];
    for (int o = 0; o < a.n_obj; ++o) {
        unsigned long long c_int = 0, c_uni = 0, c_fg = 0, c_gt = 0;
        if (live) {
            const uint8_t id = a.ids[o];
            const uint32_t id4 = id * 0x01010101u;
            const uint64_t mg0 = label_mask(g0, id4, id) & vm33, mg1 = label_mask(g1, id4, id) & vm33;
            const uint64_t mp0 = label_mask(p0, id4, id) & vm33, mp1 = label_mask(p1, id4, id) & vm33;
            auto bmap = [&](uint64_t m0, uint64_t m1) {
                const uint32_t seg = (uint32_t)m0, e = (uint32_t)(m0 >> 1), s = (uint32_t)m1, se = (uint32_t)(m1 >> 1);
                uint32_t b;
                if (lastrow) b = (seg ^ e) & ~lastcol;                      // b[-1, :] = seg ^ e ; b[-1, -1] = 0
                else b = (((seg ^ e) | (seg ^ s) | (seg ^ se)) & ~lastcol) | ((seg ^ s) & lastcol);   // b[:, -1] = seg ^ s
                return b & vm;
            };
            const uint32_t bgw = bmap(mg0, mg1), bpw = bmap(mp0, mp1);
            const size_t widx = (((size_t)o * a.N + n) * a.H + y) * a.WW + j;
            a.bg[widx] = bgw;
            a.bp[widx] = bpw;
            const uint32_t sg = (uint32_t)mg0 & vm, sp = (uint32_t)mp0 & vm;
            c_int = __popc(sg & sp);
            c_uni = __popc(sg | sp);
            c_fg = __popc(bpw);
            c_gt = __popc(bgw);
        }
        c_int = wave_sum(c_int); c_uni = wave_sum(c_uni); c_fg = wave_sum(c_fg); c_gt = wave_sum(c_gt);
        const int wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { red[wave][0] = c_int; red[wave][1] = c_uni; red[wave][2] = c_fg; red[wave][3] = c_gt; }
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
            for (int dy = -a.r; dy <= a.r; ++dy) {
                const int yy = y + dy;
                if (yy < 0 || yy >= a.H) continue;
                const int w = a.hw[dy < 0 ? -dy : dy];
                const uint32_t* row = oth + (size_t)yy * a.WW;
                const uint64_t cur = row[j];
                const uint64_t prev = j > 0 ? row[j - 1] : 0u;
                const uint64_t next = j + 1 < a.WW ? row[j + 1] : 0u;
                if (!(cur | prev | next)) continue;
                // a set bit at x covers x-w .. x+w: smear towards higher x inside [prev | cur], towards lower x inside [cur | next]
                uint64_t lo = prev | (cur << 32), hi = cur | (next << 32);
                uint64_t sl = lo, sr = hi;
                for (int covered = 0; covered < w;) {
                    const int s = min(covered + 1, w - covered);
                    sl |= sl << s;
                    sr |= sr >> s;
                    covered += s;
                }
                dil |= (uint32_t)(sl >> 32) | (uint32_t)sr;
            }
            c = __popc(mine & dil);
        }
    }
    c = wave_sum(c);
    __shared__ unsigned long long red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long v = red[0] + red[1] + red[2] + red[3];
        if (v) atomicAdd(a.counts + ((size_t)n * a.n_obj + o) * 6 + 4 + blockIdx.z, v);
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
    IVOSW_REQUIRE(N > 0 && H > 0 && W > 0, "N, H, W must be positive");
    IVOSW_REQUIRE(n_obj > 0 && n_obj <= JF_MAX_OBJ, "1..32 object ids");
    IVOSW_REQUIRE((long)N * n_obj <= 65535, "N * n_obj must fit one grid dimension (65535): split the sequence");
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
    hipLaunchKernelGGL(jf_boundary_kernel, dim3(nblk, N), dim3(256), 0, st, a);
    hipLaunchKernelGGL(jf_match_kernel, dim3(nblk, N * n_obj, 2), dim3(256), 0, st, a);
    IVOSW_CHECK_LAUNCH();
    return IVOSW_OK;
}
