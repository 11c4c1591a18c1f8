// Segmentation-head epilogue feeding the assessment path — SURVEY §8(f) row 4.
//
// Reference (utils/utils_manet.py, once per frame and once per interaction):
//     pred_label = nn.functional.interpolate(pred_label, size=(h, w), mode='bilinear', align_corners=True)   :78-79,110-111,146-147
//     probs.append(pred_label); pred_label = torch.argmax(pred_label, dim=1); pred_masks.append(pred_label.float())   :80-82
//     all_P = torch.softmax(torch.cat(probs_reverse, 0), 1)                                                    :161
// i.e. the (O+1)-channel logits are upsampled to full resolution and kept for every frame (n x (O+1) x H x W fp32), read
// again by argmax, concatenated (another copy) and read again by softmax.  Here ONE kernel per frame batch reads the
// low-resolution logits (cache resident) and writes each output exactly once: the label map (int64 for the propagation
// step, optionally uint8 for the J/F kernels and float for final_masks) and the softmax probabilities straight into
// their slot of all_P — with caller-chosen strides, so all_P can be stored object-major ([C][n][H][W]) and the
// per-object soft masks the assessment network takes are contiguous slices (no transpose copy).
//
// Arithmetic follows ATen: upsample_bilinear2d with align_corners (scale = (in-1)/(out-1), src = scale * dst, lambda =
// src - floor, clamped +1 neighbour), evaluated without FMA contraction (#pragma clang fp contract(off)); argmax = first maximum; softmax =
// exp(x - max) / sum(exp(x - max)).  HBM-bound on the probability stream ((O+1) x 4 B per pixel).
#include "common.h"
#include <stdlib.h>

namespace ivosw {

namespace {
struct SegArgs {
    const float* logits;   // [n][C][hs][ws]
    int n, C, hs, ws, H, W;
    float rh, rw;          // (hs-1)/(H-1), (ws-1)/(W-1) (0 when the output extent is 1)
    float* probs;          // element (f, c, y, x) at probs[f*sn + c*sc + y*W + x], or nullptr
    long sn, sc;
    int64_t* label_i64;    // [n][H][W] or nullptr
    uint8_t* label_u8;
    float* label_f32;
};

__device__ __forceinline__ float sample(const float* __restrict__ p, int ws, int h1, int h1p, int w1, int w1p, float h0l, float h1l,
                                        float w0l, float w1l) {
#pragma clang fp contract(off)
    const float* r0 = p + (size_t)h1 * ws;
    const float* r1 = r0 + (size_t)h1p * ws;
    const float top = w0l * r0[w1] + w1l * r0[w1 + w1p];
    const float bot = w0l * r1[w1] + w1l * r1[w1 + w1p];
    return h0l * top + h1l * bot;
}
}  // namespace

// CMAX > 0: channel values live in registers (C <= CMAX); CMAX == 0: any C, channels re-sampled per pass.
// V = pixels per thread (consecutive in the H*W plane): V = 4 turns every output stream into 16-byte (labels: 32 / 4 /
// 16-byte) stores; it needs H*W and the probability strides to be multiples of 4, otherwise V = 1.
// FULL: C == CMAX, the per-channel guards disappear (1090 VALU instructions per wave made this kernel VALU-bound: the
// guards, 64-bit index arithmetic and a 64-bit division per pixel were a third of them).
template <int CMAX, int V, bool FULL = false>
__global__ __launch_bounds__(256) void seg_epilogue_kernel(SegArgs a) {
    // no FMA contraction: fma(rw, x, -floor) keeps the exact product and moves the interpolation weight by an ulp of the
    // SOURCE coordinate (1.5e-5 at x ~ 200), i.e. the logits by ~3e-5 — ATen rounds scale * index to fp32 first
    // (HIP's __fmul_rn / __fsub_rn are plain operators inside header functions compiled with contraction ON: they fuse
    // after inlining no matter what the caller says, so the arithmetic below uses bare operators under this pragma)
#pragma clang fp contract(off)
    const int f = blockIdx.y;
    const unsigned idx0 = (blockIdx.x * 256u + threadIdx.x) * V;          // H * W < 2^31 (checked by the launcher)
    if (idx0 >= (unsigned)(a.H * a.W)) return;
    const int C = FULL ? CMAX : a.C;
    unsigned yy = idx0 / (unsigned)a.W, xx = idx0 - yy * (unsigned)a.W;      // pixel p: (yy, xx), advanced incrementally
    const float* base = a.logits + (size_t)f * a.C * a.hs * a.ws;
    const size_t plane = (size_t)a.hs * a.ws;
    constexpr int CR = CMAX > 0 ? CMAX : 1;
    float pv[CR][V];
    int am[V];
#pragma unroll
    for (int p = 0; p < V; ++p) {
        const unsigned idx = idx0 + p;
        const int y = (int)yy, x = (int)xx;
        if (++xx == (unsigned)a.W) { xx = 0; ++yy; }
        const float h1r = a.rh * (float)y, w1r = a.rw * (float)x;
        const int h1 = (int)h1r, w1 = (int)w1r;
        const int h1p = h1 < a.hs - 1 ? 1 : 0, w1p = w1 < a.ws - 1 ? 1 : 0;
        const float h1l = h1r - (float)h1, w1l = w1r - (float)w1;
        const float h0l = 1.f - h1l, w0l = 1.f - w1l;
        float best;
        am[p] = 0;
        if (CMAX > 0) {
#pragma unroll
            for (int c = 0; c < CMAX; ++c) pv[c][p] = c < C ? sample(base + c * plane, a.ws, h1, h1p, w1, w1p, h0l, h1l, w0l, w1l) : 0.f;
            best = pv[0][p];
#pragma unroll
            for (int c = 1; c < CMAX; ++c)
                if (c < C && pv[c][p] > best) { best = pv[c][p]; am[p] = c; }
            if (a.probs) {
                // VALU-bound kernel: exp through the hardware exp2 (v_exp_f32; arguments are <= 0, the absolute error stays
                // below 1e-7) and ONE reciprocal per pixel instead of C IEEE divisions — inside the 2e-6 bar against
                // torch.softmax (the generic-C kernel below keeps expf and the division)
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < CMAX; ++c)
                    if (c < C) { pv[c][p] = __expf(pv[c][p] - best); s += pv[c][p]; }
                const float inv = 1.0f / s;
#pragma unroll
                for (int c = 0; c < CMAX; ++c)
                    if (c < C) pv[c][p] = pv[c][p] * inv;
            }
        } else {
            best = sample(base, a.ws, h1, h1p, w1, w1p, h0l, h1l, w0l, w1l);
            for (int c = 1; c < C; ++c) {
                const float v = sample(base + c * plane, a.ws, h1, h1p, w1, w1p, h0l, h1l, w0l, w1l);
                if (v > best) { best = v; am[p] = c; }
            }
            if (a.probs) {
                float s = 0.f;
                for (int c = 0; c < C; ++c) s += expf(sample(base + c * plane, a.ws, h1, h1p, w1, w1p, h0l, h1l, w0l, w1l) - best);
                float* out = a.probs + (size_t)f * a.sn + idx;
                for (int c = 0; c < C; ++c)
                    out[(size_t)c * a.sc] = expf(sample(base + c * plane, a.ws, h1, h1p, w1, w1p, h0l, h1l, w0l, w1l) - best) / s;
            }
        }
    }
    if (CMAX > 0 && a.probs) {
        float* out = a.probs + (size_t)f * a.sn + idx0;
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (c < C) {
                if (V == 4) *reinterpret_cast<float4*>(out + (size_t)c * a.sc) = make_float4(pv[c][0], pv[c][1], pv[c][2], pv[c][V - 1]);
                else out[(size_t)c * a.sc] = pv[c][0];
            }
    }
    const size_t li = (size_t)f * a.H * a.W + idx0;
    if (V == 4) {
        if (a.label_i64) {
            *reinterpret_cast<longlong2*>(a.label_i64 + li) = make_longlong2(am[0], am[1]);
            *reinterpret_cast<longlong2*>(a.label_i64 + li + 2) = make_longlong2(am[2], am[V - 1]);
        }
        if (a.label_u8) *reinterpret_cast<uint32_t*>(a.label_u8 + li) = (uint32_t)am[0] | ((uint32_t)am[1] << 8) | ((uint32_t)am[2] << 16) | ((uint32_t)am[V - 1] << 24);
        if (a.label_f32) *reinterpret_cast<float4*>(a.label_f32 + li) = make_float4((float)am[0], (float)am[1], (float)am[2], (float)am[V - 1]);
    } else {
        if (a.label_i64) a.label_i64[li] = am[0];
        if (a.label_u8) a.label_u8[li] = (uint8_t)am[0];
        if (a.label_f32) a.label_f32[li] = (float)am[0];
    }
}

}  // namespace ivosw

extern "C" int ivosw_seg_epilogue(const float* logits, int n, int C, int hs, int ws, int H, int W, float* probs,
                                  long probs_stride_n, long probs_stride_c, int64_t* label_i64, uint8_t* label_u8,
                                  float* label_f32, ivosw_stream_t stream) {
    using namespace ivosw;
    IVOSW_REQUIRE(logits, "null logits");
    IVOSW_ON_DEVICE_OF(logits);
    IVOSW_REQUIRE(probs || label_i64 || label_u8 || label_f32, "no output requested");
    IVOSW_REQUIRE(n > 0 && n <= 65535 && C > 0 && hs > 0 && ws > 0 && H > 0 && W > 0, "bad shape");
    IVOSW_REQUIRE((long)H * W < (1L << 31) && (long)C * hs * ws < (1L << 31), "frame too large for 32-bit pixel indices");
    IVOSW_REQUIRE(!label_u8 || C <= 256, "uint8 labels need C <= 256");
    IVOSW_REQUIRE(!probs || (probs_stride_c >= (long)H * W && probs_stride_n >= (long)H * W), "probability strides overlap");
    SegArgs a{};
    a.logits = logits; a.n = n; a.C = C; a.hs = hs; a.ws = ws; a.H = H; a.W = W;
    a.rh = H > 1 ? (float)(hs - 1) / (float)(H - 1) : 0.f;      // ATen area_pixel_compute_scale, align_corners = true
    a.rw = W > 1 ? (float)(ws - 1) / (float)(W - 1) : 0.f;
    a.probs = probs; a.sn = probs_stride_n; a.sc = probs_stride_c;
    a.label_i64 = label_i64; a.label_u8 = label_u8; a.label_f32 = label_f32;
    hipStream_t st = as_stream(stream);
    const bool vec = getenv("IVOSW_SEG_SCALAR") == nullptr && ((long)H * W) % 4 == 0 && (!probs || (probs_stride_n % 4 == 0 && probs_stride_c % 4 == 0 && ((uintptr_t)probs & 15) == 0)) &&
                     (((uintptr_t)label_i64 | (uintptr_t)label_f32) & 15) == 0 && ((uintptr_t)label_u8 & 3) == 0;
    const long items = vec ? (long)H * W / 4 : (long)H * W;
    const dim3 grid((unsigned)((items + 255) / 256), n);
#define IVOSW_SEG_LAUNCH(CM)                                                                                   \
    do {                                                                                                       \
        if (vec && C == CM) hipLaunchKernelGGL((seg_epilogue_kernel<CM, 4, true>), grid, dim3(256), 0, st, a); \
        else if (vec) hipLaunchKernelGGL((seg_epilogue_kernel<CM, 4>), grid, dim3(256), 0, st, a);             \
        else hipLaunchKernelGGL((seg_epilogue_kernel<CM, 1>), grid, dim3(256), 0, st, a);                      \
    } while (0)
    if (C <= 4) IVOSW_SEG_LAUNCH(4);
    else if (C <= 8) IVOSW_SEG_LAUNCH(8);
    else if (C <= 16) IVOSW_SEG_LAUNCH(16);
    else hipLaunchKernelGGL((seg_epilogue_kernel<0, 1>), dim3((unsigned)(((long)H * W + 255) / 256), n), dim3(256), 0, st, a);
#undef IVOSW_SEG_LAUNCH
    IVOSW_CHECK_LAUNCH();
    return IVOSW_OK;
}
