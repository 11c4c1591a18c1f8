// Assessment front end: mask -> bounding box (K1/K2) and fused ROI resample + normalise (K3).
//
// Reference: AssessNet.forward/all2yxhw/get_ROI_grid (models/assessment.py:164-174,110-161,75-108) and
// Encoder.forward's (f-mean)/std (:47).  The reference copies the mask to the host and loops in numpy;
// here a coalesced scan reduces min/max per sample with wavefront reductions + one atomic per block, the
// float64 box arithmetic runs in a per-sample epilogue thread, and the 256x256x2 sampling grid (and the
// unused inverse grid) is never materialised: sample coordinates are recomputed from (y,x,h,w) per pixel.
// Both kernels are HBM-bound streaming/gather work; no LDS staging is needed (each input byte is read once).
#include <limits.h>

#include "common.h"
#include "front.h"

namespace ivosw {

__global__ void bbox_init_kernel(int32_t* __restrict__ box, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    box[b * 4 + 0] = INT_MAX;  // ymin
    box[b * 4 + 1] = -1;       // ymax
    box[b * 4 + 2] = INT_MAX;  // xmin
    box[b * 4 + 3] = -1;       // xmax
}

// grid (S, B): block s scans elements [s*chunk, (s+1)*chunk) of sample b's flat H*W plane.
__global__ __launch_bounds__(256) void bbox_scan_kernel(const float* __restrict__ tp, int b0, SampleMap sm, int H, int W, int chunk,
                                                        int vec_ok, int32_t* __restrict__ box) {
    const int b = blockIdx.y;            // box / yxhw rows are local to the launch; the sample map uses the global index
    const size_t plane = (size_t)H * W;
    const int bg = b0 + b;
    const float* p = tp + (size_t)(bg / sm.n_frames) * sm.stride_obj + (size_t)(bg % sm.n_frames) * sm.stride_frame;
    const size_t beg = (size_t)blockIdx.x * chunk;
    const size_t end = min(plane, beg + (size_t)chunk);
    int ymin = INT_MAX, ymax = -1, xmin = INT_MAX, xmax = -1;
    if (vec_ok) {  // plane % 4 == 0, chunk % 4 == 0, base 16-B aligned: 16 B per lane, fully coalesced
        // four 16-byte loads per lane in flight, (y, x) carried incrementally (no 64-bit division per load), and the per-element
        // work only for a float4 that holds a foreground pixel at all
        size_t i = beg + (size_t)threadIdx.x * 4;
        int y = (int)(i / W), x = (int)(i - (size_t)y * W);
        const int sy = 1024 / W, sx = 1024 % W;                      // one block stride = 1024 elements
        auto visit = [&](const float4& v, int yy, int xx) {
            if (fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)) > 0.5f) {   // tm = (tp > 0.5); mask >= 0.49 on {0,1} is the same set (assessment.py:165,115)
                const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (e[j] > 0.5f) {
                        ymin = min(ymin, yy); ymax = max(ymax, yy);
                        xmin = min(xmin, xx); xmax = max(xmax, xx);
                    }
                    if (++xx == W) { xx = 0; ++yy; }
                }
            }
        };
        auto step = [&](int& yy, int& xx) { yy += sy; xx += sx; if (xx >= W) { xx -= W; ++yy; } };
        for (; i + 3 * 1024 < end; i += 4 * 1024) {
            const float4 v0 = *reinterpret_cast<const float4*>(p + i), v1 = *reinterpret_cast<const float4*>(p + i + 1024);
            const float4 v2 = *reinterpret_cast<const float4*>(p + i + 2048), v3 = *reinterpret_cast<const float4*>(p + i + 3072);
            visit(v0, y, x); step(y, x);
            visit(v1, y, x); step(y, x);
            visit(v2, y, x); step(y, x);
            visit(v3, y, x); step(y, x);
        }
        for (; i < end; i += 1024) {
            visit(*reinterpret_cast<const float4*>(p + i), y, x);
            step(y, x);
        }
    } else {
        for (size_t i = beg + threadIdx.x; i < end; i += 256) {
            if (p[i] > 0.5f) {
                const int y = (int)(i / W), x = (int)(i - (size_t)y * W);
                ymin = min(ymin, y); ymax = max(ymax, y);
                xmin = min(xmin, x); xmax = max(xmax, x);
            }
        }
    }
    ymin = wave_min(ymin); ymax = wave_max(ymax);
    xmin = wave_min(xmin); xmax = wave_max(xmax);
    __shared__ int red[4][4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { red[wave][0] = ymin; red[wave][1] = ymax; red[wave][2] = xmin; red[wave][3] = xmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            ymin = min(ymin, red[w][0]); ymax = max(ymax, red[w][1]);
            xmin = min(xmin, red[w][2]); xmax = max(xmax, red[w][3]);
        }
        if (ymax >= 0) {
            atomicMin(&box[b * 4 + 0], ymin); atomicMax(&box[b * 4 + 1], ymax);
            atomicMin(&box[b * 4 + 2], xmin); atomicMax(&box[b * 4 + 3], xmax);
        }
    }
}

// integer box -> (y,x,h,w) fp32 with the reference's mixed int/float64 arithmetic (assessment.py:118-157)
__global__ void bbox_finalize_kernel(const int32_t* __restrict__ box, int B, int H, int W, float* __restrict__ yxhw) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int y0 = box[b * 4 + 0], y1 = box[b * 4 + 1], x0 = box[b * 4 + 2], x1 = box[b * 4 + 3];
    if (y1 < 0) { y0 = 0; y1 = H; x0 = 0; x1 = W; }  // empty mask: whole frame, H and W (not H-1, W-1)
    if (y1 - y0 < 128) { const int half = (int)((128.0 - (double)(y1 - y0)) / 2.0); y0 -= half; y1 += half; }
    if (x1 - x0 < 128) { const int half = (int)((128.0 - (double)(x1 - x0)) / 2.0); x0 -= half; x1 += half; }
    const double oh = (double)(y1 - y0 + 1), ow = (double)(x1 - x0 + 1);
    const double k = (1.5 - 1.0) / 2.0;
    const double fy0 = fmax(-5.0, (double)y0 - k * oh), fy1 = fmin((double)H + 5.0, (double)y1 + k * oh);
    const double fx0 = fmax(-5.0, (double)x0 - k * ow), fx1 = fmin((double)W + 5.0, (double)x1 + k * ow);
    yxhw[b * 4 + 0] = (float)((fy1 + fy0) / 2.0);
    yxhw[b * 4 + 1] = (float)((fx1 + fx0) / 2.0);
    yxhw[b * 4 + 2] = (float)(fy1 - fy0 + 1.0);
    yxhw[b * 4 + 3] = (float)(fx1 - fx0 + 1.0);
}

void launch_mask_bbox(const float* tp, int b0, int B, int H, int W, const SampleMap& sm, float* yxhw, int32_t* scratch, hipStream_t st) {
    hipLaunchKernelGGL(bbox_init_kernel, dim3((B + 63) / 64), dim3(64), 0, st, scratch, B);
    const size_t plane = (size_t)H * W;
    int S = (int)max((size_t)1, min((size_t)64, (size_t)2048 / (size_t)B));
    size_t chunk = (plane + S - 1) / S;
    chunk = (chunk + 1023) / 1024 * 1024;  // multiple of 4 (and of the 1024-element block stride)
    S = (int)((plane + chunk - 1) / chunk);
    const int vec_ok = (plane % 4 == 0) && ((reinterpret_cast<uintptr_t>(tp) & 15) == 0) && sm.stride_frame % 4 == 0 && sm.stride_obj % 4 == 0;
    hipLaunchKernelGGL(bbox_scan_kernel, dim3(S, B), dim3(256), 0, st, tp, b0, sm, H, W, (int)chunk, vec_ok, scratch);
    hipLaunchKernelGGL(bbox_finalize_kernel, dim3((B + 63) / 64), dim3(64), 0, st, scratch, B, H, W, yxhw);
}

// ---------------------------------------------------------------- ROI sampler
__device__ __forceinline__ float lin_m1_1(int j, int n) {  // torch.linspace(-1,1,n)[j] in fp32
    const float step = 2.0f / (float)(n - 1);
    return (j < n / 2) ? __fadd_rn(-1.0f, __fmul_rn((float)j, step)) : __fsub_rn(1.0f, __fmul_rn((float)(n - 1 - j), step));
}

// grid B * 256 / ROI_R, block 256: thread j owns output column j of ROI_R consecutive output rows of one sample,
// roi[b][i][j][0..3] = (R,G,B normalised, P).  Everything that depends on the sample and the column only (theta, the source
// x, its two taps, their weights and validity) is computed once per thread instead of once per output pixel, and the rows are
// processed two at a time (32 gathers in flight): one output pixel per thread ran ~250 vector instructions for 16 loads and
// 8 B of output, VALU-bound at 180 us per 256 frames.  Per pixel the arithmetic and its order are unchanged.
constexpr int ROI_R = 8;
// two neighbouring fp32 pixels with one 8-byte load from a 4-byte-aligned address (global memory takes unaligned dwordx2)
struct __attribute__((packed, aligned(4))) PairF32 { float x, y; };
__device__ __forceinline__ float2 ld_pair(const float* p) {
    const PairF32 v = *reinterpret_cast<const PairF32*>(p);
    return make_float2(v.x, v.y);
}

template <typename T>
__global__ __launch_bounds__(256) void roi_sample_kernel(const float* __restrict__ tf, const float* __restrict__ tp,
                                                         const float* __restrict__ yxhw, int b0, SampleMap sm, int H, int W, RoiNorm nrm,
                                                         T* __restrict__ roi) {
    constexpr int GPS = 256 / ROI_R;                               // row groups per sample
    const int b = blockIdx.x / GPS, i0 = (blockIdx.x % GPS) * ROI_R, j = threadIdx.x;     // yxhw / roi rows are local to the launch
    const int bg = b0 + b, fr = bg % sm.n_frames;
    const float ry = yxhw[b * 4 + 0], rx = yxhw[b * 4 + 1], rh = yxhw[b * 4 + 2], rw = yxhw[b * 4 + 3];
    // get_ROI_grid (assessment.py:79-92), fp32, same operation order as the reference.  NOTE: HIP's __fmul_rn / __fadd_rn
    // are plain operators inside header functions, so hipcc still contracts mul+add pairs into FMAs here (as the GEMM
    // behind torch's affine_grid does); the golden-slice tests bound the resulting sample-point error (<= 3e-4 on the
    // ROI tiles, 8e-7 on the fp32 scores).  seg_epilogue.hip shows the pragma that really switches contraction off.
    const float ymin = __fsub_rn(ry, rh / 2.0f), ymax = __fadd_rn(ry, rh / 2.0f);
    const float xmin = __fsub_rn(rx, rw / 2.0f), xmax = __fadd_rn(rx, rw / 2.0f);
    const float wm = (float)(W - 1), hm = (float)(H - 1);
    const float t00 = __fsub_rn(xmax, xmin) / wm, t02 = __fsub_rn(__fadd_rn(xmin, xmax), wm) / wm;
    const float t11 = __fsub_rn(ymax, ymin) / hm, t12 = __fsub_rn(__fadd_rn(ymin, ymax), hm) / hm;
    // affine_grid + grid_sample(align_corners=True): pixel = ((g + 1) / 2) * (size - 1)
    const float gx = __fadd_rn(__fmul_rn(lin_m1_1(j, 256), t00), t02);
    const float sx = __fmul_rn(__fmul_rn(__fadd_rn(gx, 1.0f), 0.5f), wm);
    const float x0f = floorf(sx);
    const int x0 = (int)x0f;
    const float wx1 = __fsub_rn(sx, x0f), wx0 = __fsub_rn(__fadd_rn(x0f, 1.0f), sx);
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x0 + 1 >= 0 && x0 + 1 < W;
    // All 16 taps are loaded unconditionally from clamped addresses and out-of-range taps get weight 0 (zero padding:
    // they contribute +0, which leaves the fp32 sum bit-identical): predicated loads made hipcc branch around every
    // load and wait for it, 16 dependent round trips per pixel.
    // The two x taps of a row are neighbours in memory: ONE 8-byte load at xb = clamp(x0, 0, W - 2) brings both (the texture-
    // cache access count, ~19 per gathered dword load of a wave, bounds this kernel - rocprofv3 TCP_TOTAL_CACHE_ACCESSES).
    // At the frame's edges one of the taps has weight 0 and takes whichever in-range pixel the pair holds.
    const int xb = min(max(x0, 0), W - 2);
    const bool t0_hi = x0 >= W - 1, t1_lo = x0 < 0;                // tap0 = pair.y at the right edge, tap1 = pair.x at the left edge
    const size_t plane = (size_t)H * W;
    const float* src[4];
#pragma unroll
    for (int c = 0; c < 3; ++c) src[c] = tf + ((size_t)fr * 3 + c) * plane;
    src[3] = tp + (size_t)(bg / sm.n_frames) * sm.stride_obj + (size_t)fr * sm.stride_frame;
    float mu[3], sd[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {      // (f - mean) / std after the zero pad (assessment.py:47)
        mu[c] = nrm.dev ? nrm.dev[c] : nrm.mean[c];
        sd[c] = nrm.dev ? nrm.dev[3 + c] : nrm.std[c];
    }
    T* dst = roi + (((size_t)b * 256 + i0) * 256 + j) * 4;
#pragma unroll 1
    for (int r0 = 0; r0 < ROI_R; r0 += 2) {
        float v[2][4][4], k[2][4];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int i = i0 + r0 + rr;
            const float gy = __fadd_rn(__fmul_rn(lin_m1_1(i, 256), t11), t12);
            const float sy = __fmul_rn(__fmul_rn(__fadd_rn(gy, 1.0f), 0.5f), hm);
            const float y0f = floorf(sy);
            const int y0 = (int)y0f;
            const float wy1 = __fsub_rn(sy, y0f), wy0 = __fsub_rn(__fadd_rn(y0f, 1.0f), sy);
            const bool vy0 = y0 >= 0 && y0 < H, vy1 = y0 + 1 >= 0 && y0 + 1 < H;
            const int yc0 = min(max(y0, 0), H - 1), yc1 = min(max(y0 + 1, 0), H - 1);
            const int o0 = yc0 * W + xb, o1 = yc1 * W + xb;        // < H * W <= INT_MAX (checked by the host)
            const float w_nw = __fmul_rn(wx0, wy0), w_ne = __fmul_rn(wx1, wy0), w_sw = __fmul_rn(wx0, wy1), w_se = __fmul_rn(wx1, wy1);
            k[rr][0] = (vy0 && vx0) ? w_nw : 0.f; k[rr][1] = (vy0 && vx1) ? w_ne : 0.f;
            k[rr][2] = (vy1 && vx0) ? w_sw : 0.f; k[rr][3] = (vy1 && vx1) ? w_se : 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float2 p0 = ld_pair(src[c] + o0), p1 = ld_pair(src[c] + o1);
                v[rr][c][0] = t0_hi ? p0.y : p0.x; v[rr][c][1] = t1_lo ? p0.x : p0.y;
                v[rr][c][2] = t0_hi ? p1.y : p1.x; v[rr][c][3] = t1_lo ? p1.x : p1.y;
            }
        }
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            float out[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float acc = 0.f;
                acc = __fadd_rn(acc, __fmul_rn(v[rr][c][0], k[rr][0]));
                acc = __fadd_rn(acc, __fmul_rn(v[rr][c][1], k[rr][1]));
                acc = __fadd_rn(acc, __fmul_rn(v[rr][c][2], k[rr][2]));
                acc = __fadd_rn(acc, __fmul_rn(v[rr][c][3], k[rr][3]));
                if (c < 3) acc = __fsub_rn(acc, mu[c]) / sd[c];
                out[c] = acc;
            }
            T* d = dst + (size_t)(r0 + rr) * 256 * 4;
            if constexpr (sizeof(T) == 4) {
                *reinterpret_cast<float4*>(d) = make_float4(out[0], out[1], out[2], out[3]);
            } else {
                *reinterpret_cast<uint2*>(d) = make_uint2(pack2_bf16(out[0], out[1]), pack2_bf16(out[2], out[3]));
            }
        }
    }
}

void launch_roi_sample(const float* tf, const float* tp, const float* yxhw, int b0, int B, int H, int W, int dtype,
                       const SampleMap& sm, const RoiNorm& nrm, void* roi, hipStream_t st) {
    if (dtype != IVOSW_BF16)
        hipLaunchKernelGGL(roi_sample_kernel<float>, dim3(B * (256 / ROI_R)), dim3(256), 0, st, tf, tp, yxhw, b0, sm, H, W, nrm,
                           static_cast<float*>(roi));
    else
        hipLaunchKernelGGL(roi_sample_kernel<bf16_t>, dim3(B * (256 / ROI_R)), dim3(256), 0, st, tf, tp, yxhw, b0, sm, H, W, nrm,
                           static_cast<bf16_t*>(roi));
}

}  // namespace ivosw

using namespace ivosw;

extern "C" int ivosw_mask_bbox(const float* tp, int B, int H, int W, float* yxhw, int32_t* scratch,
                               ivosw_stream_t stream) {
    IVOSW_REQUIRE(tp && yxhw && scratch, "null pointer");
    IVOSW_ON_DEVICE_OF(yxhw);
    IVOSW_REQUIRE(B > 0 && H > 0 && W > 0, "B, H, W must be positive");
    launch_mask_bbox(tp, 0, B, H, W, SampleMap{B, (long)H * W, 0}, yxhw, scratch, as_stream(stream));
    IVOSW_CHECK_LAUNCH();
    return IVOSW_OK;
}

extern "C" int ivosw_roi_sample(const float* tf, const float* tp, const float* yxhw, int B, int H, int W, int dtype,
                                void* roi, ivosw_stream_t stream) {
    IVOSW_REQUIRE(tf && tp && yxhw && roi, "null pointer");
    IVOSW_ON_DEVICE_OF(roi);
    IVOSW_REQUIRE(B > 0 && H > 1 && W > 1, "B must be positive and H, W > 1");
    IVOSW_REQUIRE((long)H * W <= INT_MAX && B <= (1 << 24), "frame or batch too large (H * W <= INT_MAX, B <= 2^24)");
    IVOSW_REQUIRE(dtype == IVOSW_F32 || dtype == IVOSW_BF16 || dtype == IVOSW_F32X3, "dtype must be IVOSW_F32, IVOSW_BF16 or IVOSW_F32X3");
    RoiNorm nrm{{0.485f, 0.456f, 0.406f}, {0.229f, 0.224f, 0.225f}, nullptr};  // Encoder.mean/std (assessment.py:41-44)
    launch_roi_sample(tf, tp, yxhw, 0, B, H, W, dtype, SampleMap{B, (long)H * W, 0}, nrm, roi, as_stream(stream));
    IVOSW_CHECK_LAUNCH();
    return IVOSW_OK;
}

// ---------------------------------------------------------------- shader-clock probe (measurement aid, bench.py roofline.sclk_mhz)
// One wave spins for `spin_us` of wall time and reports {shader cycles (s_memtime), 100 MHz wall ticks (s_memrealtime)} of the
// interval: launched on a side stream beside a forward pass it gives the shader clock the chip actually ran at under that load
// (the chip clocks to its power budget: DESIGN section 5).  out: 2 x uint64 on the device.
namespace ivosw {
__global__ void clock_probe_kernel(unsigned long long* __restrict__ out, unsigned long long spin_ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long r0 = wall_clock64();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < spin_ticks) {
        __builtin_amdgcn_s_sleep(32);
        r1 = wall_clock64();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[0] = t1 - t0;
    out[1] = wall_clock64() - r0;
}
}  // namespace ivosw

extern "C" int ivosw_clock_probe(unsigned long long* out2, int spin_us, ivosw_stream_t stream) {
    using namespace ivosw;
    IVOSW_REQUIRE(out2 && spin_us > 0 && spin_us <= 1000000, "null pointer / spin_us out of range");
    IVOSW_ON_DEVICE_OF(out2);
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, as_stream(stream), out2, (unsigned long long)spin_us * 100ull);
    IVOSW_CHECK_LAUNCH();
    return IVOSW_OK;
}
