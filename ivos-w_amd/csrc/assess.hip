// AssessNet.forward on the MI355X: plan of the 54-conv tower, weight packing, chunked execution.
//
// Reference: AssessNet.forward (models/assessment.py:164-182), Encoder (:12-63), torchvision ResNet-50 v1.5.
// Frames run through the tower in chunks (default 256 bf16 / 64 fp32 frames at res2, doubling per stage).  Chunks
// bound the workspace; they are NOT a cache-residency device: an Infinity-Cache-sized working set streams at
// ~7 TB/s against ~6 TB/s from HBM (tools/ubench/dma_bench), while small launches cost occupancy, so the bf16
// default is as large as the benchmark batch.  In bf16 mode the stride-1 bottlenecks of res2 run as ONE fused kernel
// each (bottleneck.hip); everything else is layer by layer (conv.hip).
#ifdef IVOSW_PROBES
#include "../../include/ivosw_probe.h"
#endif
#include <limits.h>
#include <algorithm>
#include <vector>

#include <mutex>

#include "conv.h"
#include "front.h"

namespace ivosw {

struct ConvPlan {
    int Cin, Cout, K, stride, pad;  // K = kernel size (1 or 3)
    int t_w, t_bn;                  // state_dict indices: conv weight; bn weight (bias, mean, var follow)
    size_t w_off, b_off;            // byte offsets into the packed arena
    size_t fw_off;                  // bf16: 1x1 weights in MFMA-operand order for conv1x1_wide_kernel (0 = none)
};

struct BlockPlan {
    int c1, c2, c3, ds;  // indices into convs (ds = -1 if none)
    size_t cat_fw_off;            // bf16: the concatenated weights in MFMA-operand order
    size_t cat_w_off, cat_b_off;  // first blocks: conv3 | downsample concatenated along K, bias sum (conv3 absorbs the downsample conv)
    size_t f1_off, f2_off, f3_off;  // bf16, identity blocks: conv1/2/3 weights in MFMA-operand order (0 = none)
    size_t fwd1_off;                // bf16, res3's first block: conv1 in MFMA-operand order for res2's last block (conv1 forwarding)
};

struct Plan {
    std::vector<ConvPlan> convs;
    std::vector<BlockPlan> blocks;
    size_t norm_off, zero_off, fcw_off, fcb_off, stem_w_off, stem_b_off, total;
    size_t chain_off = 0;           // bf16: res2's weight-fragment stream for res2_chain_kernel (0 = none)
    int t_stem_w3, t_stem_w1, t_stem_bn, t_fcw, t_fcb, t_mean, t_std;
};

static Plan make_plan(int dtype) {
    Plan P;
    const size_t es = (dtype == IVOSW_BF16) ? 2 : 4;
    size_t off = 0;
    auto take = [&](size_t bytes) { off = align_up(off, 256); const size_t o = off; off += bytes; return o; };
    P.norm_off = take(8 * sizeof(float));
    P.zero_off = take(256);  // page of zeros: source of zero-padding taps for the LDS-DMA conv loader
    P.fcw_off = take(2048 * sizeof(float));
    P.fcb_off = take(sizeof(float));
    P.stem_w_off = take((size_t)64 * ((dtype == IVOSW_BF16) ? 256 : 224) * es);
    P.stem_b_off = take(64 * sizeof(float));
    // state_dict order: 0 mean, 1 std, 2 conv1_m.w, 3 conv1_m.b, 4 conv1_p.w, 5 conv1_n.w,
    // 6 conv1.w, 7..11 bn1.{w,b,rm,rv,nbt}, then per block conv1.w, bn1x5, conv2.w, bn2x5, conv3.w, bn3x5,
    // [downsample.0.w, downsample.1x5], finally fc1.w, fc1.b
    P.t_mean = 0; P.t_std = 1; P.t_stem_w1 = 4; P.t_stem_w3 = 6; P.t_stem_bn = 7;
    int t = 12;
    const int nblk[4] = {3, 4, 6, 3}, planes[4] = {64, 128, 256, 512}, strides[4] = {1, 2, 2, 2};
    int inpl = 64;
    auto add = [&](int cin, int cout, int k, int stride) {
        ConvPlan c{};
        c.Cin = cin; c.Cout = cout; c.K = k; c.stride = stride; c.pad = (k == 3) ? 1 : 0;
        c.t_w = t; c.t_bn = t + 1; t += 6;
        c.w_off = take((size_t)cout * k * k * cin * es);
        c.b_off = take((size_t)cout * sizeof(float));
        if (dtype == IVOSW_BF16 && k == 1 && cout % 256 == 0 && cin >= 384) c.fw_off = take((size_t)cout * cin * es);
        P.convs.push_back(c);
        return (int)P.convs.size() - 1;
    };
    for (int s = 0; s < 4; ++s)
        for (int b = 0; b < nblk[s]; ++b) {
            const int st = (b == 0) ? strides[s] : 1;
            BlockPlan bp{};
            bp.c1 = add(inpl, planes[s], 1, 1);
            bp.c2 = add(planes[s], planes[s], 3, st);
            bp.c3 = add(planes[s], planes[s] * 4, 1, 1);
            bp.ds = (b == 0) ? add(inpl, planes[s] * 4, 1, st) : -1;
            if (b == 0) {
                bp.cat_w_off = take((size_t)planes[s] * 4 * (planes[s] + inpl) * es);
                bp.cat_b_off = take((size_t)planes[s] * 4 * sizeof(float));
                if (dtype == IVOSW_BF16 && (planes[s] + inpl >= 384 || s == 0)) bp.cat_fw_off = take((size_t)planes[s] * 4 * (planes[s] + inpl) * es);
                if (dtype == IVOSW_BF16 && s == 0) {     // res2's first block (stride 1) is fused as a whole, too
                    bp.f1_off = take((size_t)planes[s] * inpl * es);
                    bp.f2_off = take((size_t)planes[s] * 9 * planes[s] * es);
                }
                if (dtype == IVOSW_BF16 && s == 1) {     // res3's first 1x1 runs inside res2's last block (conv1 forwarding)
                    bp.fwd1_off = take((size_t)planes[s] * inpl * es);
                    bp.f2_off = take((size_t)planes[s] * 9 * planes[s] * es);    // ... and the rest of the block in stage_first_kernel
                }
            } else if (dtype == IVOSW_BF16) {
                bp.f1_off = take((size_t)planes[s] * inpl * es);
                bp.f2_off = take((size_t)planes[s] * 9 * planes[s] * es);
                bp.f3_off = take((size_t)planes[s] * 4 * planes[s] * es);
            }
            P.blocks.push_back(bp);
            inpl = planes[s] * 4;
        }
    P.t_fcw = t; P.t_fcb = t + 1;
    if (dtype == IVOSW_BF16) P.chain_off = take(res2_chain_stream_bytes());
    P.total = align_up(off, 256);
    return P;
}

// weights of the res2 stage kernel (res2_stage.hip) out of the packed bf16 arena
// A fragment copy the plan does not hold has offset 0: its pointer stays NULL (base + 0 would silently be the first weights of the
// arena), which is what res2_stage_ok() looks at before the per-block kernels / the probe's error take over.
static Res2StageArgs res2_stage_args(const Plan& P, const char* base) {
    Res2StageArgs q{};
    auto at = [&](size_t off) -> const void* { return off ? base + off : nullptr; };
    q.zeros = at(P.zero_off);
    for (int b = 0; b < 3; ++b) {
        const BlockPlan& bp = P.blocks[b];
        const BlockPlan& nx = P.blocks[b + 1];
        q.fb[b] = at(bp.f2_off); q.bb[b] = reinterpret_cast<const float*>(base + P.convs[bp.c2].b_off);
        q.fc[b] = at(b == 0 ? bp.cat_fw_off : bp.f3_off);
        q.bc[b] = static_cast<const float*>(b == 0 ? at(bp.cat_b_off) : base + P.convs[bp.c3].b_off);
        q.fd[b] = at(b == 2 ? nx.fwd1_off : nx.f1_off); q.bd[b] = reinterpret_cast<const float*>(base + P.convs[nx.c1].b_off);
    }
    q.fa0 = at(P.blocks[0].f1_off); q.ba0 = reinterpret_cast<const float*>(base + P.convs[P.blocks[0].c1].b_off);
    return q;
}

static const Plan& plan_for(int dtype) {
    static const Plan pb = make_plan(IVOSW_BF16), pf = make_plan(IVOSW_F32);
    return dtype == IVOSW_BF16 ? pb : pf;
}

__global__ void copy_small_kernel(const float* __restrict__ a, int na, const float* __restrict__ b, int nb, float* __restrict__ o) {
    const int i = threadIdx.x;
    if (i < na) o[i] = a[i];
    if (i < nb) o[na + i] = b[i];
}

// per-frame element counts
constexpr size_t E_ROI = 256 * 256 * 4, E_BIG = 64 * 64 * 256;
// per-frame output elements of res2..res5
constexpr size_t E_OUT[4] = {64 * 64 * 256, 32 * 32 * 512, 16 * 16 * 1024, 8 * 8 * 2048};

// Chunk schedule.  Stage s (res2..res5) runs on c0 * 2^s frames at a time: every stage halves H and W and
// doubles C, so doubling the frames per launch keeps the per-stage tensor size constant (c0 * 2 MB bf16) and every
// conv launch keeps >= 512 workgroups for the 256 CUs.
// fp32 modes: 16 -> 64 frames per res2 launch = 6.8 k -> 9.9 k frames/s; 128 (round 6): fp32 9.96 k, the three-pass mode 21.32 -> 21.57 k (two alternating rounds), at
// twice the workspace (11.5 instead of 5.75 GiB at batch 256 - of 288 GB)
static int default_chunk(int dtype) { return dtype == IVOSW_BF16 ? 256 : std::max(1, tune_get("F32_CHUNK", 128)); }

struct Bufs {
    float* yxhw; int32_t* box; float* pooled;
    char *roi, *stem, *pa, *pb, *m1, *m2, *ds;
    char* in[4];  // in[s]: input of stage s+1 = output of stage s, for the frames of the enclosing chunk (s = 0..2)
};

static size_t carve(Arena& ar, int dtype, int B, int c0, Bufs* out) {
    const size_t es = (dtype == IVOSW_BF16) ? 2 : 4;
    Bufs b{};
    const size_t c3 = (size_t)std::min(B, c0 * 8);
    b.yxhw = ar.take<float>((size_t)B * 4);
    b.box = ar.take<int32_t>((size_t)B * 4);
    b.pooled = ar.take<float>(c3 * 2048);
    b.roi = ar.take<char>(c0 * E_ROI * es);
    b.stem = ar.take<char>(c0 * E_BIG * es);
    b.pa = ar.take<char>(c0 * E_BIG * es);
    b.pb = ar.take<char>(c0 * E_BIG * es);
    b.ds = ar.take<char>(c0 * E_BIG * es);
    b.m1 = ar.take<char>(c0 * E_BIG * es);
    b.m2 = ar.take<char>(c0 * E_BIG / 4 * es);
    for (int s = 0; s < 3; ++s) b.in[s] = ar.take<char>((size_t)c0 * 2 * E_BIG * es);
    b.in[3] = nullptr;
    if (out) *out = b;
    return align_up(ar.off, 256);
}

}  // namespace ivosw

using namespace ivosw;

// Which precision an arena was packed for (IVOSW_F32 and IVOSW_F32X3 share plan, sizes and layout, but the x3 pack
// rewrites every weight K-tile in place as [hi | lo] bf16 - a forward call with the other dtype passed every check and returned
// garbage).  The tag lives in two places: float slot 6 of the arena's 8-float normalisation block (for whoever inspects or copies
// an arena) and a host-side table keyed by the arena's address, which the forward entry points check without touching the device.
static std::mutex g_pack_mu;
static std::vector<std::pair<const void*, int>> g_pack_dtype;
static void pack_dtype_record(const void* packed, int dtype) {
    std::lock_guard<std::mutex> lock(g_pack_mu);
    for (auto& e : g_pack_dtype)
        if (e.first == packed) { e.second = dtype; return; }
    g_pack_dtype.emplace_back(packed, dtype);
}
// The check covers arenas whose tag this process knows: packed here (recorded at pack time) or seen before.  An address the table does not
// hold - an arena filled by a device copy, or through the C ABI from another library instance - is looked up ONCE in the arena itself (the
// 4-byte device tag, a blocking copy on the caller's stream: first use only) and cached.  -1: no valid tag, not checkable.
// ivosw_assess_forget() drops an address when its memory is freed or reused for an arena of another precision by a device copy.
static int pack_dtype_lookup(const void* packed, hipStream_t st = nullptr) {
    {
        std::lock_guard<std::mutex> lock(g_pack_mu);
        for (auto& e : g_pack_dtype)
            if (e.first == packed) return e.second;
    }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (st && hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return -1;      // no blocking copy inside a capture
    unsigned tag = 0;
    const char* slot = static_cast<const char*>(packed) + plan_for(IVOSW_BF16).norm_off + 6 * sizeof(float);     // (norm_off is the same in every plan)
    if (hipMemcpyAsync(&tag, slot, sizeof(tag), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { (void)hipGetLastError(); return -1; }
    const int dtype = (tag >> 16) == 0x4956u && (int)(tag & 0xffffu) <= IVOSW_F32X3 ? (int)(tag & 0xffffu) : -1;
    if (dtype >= 0) pack_dtype_record(packed, dtype);
    return dtype;
}
extern "C" int ivosw_assess_forget(const void* packed) {
    std::lock_guard<std::mutex> lock(g_pack_mu);
    for (size_t i = 0; i < g_pack_dtype.size(); ++i)
        if (g_pack_dtype[i].first == packed) { g_pack_dtype.erase(g_pack_dtype.begin() + i); return IVOSW_OK; }
    return IVOSW_OK;
}

extern "C" size_t ivosw_assess_packed_bytes(int dtype) {
    if (dtype != IVOSW_F32 && dtype != IVOSW_BF16 && dtype != IVOSW_F32X3) return 0;
    return plan_for(dtype).total;
}

extern "C" int ivosw_assess_pack(void* packed, int dtype, const void* const* tensors, int ntensors, ivosw_stream_t stream) {
    IVOSW_REQUIRE(packed && tensors, "null pointer");
    IVOSW_ON_DEVICE_OF(packed);
    IVOSW_REQUIRE(dtype == IVOSW_F32 || dtype == IVOSW_BF16 || dtype == IVOSW_F32X3, "dtype must be IVOSW_F32, IVOSW_BF16 or IVOSW_F32X3");
    IVOSW_REQUIRE(ntensors == IVOSW_ASSESS_NTENSORS, "expected the 326 tensors of AssessNet.state_dict()");
    const Plan& P = plan_for(dtype);
    IVOSW_REQUIRE(P.t_fcb == IVOSW_ASSESS_NTENSORS - 1, "internal: plan/state_dict mismatch");
    hipStream_t st = as_stream(stream);
    char* base = static_cast<char*>(packed);
    auto T = [&](int i) { return static_cast<const float*>(tensors[i]); };
    for (int i : {P.t_mean, P.t_std, P.t_stem_w1, P.t_stem_w3, P.t_fcw, P.t_fcb}) IVOSW_REQUIRE(tensors[i], "null tensor");
    (void)hipMemsetAsync(base + P.zero_off, 0, 256, st);
    hipLaunchKernelGGL(copy_small_kernel, dim3(1), dim3(64), 0, st, T(P.t_mean), 3, T(P.t_std), 3,
                       reinterpret_cast<float*>(base + P.norm_off));
    (void)hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(base + P.norm_off + 6 * sizeof(float)), 0x49560000 | dtype, 1, st);   // 'IV' | dtype
    pack_dtype_record(packed, dtype);
    (void)hipMemcpyAsync(base + P.fcw_off, T(P.t_fcw), 2048 * sizeof(float), hipMemcpyDeviceToDevice, st);
    (void)hipMemcpyAsync(base + P.fcb_off, T(P.t_fcb), sizeof(float), hipMemcpyDeviceToDevice, st);
    launch_pack_stem(T(P.t_stem_w3), T(P.t_stem_w1), T(P.t_stem_bn), T(P.t_stem_bn + 1), T(P.t_stem_bn + 2),
                     T(P.t_stem_bn + 3), dtype, base + P.stem_w_off, reinterpret_cast<float*>(base + P.stem_b_off), st);
    for (const ConvPlan& c : P.convs) {
        for (int j = 0; j < 4; ++j) IVOSW_REQUIRE(tensors[c.t_bn + j] && tensors[c.t_w], "null tensor");
        launch_pack_conv(T(c.t_w), T(c.t_bn), T(c.t_bn + 1), T(c.t_bn + 2), T(c.t_bn + 3), c.Cout, c.Cin, c.K, c.K, dtype,
                         base + c.w_off, reinterpret_cast<float*>(base + c.b_off), st);
    }
    for (const ConvPlan& c : P.convs)
        if (c.fw_off) launch_fragpack(base + c.w_off, c.Cout, c.Cin, base + c.fw_off, st);
    for (const BlockPlan& bp : P.blocks)
        if (bp.f1_off) {
            const ConvPlan &c1 = P.convs[bp.c1], &c2 = P.convs[bp.c2], &c3 = P.convs[bp.c3];
            launch_fragpack(base + c1.w_off, c1.Cout, c1.Cin, base + bp.f1_off, st);
            launch_fragpack(base + c2.w_off, c2.Cout, 9 * c2.Cin, base + bp.f2_off, st);
            if (bp.f3_off) launch_fragpack(base + c3.w_off, c3.Cout, c3.Cin, base + bp.f3_off, st);
        }
    for (const BlockPlan& bp : P.blocks)
        if (bp.fwd1_off) {
            launch_fragpack(base + P.convs[bp.c1].w_off, P.convs[bp.c1].Cout, P.convs[bp.c1].Cin, base + bp.fwd1_off, st);
            if (bp.f2_off && !bp.f1_off) launch_fragpack(base + P.convs[bp.c2].w_off, P.convs[bp.c2].Cout, 9 * P.convs[bp.c2].Cin, base + bp.f2_off, st);
        }
    for (const BlockPlan& bp : P.blocks)
        if (bp.ds >= 0) {
            const ConvPlan &c3 = P.convs[bp.c3], &cd = P.convs[bp.ds];
            launch_concat_k(base + c3.w_off, reinterpret_cast<const float*>(base + c3.b_off), c3.Cin, base + cd.w_off,
                            reinterpret_cast<const float*>(base + cd.b_off), cd.Cin, c3.Cout, dtype, base + bp.cat_w_off,
                            reinterpret_cast<float*>(base + bp.cat_b_off), st);
            if (bp.cat_fw_off) launch_fragpack(base + bp.cat_w_off, c3.Cout, c3.Cin + cd.Cin, base + bp.cat_fw_off, st);
        }
    if (P.chain_off) {
        Res2ChainPackArgs q{};
        auto W = [&](int ci) { return reinterpret_cast<const bf16_t*>(base + P.convs[ci].w_off); };
        auto Bi = [&](int ci) { return reinterpret_cast<const float*>(base + P.convs[ci].b_off); };
        for (int b = 0; b < 4; ++b) { q.w1[b] = W(P.blocks[b].c1); q.b1[b] = Bi(P.blocks[b].c1); }
        for (int b = 0; b < 3; ++b) { q.w2[b] = W(P.blocks[b].c2); q.b2[b] = Bi(P.blocks[b].c2); q.w3[b] = W(P.blocks[b].c3); q.b3[b] = Bi(P.blocks[b].c3); }
        q.wd = W(P.blocks[0].ds); q.bd = Bi(P.blocks[0].ds);
        q.out = base + P.chain_off;
        launch_res2_chain_pack(q, st);
    }
    if (dtype == IVOSW_F32X3) {
        // the three-pass mode reads pre-split weights (conv.hip: ktile_mma_x3): every conv's K-major array and the concatenated
        // [conv3 | downsample] arrays, in place, AFTER everything that read them as fp32 (the concatenation above)
        launch_split_weights_x3(base + P.stem_w_off, 64, 224, st);       // the stem: 7 filter rows x 8 pixels x 4 channels per output channel
        for (const ConvPlan& c : P.convs) launch_split_weights_x3(base + c.w_off, c.Cout, c.K * c.K * c.Cin, st);
        for (const BlockPlan& bp : P.blocks)
            if (bp.ds >= 0) launch_split_weights_x3(base + bp.cat_w_off, P.convs[bp.c3].Cout, P.convs[bp.c3].Cin + P.convs[bp.ds].Cin, st);
    }
    IVOSW_CHECK_LAUNCH();
    return IVOSW_OK;
}

static size_t ws_single(int dtype, int B, int chunk) {
    if (chunk <= 0) chunk = default_chunk(dtype);
    if (chunk > B) chunk = B;
    Arena ar(nullptr);
    return carve(ar, dtype, B, chunk, nullptr) + 256;
}
// Two-stream split (tunable STREAMS2, default on; fp32 too - STREAMS2_F32, 9.84 k -> 10.2 k frames/s): a default-chunk batch of >= STREAMS2_MIN (64) units runs as two
// independent halves, the second on the library's side stream — the fabric-bound phases of one half's kernels meet the
// MFMA-bound phases of the other's (2 x 128 frames: 63.2 k -> 65.8 k frames/s on one box, tools/two_stream_probe.py; four
// quarters are slower; at eval sizes: B = 100 + 7 %, 160 + 13 %, 300 + 12 %, 64 + 0.3 %, tools/split_min_sweep.sh).  Per-frame results do not depend on the batch a frame travels in (tests: chunk schedules, batch
// permutations, B = 256 against B = 8 / 64), so the split is invisible in the scores.  Each half gets its own workspace.
static int split_first_half(int B) { return (B / 2 + 7) / 8 * 8; }
static bool split_wanted(int dtype, int B, int chunk, int tap_stage) {
    // both halves must be non-empty whatever STREAMS2_MIN is tuned to: ceil8(B / 2) reaches B for B <= 8
    return (dtype == IVOSW_BF16 || tune_get("STREAMS2_F32", 1)) && chunk <= 0 && tap_stage == 0 && tune_get("STREAMS2", 1) &&
           B >= std::max(2, tune_get("STREAMS2_MIN", 64)) && split_first_half(B) < B;
}
static size_t ws_split(int dtype, int B) {
    const int B0 = split_first_half(B);
    return align_up(ws_single(dtype, B0, 0), 256) + ws_single(dtype, B - B0, 0);
}

extern "C" size_t ivosw_assess_ws_bytes(int dtype, int B, int H, int W, int chunk) {
    if ((dtype != IVOSW_F32 && dtype != IVOSW_BF16 && dtype != IVOSW_F32X3) || B <= 0 || H <= 0 || W <= 0) return 0;
    const size_t one = ws_single(dtype, B, chunk);
    return split_wanted(dtype, B, chunk, 0) ? std::max(one, ws_split(dtype, B)) : one;
}

extern "C" int ivosw_assess_split(int dtype, int B, int chunk) { return split_wanted(dtype, B, chunk, 0) ? 1 : 0; }

extern "C" const char* ivosw_assess_dominant_kernel(int dtype) {
    return dtype == IVOSW_BF16 ? "conv_igemm*|conv1x1_wide*|conv3x3_patch*|bneck*|res2_stage*|res2_chain_kernel*|gemm_8phase*|stage_first*|stem_pool*" : dtype == IVOSW_F32X3 ? "conv_igemm*|stem_pool_x3*" : "conv_igemm*";   // the tower's contraction kernels (one family)
}

static int assess_forward_impl(const void* packed, int dtype, const float* tf, const float* tp, const SampleMap& sm, int B, int H, int W,
                               float* scores, void* ws, size_t ws_bytes, int chunk, int tap_stage, void* tap_out,
                               ivosw_stream_t stream);

extern "C" int ivosw_assess_forward(const void* packed, int dtype, const float* tf, const float* tp, int B, int H, int W,
                                    float* scores, void* ws, size_t ws_bytes, int chunk, int tap_stage, void* tap_out,
                                    ivosw_stream_t stream) {
    return assess_forward_impl(packed, dtype, tf, tp, SampleMap{B, (long)H * W, 0}, B, H, W, scores, ws, ws_bytes, chunk, tap_stage,
                               tap_out, stream);
}

extern "C" int ivosw_assess_forward_objects(const void* packed, int dtype, const float* tf, int n_frames, const float* masks,
                                            long mask_stride_frame, long mask_stride_obj, int n_obj, int H, int W, float* scores,
                                            void* ws, size_t ws_bytes, int chunk, ivosw_stream_t stream) {
    IVOSW_REQUIRE(n_frames > 0 && n_obj > 0, "n_frames and n_obj must be positive");
    IVOSW_REQUIRE((long)n_frames * n_obj < (1L << 30), "too many (frame, object) units for one call");
    IVOSW_REQUIRE(mask_stride_frame >= (long)H * W || n_frames == 1, "mask planes of consecutive frames overlap");
    IVOSW_REQUIRE(mask_stride_obj >= 0 && mask_stride_frame >= 0, "negative mask stride");
    return assess_forward_impl(packed, dtype, tf, masks, SampleMap{n_frames, mask_stride_frame, mask_stride_obj}, n_frames * n_obj, H,
                               W, scores, ws, ws_bytes, chunk, 0, nullptr, stream);
}

// units [u0, u0 + B) of the batch on stream st with their own workspace; `slot` = which of the (up to two) concurrent profiler spans
static void assess_forward_range(const void* packed, int dtype, const float* tf, const float* tp, const SampleMap& sm, int u0, int B,
                                 int H, int W, float* scores, void* ws, int chunk, int tap_stage, void* tap_out, int slot, hipStream_t st);

// One helper stream + two events per device for the two-stream split (created on first use, all-or-nothing; the call holds
// the device's mutex while it enqueues so two host threads cannot interleave their fork / join pairs).
namespace {
struct Side2 {
    std::mutex mu;
    hipStream_t stream = nullptr;
    hipEvent_t ev[2] = {};
    bool tried = false;
};
Side2* side2_for_current_device() {
    static Side2 sides[64];
    static std::mutex init_mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    Side2& s = sides[dev];
    std::lock_guard<std::mutex> lk(init_mu);
    if (!s.tried) {
        s.tried = true;
        bool ok = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) == hipSuccess;
        int made = 0;
        for (; ok && made < 2; ++made) ok = hipEventCreateWithFlags(&s.ev[made], hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            for (int i = 0; i < made; ++i) if (s.ev[i]) (void)hipEventDestroy(s.ev[i]);
            if (s.stream) (void)hipStreamDestroy(s.stream);
            s.stream = nullptr;
            (void)hipGetLastError();
        }
    }
    return s.stream ? &s : nullptr;
}
}  // namespace

static int assess_forward_impl(const void* packed, int dtype, const float* tf, const float* tp, const SampleMap& sm, int B, int H, int W,
                               float* scores, void* ws, size_t ws_bytes, int chunk, int tap_stage, void* tap_out,
                               ivosw_stream_t stream) {
    IVOSW_REQUIRE(packed && tf && tp && scores && ws, "null pointer");
    IVOSW_ON_DEVICE_OF(scores);
    IVOSW_REQUIRE(dtype == IVOSW_F32 || dtype == IVOSW_BF16 || dtype == IVOSW_F32X3, "dtype must be IVOSW_F32, IVOSW_BF16 or IVOSW_F32X3");
    IVOSW_REQUIRE(B > 0 && H > 1 && W > 1, "B must be positive and H, W > 1");
    IVOSW_REQUIRE((long)H * W <= INT_MAX, "frame too large (H * W <= INT_MAX)");
    IVOSW_REQUIRE(tap_stage >= 0 && tap_stage <= 8, "tap_stage out of range");
    IVOSW_REQUIRE(tap_stage == 0 || tap_out, "tap_out is null");
    {
        const int packed_for = pack_dtype_lookup(packed, as_stream(stream));
        IVOSW_REQUIRE(packed_for < 0 || packed_for == dtype, "the arena was packed for another dtype (ivosw_assess_pack's dtype must be the forward call's)");
    }
    const bool want_split = split_wanted(dtype, B, chunk, tap_stage);
    if (chunk <= 0) chunk = default_chunk(dtype);
    if (chunk > B) chunk = B;
    IVOSW_REQUIRE(tap_stage == 0 || B <= chunk, "taps need B <= chunk");
    if (ws_bytes < ws_single(dtype, B, chunk)) {
        set_error("ivosw_assess_forward: workspace %zu < %zu", ws_bytes, ivosw_assess_ws_bytes(dtype, B, H, W, chunk));  // B = units
        return IVOSW_ERR_WS;
    }
    hipStream_t st = as_stream(stream);
    Side2* sd = (want_split && ws_bytes >= ws_split(dtype, B)) ? side2_for_current_device() : nullptr;
    if (sd) {
        std::lock_guard<std::mutex> lk(sd->mu);
        const int B0 = split_first_half(B), B1 = B - B0;
        char* ws1 = static_cast<char*>(ws) + align_up(ws_single(dtype, B0, 0), 256);
        bool ok = hipEventRecord(sd->ev[0], st) == hipSuccess && hipStreamWaitEvent(sd->stream, sd->ev[0], 0) == hipSuccess;
        if (ok) {
            span_group_begin();
            assess_forward_range(packed, dtype, tf, tp, sm, 0, B0, H, W, scores, ws, std::min(default_chunk(dtype), B0), 0, nullptr, 0, st);
            assess_forward_range(packed, dtype, tf, tp, sm, B0, B1, H, W, scores + B0, ws1, std::min(default_chunk(dtype), B1), 0, nullptr, 1, sd->stream);
            span_group_end();
            ok = hipEventRecord(sd->ev[1], sd->stream) == hipSuccess && hipStreamWaitEvent(st, sd->ev[1], 0) == hipSuccess;
            if (!ok) {
                (void)hipGetLastError();
                set_error("ivosw_assess_forward: the join of the two streams failed");
                return IVOSW_ERR_LAUNCH;
            }
            IVOSW_CHECK_LAUNCH();
            return IVOSW_OK;
        }
        (void)hipGetLastError();        // the fork failed before anything was enqueued on the side stream: run on one stream
    }
    assess_forward_range(packed, dtype, tf, tp, sm, 0, B, H, W, scores, ws, chunk, tap_stage, tap_out, 0, st);
    IVOSW_CHECK_LAUNCH();
    return IVOSW_OK;
}

static void assess_forward_range(const void* packed, int dtype, const float* tf, const float* tp, const SampleMap& sm, int u0, int B,
                                 int H, int W, float* scores, void* ws, int chunk, int tap_stage, void* tap_out, int slot, hipStream_t st) {
    const Plan& P = plan_for(dtype);
    const size_t es = (dtype == IVOSW_BF16) ? 2 : 4;
    const char* base = static_cast<const char*>(packed);
    Arena ar(ws);
    Bufs bf{};
    carve(ar, dtype, B, chunk, &bf);
    auto tap = [&](int stage, const void* src, size_t bytes) {
        if (tap_stage != stage) return;
        (void)hipMemcpyAsync(tap_out, src, bytes, hipMemcpyDeviceToDevice, st);
        // IVOSW_F32X3 keeps its activations (stem output .. res5) in the split hi | lo layout: the tap hands out plain fp32
        if (dtype == IVOSW_F32X3 && stage >= 2 && stage <= 7) launch_unsplit_x3(tap_out, bytes / sizeof(float), st);
    };

    // K1/K2: mask -> (y,x,h,w) for the whole batch, on device
    launch_mask_bbox(tp, u0, B, H, W, sm, bf.yxhw, bf.box, st);
    // Encoder.mean/std come from the checkpoint: the sampler reads them from the packed arena
    RoiNorm nrm{{0.f, 0.f, 0.f}, {1.f, 1.f, 1.f}, reinterpret_cast<const float*>(base + P.norm_off)};

    const int nblk[4] = {3, 4, 6, 3}, first_blk[4] = {0, 3, 7, 13}, hw_in[4] = {64, 64, 32, 16};
    const int cs[4] = {chunk, chunk * 2, chunk * 4, chunk * 8};

    // one bottleneck stage (K5) on nb frames: x -> out (the last block writes straight into `out`)
    // conv1 forwarding through res2 (tunable FWD2): block b of res2 also applies block b+1's first 1x1 to its output tile while
    // it is on chip; the last block applies res3's first 1x1, so that layer (a 0.8 GB pass over HBM) is never launched
    const bool fwd2 = dtype == IVOSW_BF16 && tune_get("FWD2", 1) && tune_get("FUSE_WIDE", 1) && tune_get("FUSE_WIDE2", 1) &&
                      tune_get("HALO64S", 1) && tune_get("HALO64S_DS", 1) && P.blocks[3].fwd1_off;
    // ... and res2's last block then writes its output at the even pixels only (compact [nb,32,32,256]): with conv1 forwarded, the
    // only reader left is res3's stride-2 downsample.  Off when an intermediate is tapped (the res2 tap wants the whole tensor).
    const bool ys2 = fwd2 && tap_stage == 0 && tune_get("FUSE_DS", 1) && tune_get("YS2", 1);
    // ... or the whole of res2 in ONE launch (res2_stage.hip, tunable RES2_STAGE, default on): a workgroup carries its 8 x 16 tile
    // through the three blocks, y0 / y1 never reach HBM.  Same summation orders as the per-block kernels: bit-identical stage output, so
    // the choice moves no score.  In time the two are level (frames/s at B = 32 / 64 / 100 / 160 / 256: +0.8 / -1.2 / +1.9 / -0.6 /
    // -0.8 %, alternating runs on one box; 946 against 983 us on one stream); the stage kernel moves 2.55 GB less HBM traffic per
    // 256-frame pass (9.23 against 11.77 GB) in four launches fewer, which is why it is the default.
    const bool stage2 = fwd2 && tune_get("RES2_STAGE", 1);
    // Snake order (tunable SNAKE, default on): consecutive launches of the tower walk their pixel tiles in OPPOSITE directions, so a
    // launch starts with the tiles its producer wrote last - still in the memory-side cache (256 MB for ~128 MB tensors per stream) -
    // instead of the ones written first and long evicted.  Tiles are independent: the order cannot change a result.
    // From SNAKE_MIN (96) frames per stream: below that the tensors fit the cache either way (B = 32 / 100 / 160: +-0 / -0.6 / +0.4 %).
    const bool snake = dtype == IVOSW_BF16 && tune_get("SNAKE", 1) != 0 && B >= tune_get("SNAKE_MIN", 96);
    int dir = 1;                                     // the stem walks backward (its producer, the ROI crop, walks forward), the launch behind it forward ...
    auto next_dir = [&]() { const int d = snake ? dir : 0; dir ^= 1; return d; };
    auto run_stage = [&](int s, const char* x_in, int nb, char* out, int foff) {
        const char* x = x_in;
        int hw = hw_in[s];
        if (s == 0 && stage2 && P.chain_off && tune_get("RES2_CHAIN", 1)) {
            // round 6: the register-chained form of the stage kernel (res2_chain.hip); RES2_CHAIN=0 is the round-3 stage kernel
            Res2ChainArgs q{};
            q.x = x; q.y = out; q.t1out = bf.m1 + (size_t)foff * 64 * 64 * 128 * es;
            q.wstream = base + P.chain_off; q.zeros = base + P.zero_off;
            q.B = nb; q.y_s2 = ys2 ? 1 : 0;
            q.rev = next_dir();
            launch_res2_chain(q, st);
            return;
        }
        if (s == 0 && stage2) {
            Res2StageArgs q = res2_stage_args(P, base);
            q.x = x; q.y = out; q.t1out = bf.m1 + (size_t)foff * 64 * 64 * 128 * es;
            q.B = nb; q.y_s2 = ys2 ? 1 : 0;
            q.rev = next_dir();
            if (res2_stage_ok(q)) {
                launch_res2_stage(q, st);
                return;
            }
        }
        for (int b = 0; b < nblk[s]; ++b) {
            const BlockPlan& bp = P.blocks[first_blk[s] + b];
            const ConvPlan &c1 = P.convs[bp.c1], &c2 = P.convs[bp.c2], &c3 = P.convs[bp.c3];
            const int ho = hw / c2.stride;
            char* y = (b == nblk[s] - 1) ? out : ((x == bf.pa) ? bf.pb : bf.pa);
            auto mk = [&](const ConvPlan& c, const void* in, int hin, int hout, const void* res, void* o, int relu) {
                ConvArgs q{};
                q.zeros = base + P.zero_off; q.x = in; q.w = base + c.w_off; q.bias = reinterpret_cast<const float*>(base + c.b_off); q.res = res; q.y = o;
                q.B = nb; q.H = hin; q.W = hin; q.Cin = c.Cin; q.Ho = hout; q.Wo = hout; q.Cout = c.Cout;
                q.KH = c.K; q.KW = c.K; q.stride = c.stride; q.pad = c.pad; q.relu = relu;
                q.rev = next_dir();
                if (dtype == IVOSW_BF16 && tune_get("G8", 0) && tune_get("WIDE1X1", 1) && conv1x1_g8_ok(q)) launch_conv1x1_g8(q, st);
                else if (dtype == IVOSW_BF16 && c.fw_off && conv1x1_wide_ok(q) && tune_get("WIDE1X1", 1)) launch_conv1x1_wide(q, base + c.fw_off, st);
                else launch_conv(q, dtype, false, st);
            };
            if (dtype == IVOSW_BF16 && bp.f1_off && tune_get("FUSE_WIDE", 1)) {
                BneckWideArgs q{};
                q.x = x; q.y = y; q.zeros = base + P.zero_off;
                q.fa = base + bp.f1_off; q.ba = reinterpret_cast<const float*>(base + c1.b_off);
                q.fb = base + bp.f2_off; q.bb = reinterpret_cast<const float*>(base + c2.b_off);
                q.fc = base + bp.f3_off; q.bc = reinterpret_cast<const float*>(base + c3.b_off);
                if (bp.ds >= 0) {
                    q.ds = 1; q.fc = base + bp.cat_fw_off; q.bc = reinterpret_cast<const float*>(base + bp.cat_b_off);
                }
                q.B = nb; q.H = hw; q.W = hw; q.Cin = c1.Cin; q.Cmid = c1.Cout;
                if (s == 0 && fwd2) {
                    // t1 ping-pong: b0 -> m2 -> b1 -> ds -> b2 -> m1 (res3's t1 for the frames of the enclosing res3 chunk)
                    const BlockPlan& nx = P.blocks[first_blk[0] + b + 1];          // the next block (res3's first after b == 2)
                    const ConvPlan& n1c = P.convs[nx.c1];
                    q.t1in = b == 0 ? nullptr : (b == 1 ? bf.m2 : bf.ds);
                    q.t1out = b == 0 ? bf.m2 : (b == 1 ? bf.ds : bf.m1 + (size_t)foff * 64 * 64 * 128 * es);
                    q.fd = base + (b == 2 ? nx.fwd1_off : nx.f1_off);
                    q.bd = reinterpret_cast<const float*>(base + n1c.b_off);
                    q.nd = n1c.Cout;
                    q.y_s2 = (b == 2 && ys2) ? 1 : 0;
                }
                if (s == 1 && b == 1 && bp.ds < 0 && snake && bneck_wide_fusable(q) && tune_get("DF3", 2) > 1 && nb >= tune_get("DF3_MIN", 128)) {
                    // res3's three identity blocks DEPTH FIRST over DF3 (2) groups of the launch's frames: b1 b2 b3 for the first half, then
                    // for the second.  Every block's input was written by the launch right before it and its x + y (64 + 64 MB per stream
                    // instead of 128 + 128) fit the memory-side cache together with the other stream's.  Costs the tail balancing of a
                    // 512-workgroup launch (256 workgroups are a single round): + 0.8 % frames/s on average over five boxes at 128 frames per
                    // stream (- 0.2 .. + 2.5 %), - 0.5 % with four groups; a frame's result does not depend on the launch it travels in.
                    const size_t fe = (size_t)hw * hw * c1.Cin * es;
                    const int G = tune_get("DF3", 2);
                    char* y1 = (x == bf.pa) ? bf.pb : bf.pa;
                    char* y2 = (y1 == bf.pa) ? bf.pb : bf.pa;
                    char* bufs[4] = {const_cast<char*>(x), y1, y2, out};
                    for (int g = 0; g < G; ++g) {
                        const int lo = (int)((long)nb * g / G), hi = (int)((long)nb * (g + 1) / G);
                        for (int bb = 1; bb <= 3; ++bb) {
                            const BlockPlan& bq = P.blocks[first_blk[1] + bb];
                            const ConvPlan &d1 = P.convs[bq.c1], &d2 = P.convs[bq.c2], &d3 = P.convs[bq.c3];
                            BneckWideArgs r = q;
                            r.fa = base + bq.f1_off; r.ba = reinterpret_cast<const float*>(base + d1.b_off);
                            r.fb = base + bq.f2_off; r.bb = reinterpret_cast<const float*>(base + d2.b_off);
                            r.fc = base + bq.f3_off; r.bc = reinterpret_cast<const float*>(base + d3.b_off);
                            r.x = bufs[bb - 1] + (size_t)lo * fe; r.y = bufs[bb] + (size_t)lo * fe;
                            r.B = hi - lo;
                            r.rev = next_dir();
                            if (IVOSW_ABLATION && bb == 3 && tune_get("YS2ABL", 0)) r.debug |= 16;      // ablation: the stage's last block writes even pixels only
                            launch_bneck_wide(r, st);
                        }
                    }
                    x = out;
                    break;
                }
                q.rev = next_dir();                  // (after the depth-first branch, which takes a direction per launch of its own)
                if (bp.ds < 0 && bneck_wide_fusable(q) && bneck_stage_fusable(q) && b + 1 < nblk[s]) {
                    // the rest of the stage is identity blocks of this shape: chain them inside one launch
                    BneckStageArgs sa{};
                    const char* xi = x;
                    for (int bb = b; bb < nblk[s]; ++bb) {
                        const BlockPlan& bq = P.blocks[first_blk[s] + bb];
                        const ConvPlan &d1 = P.convs[bq.c1], &d2 = P.convs[bq.c2], &d3 = P.convs[bq.c3];
                        // in place (tunable INPLACE4): a frame belongs to ONE workgroup, phase A has consumed x before phase C writes, and an
                        // element of y goes where the residual it was formed from came from - half the footprint of the chain in the caches
                        char* yi = (bb == nblk[s] - 1) ? out : ((tune_get("INPLACE4", 1) && bb > b) ? const_cast<char*>(xi) : ((xi == bf.pa) ? bf.pb : bf.pa));
                        BneckWideArgs& r = sa.blk[sa.n++];
                        r = q;
                        r.x = xi; r.y = yi;
                        r.fa = base + bq.f1_off; r.ba = reinterpret_cast<const float*>(base + d1.b_off);
                        r.fb = base + bq.f2_off; r.bb = reinterpret_cast<const float*>(base + d2.b_off);
                        r.fc = base + bq.f3_off; r.bc = reinterpret_cast<const float*>(base + d3.b_off);
                        if (IVOSW_ABLATION && s == 2 && bb == nblk[s] - 1 && tune_get("YS2ABL", 0)) r.debug |= 16;
                        xi = yi;
                    }
                    launch_bneck_wide_stage(sa, st);
                    x = xi;
                    break;
                }
                if (bneck_wide_fusable(q)) {
                    launch_bneck_wide(q, st);
                    x = y;
                    continue;
                }
            }
            if (dtype == IVOSW_BF16 && (bp.ds < 0 || c2.stride == 1) && tune_get("FUSE", 1)) {
                BneckArgs q{};
                q.x = x; q.y = y; q.zeros = base + P.zero_off;
                q.wa = base + c1.w_off; q.ba = reinterpret_cast<const float*>(base + c1.b_off);
                q.wb = base + c2.w_off; q.bb = reinterpret_cast<const float*>(base + c2.b_off);
                q.wc = base + c3.w_off; q.bc = reinterpret_cast<const float*>(base + c3.b_off);
                if (bp.ds >= 0) {
                    q.wd = base + P.convs[bp.ds].w_off;
                    q.bd = reinterpret_cast<const float*>(base + P.convs[bp.ds].b_off);
                }
                q.B = nb; q.H = hw; q.W = hw; q.Cin = c1.Cin; q.Cmid = c1.Cout;
                if (bneck_fusable(q)) {
                    launch_bneck(q, st);
                    x = y;
                    continue;
                }
            }
            if (dtype == IVOSW_BF16 && s == 1 && b == 0 && fwd2 && bp.ds >= 0 && bp.f2_off && bp.cat_fw_off && tune_get("FIRST3", 1)) {
                // res3's first block behind its forwarded conv1 as ONE launch: t2 never leaves LDS (stage_first.hip; bit-identical to the
                // two layer launches below)
                StageFirstArgs q{};
                q.t1 = bf.m1; q.x2 = x; q.y = y; q.zeros = base + P.zero_off;
                q.fw2 = base + bp.f2_off; q.b2 = reinterpret_cast<const float*>(base + c2.b_off);
                q.fwc = base + bp.cat_fw_off; q.bc = reinterpret_cast<const float*>(base + bp.cat_b_off);
                q.B = nb; q.Ho = ho; q.Wo = ho; q.Cm = c2.Cout; q.C2 = P.convs[bp.ds].Cin;
                q.H2 = ys2 ? hw / 2 : hw; q.W2 = q.H2; q.stride2 = ys2 ? 1 : 2;
                if (stage_first_ok(q)) {
                    q.rev = next_dir();
                    launch_stage_first(q, st);
                    x = y;
                    hw = ho;
                    continue;
                }
            }
            if (!(s == 1 && b == 0 && fwd2)) mk(c1, x, hw, hw, nullptr, bf.m1, 1);     // forwarded: res2's last block wrote it
            if (dtype == IVOSW_F32X3 && s == 0 && bp.ds < 0 && hw == 64 && c2.K == 3 && c2.stride == 1 && c2.Cin == 64 && c2.Cout == 64 && c3.Cin == 64 &&
                c3.Cout == 256 && tune_get("FUSE_TAIL_X3", 1)) {
                // the three-pass mode's identity blocks of res2 behind their conv1: 3x3 -> conv3 + residual as one launch (conv.hip: t2 stays on the CU)
                launch_res2_tail_x3(bf.m1, x, base + c2.w_off, reinterpret_cast<const float*>(base + c2.b_off), base + c3.w_off,
                                    reinterpret_cast<const float*>(base + c3.b_off), base + P.zero_off, y, nb, hw, hw, next_dir(), st);
                x = y;
                continue;
            }
            if (dtype == IVOSW_F32X3 && s == 1 && bp.ds < 0 && hw == 32 && c2.K == 3 && c2.stride == 1 && c2.Cin == 128 && c2.Cout == 128 && c3.Cin == 128 &&
                c3.Cout == 512 && tune_get("FUSE_TAIL3_X3", 1)) {
                launch_res3_tail_x3(bf.m1, x, base + c2.w_off, reinterpret_cast<const float*>(base + c2.b_off), base + c3.w_off,
                                    reinterpret_cast<const float*>(base + c3.b_off), base + P.zero_off, y, nb, hw, hw, next_dir(), st);
                x = y;
                continue;
            }
            if (dtype == IVOSW_F32X3 && s == 0 && bp.ds >= 0 && hw == 64 && c2.K == 3 && c2.stride == 1 && c2.Cin == 64 && c2.Cout == 64 && c3.Cin == 64 &&
                c3.Cout == 256 && P.convs[bp.ds].Cin == 64 && P.convs[bp.ds].stride == 1 && tune_get("FUSE_DS", 1) && tune_get("FUSE_TAIL_X3", 1)) {
                // res2's first block: 3x3 -> [conv3 | downsample] (K = 64 + 64, the block input as the second half) as one launch
                launch_res2_tail_x3(bf.m1, nullptr, base + c2.w_off, reinterpret_cast<const float*>(base + c2.b_off), base + bp.cat_w_off,
                                    reinterpret_cast<const float*>(base + bp.cat_b_off), base + P.zero_off, y, nb, hw, hw, next_dir(), st, x);
                x = y;
                continue;
            }
            mk(c2, bf.m1, hw, ho, nullptr, bf.m2, 1);
            const void* idt = x;
            if (bp.ds >= 0 && tune_get("FUSE_DS", 1)) {
                // first block of a stage: relu(conv3(t2) + downsample(x)) as ONE GEMM over K = C + Cin (the block input,
                // sampled at the downsample stride, is the second A source): the 4C-channel downsample output never
                // goes to HBM and back
                const ConvPlan& cd = P.convs[bp.ds];
                ConvArgs q{};
                q.zeros = base + P.zero_off; q.x = bf.m2; q.w = base + bp.cat_w_off; q.bias = reinterpret_cast<const float*>(base + bp.cat_b_off);
                q.res = nullptr; q.y = y; q.B = nb; q.H = ho; q.W = ho; q.Cin = c3.Cin; q.Ho = ho; q.Wo = ho; q.Cout = c3.Cout;
                q.KH = 1; q.KW = 1; q.stride = 1; q.pad = 0; q.relu = 1;
                q.x2 = x; q.Cin2 = cd.Cin; q.H2 = hw; q.W2 = hw; q.stride2 = cd.stride;
                q.rev = next_dir();
                if (s == 1 && ys2) { q.H2 = hw / 2; q.W2 = hw / 2; q.stride2 = 1; }      // res2's output arrives already subsampled
                if (dtype == IVOSW_BF16 && tune_get("G8", 0) && tune_get("WIDE1X1", 1) && conv1x1_g8_ok(q)) launch_conv1x1_g8(q, st);
                else if (dtype == IVOSW_BF16 && bp.cat_fw_off && conv1x1_wide_ok(q) && tune_get("WIDE1X1", 1)) launch_conv1x1_wide(q, base + bp.cat_fw_off, st);
                else launch_conv(q, dtype, false, st);
                x = y;
                hw = ho;
                continue;
            }
            if (bp.ds >= 0) {
                mk(P.convs[bp.ds], x, hw, ho, nullptr, bf.ds, 0);
                idt = bf.ds;
            }
            mk(c3, bf.m2, ho, ho, idt, y, 1);  // relu(bn3(conv3) + identity)
            x = y;
            hw = ho;
        }
    };

    for (int f3 = 0; f3 < B; f3 += cs[3]) {
        const int n3 = std::min(cs[3], B - f3);
        for (int f2 = f3; f2 < f3 + n3; f2 += cs[2]) {
            const int n2 = std::min(cs[2], f3 + n3 - f2);
            for (int f1 = f2; f1 < f2 + n2; f1 += cs[1]) {
                const int n1 = std::min(cs[1], f2 + n2 - f1);
                for (int f0 = f1; f0 < f1 + n1; f0 += cs[0]) {
                    const int nb = std::min(cs[0], f1 + n1 - f0);
                    // K3: ROI crop-resize + normalise -> NHWC4
                    span_close(st, slot);
                    launch_roi_sample(tf, tp, bf.yxhw + (size_t)f0 * 4, u0 + f0, nb, H, W, dtype, sm, nrm, bf.roi, st);
                    tap(1, bf.roi, nb * E_ROI * es);
                    // K4: stem 7x7/2 (RGB|P) + BN + ReLU, then 3x3/2 max pool (bf16: one fused kernel unless the stem tap is wanted)
                    span_open(st, slot);
                    if (dtype == IVOSW_BF16 && tap_stage != 2 && tune_get("FUSE_STEM", 1)) {
                        dir = 1;                 // every res2 chunk starts the alternation anew at its stem
                        launch_stem_pool(bf.roi, base + P.stem_w_off, reinterpret_cast<const float*>(base + P.stem_b_off), nb, bf.pa, st, next_dir());
                    } else if (dtype == IVOSW_F32X3 && tap_stage != 2 && tune_get("FUSE_STEM_X3", 1)) {
                        // the three-pass mode's stem + pool as one launch (stem.hip): fp32 ROI tile in, split pooled map out
                        launch_stem_pool_x3(bf.roi, base + P.stem_w_off, reinterpret_cast<const float*>(base + P.stem_b_off), nb, bf.pa, st, 0);
                    } else {
                        ConvArgs a{};
                        a.zeros = base + P.zero_off; a.x = bf.roi; a.w = base + P.stem_w_off; a.bias = reinterpret_cast<const float*>(base + P.stem_b_off);
                        a.res = nullptr; a.y = bf.stem; a.B = nb; a.H = 256; a.W = 256; a.Cin = 4; a.Ho = 128; a.Wo = 128;
                        a.Cout = 64; a.KH = 7; a.KW = 7; a.stride = 2; a.pad = 3; a.relu = 1;
                        launch_conv(a, dtype, true, st);
                        tap(2, bf.stem, (size_t)nb * 128 * 128 * 64 * es);
                        launch_maxpool(bf.stem, nb, 128, 128, 64, dtype, bf.pa, st);
                    }
                    tap(3, bf.pa, (size_t)nb * 64 * 64 * 64 * es);
                    char* o2 = bf.in[0] + (size_t)(f0 - f1) * (ys2 ? E_OUT[0] / 4 : E_OUT[0]) * es;
                    run_stage(0, bf.pa, nb, o2, f0 - f1);
                    tap(4, o2, nb * E_OUT[0] * es);
                }
                char* o3 = bf.in[1] + (size_t)(f1 - f2) * E_OUT[1] * es;
                run_stage(1, bf.in[0], n1, o3, 0);
                tap(5, o3, n1 * E_OUT[1] * es);
            }
            char* o4 = bf.in[2] + (size_t)(f2 - f3) * E_OUT[2] * es;
            run_stage(2, bf.in[1], n2, o4, 0);
            tap(6, o4, n2 * E_OUT[2] * es);
        }
        run_stage(3, bf.in[2], n3, bf.pa, 0);  // c0*8 frames x 131k elements == c0 * E_BIG: fits a ping-pong buffer
        tap(7, bf.pa, n3 * E_OUT[3] * es);
        // K6: 8x8 average pool + fc1
        span_close(st, slot);
        launch_pool_fc(bf.pa, n3, dtype, reinterpret_cast<const float*>(base + P.fcw_off),
                       reinterpret_cast<const float*>(base + P.fcb_off), scores + f3, tap_stage == 8 ? bf.pooled : nullptr, st);
        tap(8, bf.pooled, (size_t)n3 * 2048 * sizeof(float));
    }
}

#ifdef IVOSW_PROBES
// Tuning probe: one launch of the res2 stage kernel on x [B,64,64,64] bf16 with the weights of a packed bf16 arena; y [B,64,64,256]
// (y_s2: [B,32,32,256]), t1out [B,64,64,128]; ts (may be NULL): [B*32][16] s_memtime stamps at the phase boundaries.
extern "C" int ivosw_res2_stage_probe(const void* packed, const void* x, void* y, void* t1out, int B, int y_s2, unsigned long long* ts,
                                      ivosw_stream_t stream) {
    IVOSW_REQUIRE(packed && x && y && t1out && B > 0, "null pointer");
    IVOSW_ON_DEVICE_OF(y);
    if (tune_get("RES2_CHAIN", 1) && plan_for(IVOSW_BF16).chain_off) {      // ts: [B * 32][8] stamps (res2_chain.hip) instead of [B * 32][16]
        const Plan& P = plan_for(IVOSW_BF16);
        Res2ChainArgs c{};
        c.x = x; c.y = y; c.t1out = t1out; c.wstream = static_cast<const char*>(packed) + P.chain_off; c.zeros = static_cast<const char*>(packed) + P.zero_off;
        c.B = B; c.y_s2 = y_s2; c.ts = ts;
        launch_res2_chain(c, as_stream(stream));
        IVOSW_CHECK_LAUNCH();
        return IVOSW_OK;
    }
    Res2StageArgs q = res2_stage_args(plan_for(IVOSW_BF16), static_cast<const char*>(packed));
    q.x = x; q.y = y; q.t1out = t1out; q.B = B; q.y_s2 = y_s2; q.ts = ts;
    q.debug = tune_get("R2DBG", 0);
    IVOSW_REQUIRE(res2_stage_ok(q), "the packed arena lacks the fragment-ordered res2 weights");
    launch_res2_stage(q, as_stream(stream));
    IVOSW_CHECK_LAUNCH();
    return IVOSW_OK;
}
#endif  // IVOSW_PROBES
