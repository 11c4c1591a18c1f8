// Internal launchers of the convolution tower (conv.hip), used by assess.hip.
#pragma once
#include "common.h"

namespace ivosw {

struct ConvArgs {
    const void* x;      // NHWC [B,H,W,Cin] (stem: [B,256,256,4])
    const void* w;      // packed [Cout][KH*KW*Cin], BN scale folded
    const float* bias;  // [Cout] folded BN shift
    const void* res;    // NHWC [B,Ho,Wo,Cout] residual added before the ReLU, or nullptr
    void* y;            // NHWC [B,Ho,Wo,Cout]
    const void* zeros;  // >= 16 B of device zeros: source of the zero-padding taps for the LDS-DMA loader
    int B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, relu;
    // optional K-extension (1x1 only): a SECOND input tensor x2 [B,H2,W2,Cin2] sampled at (oy*stride2, ox*stride2) supplies
    // K-tiles Cin/64 .. (Cin+Cin2)/64 - 1; the weights are [Cout][Cin + Cin2].  Folds a block's downsample conv into conv3.
    const void* x2;
    int Cin2, H2, W2, stride2;
    int nmajor;         // tile order of the wave-specialised kernel: 0 m-major, 1 n-major
    int nt;             // bit 0: non-temporal output stores, bit 1: non-temporal residual loads (generic epilogue), bit 2: also in the patch / fused kernels (tunable NT, default 3)
    int debug;          // ablation bits (IVOSW_DEBUG_CONV, tuning only): 1 skip epilogue stores, 2 skip MFMA, 4 skip DMA after the first tile
    int rev;            // 1: pixel tiles are taken in DESCENDING order (see BneckWideArgs::rev; tunable SNAKE)
    int x3;             // IVOSW_F32X3: contraction as three bf16 MFMA passes, w is the PRE-SPLIT weight array; 2 (what launch_conv sets):
                        // activations are stored split as well - [32 x bf16 hi | 32 x bf16 lo] per 32-channel group, the bytes of 32 floats -
                        // by every epilogue and read without VALU work by every consumer except the stem (fp32 ROI tile)
};

void launch_conv(const ConvArgs& a, int dtype, bool stem, hipStream_t st);
// in place: every 32-float K-tile of w [rows][K] (128 bytes) becomes [32 x bf16 hi | 32 x bf16 lo], hi = RNE(w), lo = RNE(w - hi)
void launch_split_weights_x3(void* w, long rows, int K, hipStream_t st);
// split activation layout -> plain fp32, in place (test taps of the IVOSW_F32X3 mode); nfloats % 32 == 0
void launch_unsplit_x3(void* buf, size_t nfloats, hipStream_t st);

// res3's first bottleneck behind its conv1 (3x3 stride 2, then [conv3 | downsample]) as one launch, bf16 (stage_first.hip)
struct StageFirstArgs {
    const void* t1;      // NHWC [B, 2 Ho, 2 Wo, Cm]: conv1's output (res2's last block forwards it)
    const void* x2;      // NHWC [B, H2, W2, C2]: the block input, sampled at (oy * stride2, ox * stride2) for the downsample
    void* y;             // NHWC [B, Ho, Wo, 4 Cm]
    const void* fw2;     // 3x3 weights [Cm][9 Cm] in MFMA-operand order (launch_fragpack)
    const float* b2;     // [Cm]
    const void* fwc;     // [conv3 | downsample] weights [4 Cm][Cm + C2] in MFMA-operand order
    const float* bc;     // [4 Cm]: the two folded BN shifts summed
    const void* zeros;   // >= 256 B of device zeros
    int B, Ho, Wo, Cm, C2, H2, W2, stride2;
    int nt, rev;
};
bool stage_first_ok(const StageFirstArgs& a);
void launch_stage_first(const StageFirstArgs& a, hipStream_t st);

// one whole identity bottleneck (1x1 -> 3x3 -> 1x1 + residual), bf16, fused in one kernel (bottleneck.hip)
struct BneckArgs {
    const void* x;       // NHWC [B,H,W,Cin]; also the residual
    void* y;             // NHWC [B,H,W,4*Cmid]
    const void* wa; const float* ba;   // [Cmid][Cin]      packed as by launch_pack_conv
    const void* wb; const float* bb;   // [Cmid][9*Cmid]
    const void* wc; const float* bc;   // [4*Cmid][Cmid]
    const void* wd; const float* bd;   // downsample [4*Cmid][Cin] (first block of a stage, stride 1) or null: identity residual
    const void* zeros;   // >= 256 B of device zeros
    int B, H, W, Cin, Cmid;
    unsigned long long* ts;  // optional [grid][16] s_memtime stamps at the phase boundaries (ivosw_bneck_probe), else null
    int nt;              // non-temporal y stores
    int stagger;         // > 0: odd first-round workgroups start stagger x 8128 cycles late (tunable STAGGER)
    int debug;           // ablation bits (tunable BDBG): 1 no residual read, 2 no stores, 4 K-tile 0 only (L2-hot A), 8 no MFMA
};
void* prof_begin(const ConvArgs& a, int es, hipStream_t st);   // measurement hook (ivosw_profile_*), see conv.hip
void prof_end(void* tok, hipStream_t st);
void span_open(hipStream_t st, int slot = 0);    // span mode: start / end of an uninterrupted run of tower launches (assess.hip)
void span_close(hipStream_t st, int slot = 0);
void span_group_begin();                         // the spans opened until span_group_end() belong to ONE forward pass (two streams)
void span_group_end();
bool bneck_fusable(const BneckArgs& a);
void launch_bneck(const BneckArgs& a, hipStream_t st);

// one whole identity bottleneck of a wide stage (res4: 16x16 frames, 1024 -> 256 -> 256 -> 1024), bf16, one frame per
// workgroup, weights pre-arranged in MFMA-operand order (bottleneck_wide.hip)
struct BneckWideArgs {
    const void* x;       // NHWC [B,H,W,Cin]; also the residual
    void* y;             // NHWC [B,H,W,Cin]
    const void* fa; const float* ba;   // conv1 [Cmid][Cin]     in fragment order (launch_fragpack)
    const void* fb; const float* bb;   // conv2 [Cmid][9*Cmid]
    const void* fc; const float* bc;   // conv3 [Cin][Cmid]
    const void* zeros;                 // >= 256 B of device zeros (halo variant)
    int B, H, W, Cin, Cmid;
    unsigned long long* ts;            // optional [B][8] s_memtime stamps at the phase boundaries (ivosw_bneck_wide_probe)
    int ds;                            // 1: first block of res2 (Cin = Cmid = 64): fc = [conv3 | downsample] along K (K = 128), bc = bias sum
    int debug;                         // (unused by the current kernels; the ablation bits of the res2 experiments were removed
                                       // again: ~20 branches in a kernel that is instruction-issue-bound)    // conv1 forwarding (res2, tunable FWD2): a block's first 1x1 is a pointwise function of the previous block's output, so the
    // PREVIOUS block computes it on its own tile while y is still on chip (phase D) and this block starts from t1:
    const void* t1in;                  // NHWC [B,H,W,Cmid] = relu(bn1(conv1(x))) written by the previous block, or NULL (compute it here)
    void* t1out;                       // NHWC [B,H,W,nd]: the NEXT block's conv1 applied to this block's output y, or NULL
    const void* fd; const float* bd;   // that conv1 [nd][4*Cmid] in fragment order, and its bias
    int nd;                            // 64 (next block of res2) or 128 (first block of res3)
    int y_s2;                          // 1: y is written at the even pixels only, compactly [B,H/2,W/2,4*Cmid] (its only consumer is a stride-2 1x1)
    int rev;                           // 1: tiles are taken in DESCENDING order (the launch starts with what the previous launch wrote last)
};
// a run of consecutive identity blocks of one stage (res4: one frame per workgroup) in ONE launch
struct BneckStageArgs {
    int n;
    int stagger;         // experiment (tunable STAGGER, 0 = off): cycles by which every second workgroup OF EACH XCD starts late
    BneckWideArgs blk[6];
};
// the whole of res2 (three bottlenecks at 64 x 64 + res3's forwarded conv1) in ONE launch, bf16 (res2_stage.hip): a workgroup owns an
// 8 x 16 output tile through all three blocks, y0 / y1 never leave the chip
struct Res2StageArgs {
    const void* x;                     // NHWC [B,64,64,64]: the pooled stem output
    void* y;                           // y2: NHWC [B,64,64,256], or (y_s2) its even pixels, compactly [B,32,32,256]
    void* t1out;                       // NHWC [B,64,64,128]: res3's first 1x1 (+ BN + ReLU) applied to y2
    const void* fa0; const float* ba0; // block 0 conv1 [64][64] in fragment order (launch_fragpack)
    const void* fb[3]; const float* bb[3];   // conv2 [64][9*64] of the three blocks
    const void* fc[3]; const float* bc[3];   // conv3: block 0 = [conv3 | downsample] [256][128] with the bias sum; blocks 1, 2 = [256][64]
    const void* fd[3]; const float* bd[3];   // the NEXT block's conv1 [64|64|128][256]: blocks 1, 2 of res2 and block 0 of res3
    const void* zeros;                 // >= 256 B of device zeros
    int B, y_s2;
    unsigned long long* ts;            // optional [grid][16] s_memtime stamps at the phase boundaries (ivosw_res2_stage_probe)
    int debug;                         // IVOSW_ABLATION builds only (tunable R2DBG): 1 no weight loads, 2 pixel fragments read once per phase, 4 no MFMAs
    int rev;                           // 1: tiles in descending order
};
bool res2_stage_ok(const Res2StageArgs& a);
void launch_res2_stage(const Res2StageArgs& a, hipStream_t st);
// the same stage with the pointwise chains kept in registers and every weight through one LDS ring (res2_chain.hip, round 6)
struct Res2ChainArgs {
    const void* x;                     // NHWC [B,64,64,64]: the pooled stem output
    void* y;                           // y2: NHWC [B,64,64,256], or (y_s2) its even pixels, compactly [B,32,32,256]
    void* t1out;                       // NHWC [B,64,64,128]: res3's first 1x1 (+ BN + ReLU) applied to y2
    const void* wstream;               // the tile's 528 weight fragments in consumption order (launch_res2_chain_pack)
    const void* zeros;                 // >= 256 B of device zeros
    int B, y_s2, rev;
    unsigned long long* ts;            // optional [grid][8] s_memtime stamps: start, first group landed, A0, block 0, block 1, block 2, end
};
struct Res2ChainPackArgs {             // K-major bf16 weights (BN folded) and fp32 biases of the packed arena
    const bf16_t* w1[4]; const float* b1[4];   // conv1 of res2's three blocks and of res3's first block ([64][64], [64][256] x 2, [128][256])
    const bf16_t* w2[3]; const float* b2[3];   // 3x3 [64][9 * 64]
    const bf16_t* w3[3]; const float* b3[3];   // conv3 [256][64]
    const bf16_t* wd; const float* bd;         // block 0's downsample [256][64]
    void* out;
};
size_t res2_chain_stream_bytes();
void launch_res2_chain_pack(const Res2ChainPackArgs& a, hipStream_t st);
bool res2_chain_ok(const Res2ChainArgs& a);
void launch_res2_chain(const Res2ChainArgs& a, hipStream_t st);
bool bneck_stage_fusable(const BneckWideArgs& a);
void launch_bneck_wide_stage(const BneckStageArgs& s, hipStream_t st);
bool bneck_wide_fusable(const BneckWideArgs& a);
void launch_bneck_wide(const BneckWideArgs& a, hipStream_t st);
void launch_fragpack(const void* w, int Cout, int K, void* out, hipStream_t st);
// bf16 1x1 convolution (optionally with the K-extension x2) on fragment-ordered weights `fw` (bottleneck_wide.hip)
// the same layers on the 256 x 256 8-phase contraction (gemm_8phase.hip): plain K-major weights a.w
bool conv1x1_g8_ok(const ConvArgs& a);
void launch_conv1x1_g8(const ConvArgs& a, hipStream_t st);
bool conv1x1_wide_ok(const ConvArgs& a);
void launch_conv1x1_wide(const ConvArgs& a, const void* fw, hipStream_t st);

// runtime tunables (capi.cpp): value of IVOSW_TUNE_<KEY> from the environment unless ivosw_tune_set() overrode it
int tune_get(const char* key, int dflt);
void launch_pack_conv(const float* w, const float* g, const float* b, const float* rm, const float* rv, int Cout, int Cin,
                      int KH, int KW, int dtype, void* ow, float* ob, hipStream_t st);
// [Cout][K1] | [Cout][K2] -> [Cout][K1+K2] (same dtype), bias = b1 + b2
void launch_concat_k(const void* w1, const float* b1, int K1, const void* w2, const float* b2, int K2, int Cout, int dtype,
                     void* ow, float* ob, hipStream_t st);
void launch_pack_stem(const float* w3, const float* w1, const float* g, const float* b, const float* rm, const float* rv,
                      int dtype, void* ow, float* ob, hipStream_t st);
// bf16: stem 7x7/2 + BN + ReLU + 3x3/2 max-pool fused (stem.hip): roi [B,256,256,4] -> out [B,64,64,64]
void launch_stem_pool(const void* roi, const void* w, const float* bias, int B, void* out, hipStream_t st, int rev = 0);
// IVOSW_F32X3, res2's identity blocks behind their conv1: 3x3 (64 -> 64) -> conv3 (64 -> 256) + residual x, split layout throughout (conv.hip)
void launch_res2_tail_x3(const void* t1, const void* x, const void* w2, const float* b2, const void* w3, const float* b3, const void* zeros, void* y,
                         int B, int H, int W, int rev, hipStream_t st, const void* p2 = nullptr);   // p2: res2's first block ([conv3 | downsample], no residual)
void launch_res3_tail_x3(const void* t1, const void* x, const void* w2, const float* b2, const void* w3, const float* b3, const void* zeros, void* y,
                         int B, int H, int W, int rev, hipStream_t st);   // the same for res3's identity blocks (128 -> 128 3x3, 128 -> 512 expand)
void launch_stem_pool_x3(const void* roi, const void* wsplit, const float* bias, int B, void* out, hipStream_t st, int rev = 0);   // IVOSW_F32X3: fp32 ROI in, split pooled map out
void launch_maxpool(const void* x, int B, int H, int W, int C, int dtype, void* y, hipStream_t st);
void launch_pool_fc(const void* x, int B, int dtype, const float* fcw, const float* fcb, float* score, float* pooled,
                    hipStream_t st);

}  // namespace ivosw
