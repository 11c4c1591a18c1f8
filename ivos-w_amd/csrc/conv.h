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
    int debug;          // ablation bits (IVOSW_DEBUG_CONV, tuning only): 1 skip epilogue stores, 2 skip MFMA, 4 skip DMA after the first tile
};

void launch_conv(const ConvArgs& a, int dtype, bool stem, hipStream_t st);
void launch_pack_conv(const float* w, const float* g, const float* b, const float* rm, const float* rv, int Cout, int Cin,
                      int KH, int KW, int dtype, void* ow, float* ob, hipStream_t st);
void launch_pack_stem(const float* w3, const float* w1, const float* g, const float* b, const float* rm, const float* rv,
                      int dtype, void* ow, float* ob, hipStream_t st);
void launch_maxpool(const void* x, int B, int H, int W, int C, int dtype, void* y, hipStream_t st);
void launch_pool_fc(const void* x, int B, int dtype, const float* fcw, const float* fcb, float* score, float* pooled,
                    hipStream_t st);

}  // namespace ivosw
