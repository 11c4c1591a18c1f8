// Internal launchers shared between assess_front.hip and assess.hip.
#pragma once
#include "common.h"

namespace ivosw {
struct RoiNorm {
    float mean[3];
    float std[3];
    const float* dev;  // optional device copy {mean[3], std[3]} (the Encoder.mean/std buffers of a checkpoint)
};
void launch_mask_bbox(const float* tp, int B, int H, int W, float* yxhw, int32_t* scratch, hipStream_t st);
void launch_roi_sample(const float* tf, const float* tp, const float* yxhw, int B, int H, int W, int dtype,
                       const RoiNorm& nrm, void* roi, hipStream_t st);
}  // namespace ivosw
