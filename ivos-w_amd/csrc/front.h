// Internal launchers shared between assess_front.hip and assess.hip.
#pragma once
#include "common.h"

namespace ivosw {
struct RoiNorm {
    float mean[3];
    float std[3];
    const float* dev;  // optional device copy {mean[3], std[3]} (the Encoder.mean/std buffers of a checkpoint)
};
// Which frame and which mask plane sample b of a batch reads.  Samples are object-major: b = obj * n_frames + frame, so
// O objects of one n-frame video share ONE copy of the frames (utils/utils_agent.py:118-119 scores one object at a time on
// the same all_F): frame = tf + (b % n_frames) * 3*H*W, mask = tp + (b / n_frames) * stride_obj + (b % n_frames) * stride_frame
// (strides in elements).  A plain batch is {n_frames = B, stride_frame = H*W, stride_obj = 0}.
struct SampleMap {
    int n_frames;
    long stride_frame, stride_obj;
};
void launch_mask_bbox(const float* tp, int b0, int B, int H, int W, const SampleMap& sm, float* yxhw, int32_t* scratch, hipStream_t st);
void launch_roi_sample(const float* tf, const float* tp, const float* yxhw, int b0, int B, int H, int W, int dtype,
                       const SampleMap& sm, const RoiNorm& nrm, void* roi, hipStream_t st);
}  // namespace ivosw
