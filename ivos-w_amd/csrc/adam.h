// The clamp + Adam update of one element, shared by clamp_adam_kernel (dqn.hip) and the data-parallel step's fused
// all-reduce + clamp + Adam kernel (p2p.hip) so that both evaluate the same expression tree.
// torch.optim.Adam (non-amsgrad, coupled L2): g += wd*p; m.lerp_(g, 1-b1); v = b2*v + (1-b2)*g*g;
// p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps).  The clamp comes first (reference models/agent.py:157-159);
// gscale (= 1/world) is applied before the clamp so that the clamp sees the averaged gradient.
#pragma once
#include "common.h"

namespace ivosw {

// beta^step by squaring: multiplications only, so the host (ivosw_clamp_adam) and the device (clamp_adam_dev_kernel) get the same
// float64 bits — libm's pow and the device's differ in the last place now and then, which moved step_size by a float ulp
__host__ __device__ inline double ipow(double b, int n) {
    double r = 1.0;
    while (n > 0) {
        if (n & 1) r *= b;
        b *= b;
        n >>= 1;
    }
    return r;
}

__device__ __forceinline__ float clamp_adam_elem(float g, float pi, float& mi, float& vi, float step_size, float bc2_sqrt, float beta1,
                                                 float beta2, float eps, float wd, float clampv, float gscale) {
    float gi = g * gscale;
    gi = fminf(fmaxf(gi, -clampv), clampv);
    gi = fmaf(wd, pi, gi);
    mi = fmaf(gi - mi, 1.0f - beta1, mi);
    vi = fmaf(1.0f - beta2, gi * gi, vi * beta2);
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    return pi - step_size * (mi / denom);
}

}  // namespace ivosw
