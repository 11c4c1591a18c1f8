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

// ---- device-resident state of the captured DQN step (dqn.hip owns the kernels; brain.hip's one-call step shares the layouts)
// The minibatch draw (uniform with replacement): slot b of draw number c = a splitmix64 finaliser of (seed, c, b) scaled to [0, n)
// by a 64 x 64 -> high-64 multiply: integer arithmetic only (host mirrors: ivosw_replay_draw_index, momory_pool.draw_indices).
struct DrawState {
    unsigned long long seed;
    unsigned counter;
    unsigned ticket;
};
static_assert(sizeof(DrawState) == 16, "DrawState layout (seed at byte 0, counter at byte 8)");
__host__ __device__ inline unsigned long long draw_mix(unsigned long long seed, unsigned counter, unsigned slot) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * ((unsigned long long)counter + 1) + 0xD1B54A32D192ED03ull * ((unsigned long long)slot + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Adam's step counter and bias corrections kept ON the device, so that a captured HIP graph of the DQN step replays
// correctly: tick advances step, evaluates beta1^t / beta2^t in float64 and publishes step_size / sqrt(bc2).
// Layout (32 bytes): the step counter is the int32 at byte 16; a caller resumes from host step k by writing k there.
struct AdamDevState {
    double b1t, b2t;
    int step;
    float step_size, bc2_sqrt;
    unsigned ticket;     // byte 28: workgroups of the running clamp+Adam launch that have finished (0 between launches)
};
static_assert(sizeof(AdamDevState) == 32, "AdamDevState layout (step at byte 16, ticket at byte 28)");

// up to six split-K slab sets reduced in one launch (gemm_f32.h: splitk_reduce_group_kernel; brain.hip: folded into clamp + Adam)
constexpr int REDUCE_MAX = 6;
struct ReduceGroup {
    const float* slabs[REDUCE_MAX];
    float* out[REDUCE_MAX];
    int n[REDUCE_MAX], nslab[REDUCE_MAX];
};

}  // namespace ivosw
