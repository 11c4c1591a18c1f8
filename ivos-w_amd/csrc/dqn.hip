// Fused clamp + Adam over the flat parameter arena (K9), hard target sync (K10), replay gather (K11).
// Reference: models/agent.py:157-165 (clamp, optim.Adam(lr, weight_decay) step, target sync),
// datasets/agent_dataset.py:71-115 + train_agent.py:177-182 (minibatch assembly).
#include "common.h"

namespace ivosw {

// torch.optim.Adam (non-amsgrad, coupled L2): g += wd*p; m.lerp_(g, 1-b1); v = b2*v + (1-b2)*g*g;
// p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps).  The clamp comes first (agent.py:157-159);
// grad_scale (=1/world) is applied before the clamp so the clamp sees the averaged gradient.
__global__ void clamp_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                  float* __restrict__ v, int n, float step_size, float bc2_sqrt, float beta1, float beta2,
                                  float eps, float wd, float clampv, float gscale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float gi = g[i] * gscale;
    gi = fminf(fmaxf(gi, -clampv), clampv);
    const float pi = p[i];
    gi = fmaf(wd, pi, gi);
    float mi = m[i], vi = v[i];
    mi = fmaf(gi - mi, 1.0f - beta1, mi);
    vi = fmaf(1.0f - beta2, gi * gi, vi * beta2);
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - step_size * (mi / denom);
}

__global__ void replay_gather_kernel(const float* __restrict__ old_iou, const float* __restrict__ new_iou,
                                     const float* __restrict__ ann, const float* __restrict__ nann,
                                     const int64_t* __restrict__ action, const float* __restrict__ rstep,
                                     const float* __restrict__ rdone, const int64_t* __restrict__ idx, int B, int T,
                                     float* __restrict__ state, float* __restrict__ new_state,
                                     int64_t* __restrict__ action_out, float* __restrict__ rstep_out,
                                     float* __restrict__ rdone_out) {
    const int b = blockIdx.x;
    const int64_t src = idx[b];
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        const size_t s = (size_t)src * T + t, d = ((size_t)b * T + t) * 2;
        state[d] = old_iou[s];
        state[d + 1] = ann[s];
        new_state[d] = new_iou[s];
        new_state[d + 1] = nann[s];
    }
    if (threadIdx.x == 0) {
        action_out[b] = action[src];
        rstep_out[b] = rstep[src];
        rdone_out[b] = rdone[src];
    }
}

}  // namespace ivosw

using namespace ivosw;

extern "C" int ivosw_clamp_adam(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n, int step,
                                float lr, float beta1, float beta2, float eps, float weight_decay, float clamp,
                                float grad_scale, ivosw_stream_t stream) {
    IVOSW_REQUIRE(params && grads && exp_avg && exp_avg_sq, "null pointer");
    IVOSW_REQUIRE(n > 0 && step >= 1, "n must be positive and step >= 1");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    hipLaunchKernelGGL(clamp_adam_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), params, grads, exp_avg,
                       exp_avg_sq, n, step_size, bc2_sqrt, beta1, beta2, eps, weight_decay, clamp, grad_scale);
    IVOSW_CHECK_LAUNCH();
    return IVOSW_OK;
}

extern "C" int ivosw_copy_f32(float* dst, const float* src, size_t n, ivosw_stream_t stream) {
    IVOSW_REQUIRE(dst && src, "null pointer");
    if (n == 0) return IVOSW_OK;
    hipError_t e = hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, as_stream(stream));
    if (e != hipSuccess) {
        set_error("ivosw_copy_f32: %s", hipGetErrorString(e));
        return IVOSW_ERR_LAUNCH;
    }
    return IVOSW_OK;
}

extern "C" int ivosw_replay_gather(const float* old_iou, const float* new_iou, const float* annotated,
                                   const float* next_annotated, const int64_t* action, const float* reward_step,
                                   const float* reward_done, const int64_t* idx, int B, int T, float* state,
                                   float* new_state, int64_t* action_out, float* reward_step_out, float* reward_done_out,
                                   ivosw_stream_t stream) {
    IVOSW_REQUIRE(old_iou && new_iou && annotated && next_annotated && action && reward_step && reward_done && idx,
                  "null input pointer");
    IVOSW_REQUIRE(state && new_state && action_out && reward_step_out && reward_done_out, "null output pointer");
    IVOSW_REQUIRE(B > 0 && T > 0, "B and T must be positive");
    hipLaunchKernelGGL(replay_gather_kernel, dim3(B), dim3(64), 0, as_stream(stream), old_iou, new_iou, annotated,
                       next_annotated, action, reward_step, reward_done, idx, B, T, state, new_state, action_out,
                       reward_step_out, reward_done_out);
    IVOSW_CHECK_LAUNCH();
    return IVOSW_OK;
}
