// Fused clamp + Adam over the flat parameter arena (K9), hard target sync (K10), replay gather (K11).
// Reference: models/agent.py:157-165 (clamp, optim.Adam(lr, weight_decay) step, target sync),
// datasets/agent_dataset.py:71-115 + train_agent.py:177-182 (minibatch assembly).
#include "adam.h"

namespace ivosw {

// torch.optim.Adam (non-amsgrad, coupled L2): g += wd*p; m.lerp_(g, 1-b1); v = b2*v + (1-b2)*g*g;
// p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps).  The clamp comes first (agent.py:157-159);
// grad_scale (=1/world) is applied before the clamp so the clamp sees the averaged gradient.
__global__ void clamp_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                  float* __restrict__ v, int n, float step_size, float bc2_sqrt, float beta1, float beta2,
                                  float eps, float wd, float clampv, float gscale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float mi = m[i], vi = v[i];
    p[i] = clamp_adam_elem(g[i], p[i], mi, vi, step_size, bc2_sqrt, beta1, beta2, eps, wd, clampv, gscale);
    m[i] = mi;
    v[i] = vi;
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

// The minibatch draw on the device (uniform with replacement, what torch.randint gave the eager loop): the index of slot b
// of draw number c is a splitmix64 finaliser of (seed, c, b) scaled to [0, n) by a 64 x 64 -> high-64 multiply — integer
// arithmetic only, so tests/ and a host mirror (ivos_w_amd.models.momory_pool.draw_indices) reproduce it bit for bit.  The
// draw counter lives on the device and is advanced by the LAST workgroup of the launch (a ticket, as in clamp_adam_dev), so a
// captured HIP graph replays a fresh minibatch each time with no host-side RNG launch in front of it (that launch and the
// bubble it left before the graph cost ~9.5 us of a 204 us step).
__global__ void replay_draw_gather_kernel(const float* __restrict__ old_iou, const float* __restrict__ new_iou,
                                          const float* __restrict__ ann, const float* __restrict__ nann,
                                          const int64_t* __restrict__ action, const float* __restrict__ rstep,
                                          const float* __restrict__ rdone, DrawState* __restrict__ ds, int n, int B, int T,
                                          int64_t* __restrict__ idx_out, float* __restrict__ state, float* __restrict__ new_state,
                                          int64_t* __restrict__ action_out, float* __restrict__ rstep_out,
                                          float* __restrict__ rdone_out) {
    const int b = blockIdx.x;
    const unsigned counter = ds->counter;
    const int64_t src = (int64_t)__umul64hi(draw_mix(ds->seed, counter, (unsigned)b), (unsigned long long)n);
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        const size_t s = (size_t)src * T + t, d = ((size_t)b * T + t) * 2;
        state[d] = old_iou[s];
        state[d + 1] = ann[s];
        new_state[d] = new_iou[s];
        new_state[d + 1] = nann[s];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        idx_out[b] = src;
        action_out[b] = action[src];
        rstep_out[b] = rstep[src];
        rdone_out[b] = rdone[src];
        if (atomicAdd(&ds->ticket, 1u) == gridDim.x - 1) {
            ds->counter = counter + 1;
            atomicExch(&ds->ticket, 0u);
        }
    }
}

// mask_quality[:] = pred.mean(1); state = stack([mask_quality, counts], 1) (utils/utils_agent.py:120-121) without leaving the
// device.  pred is a float64 [n_frames, n_obj] array filled with the float32 scores, so the mean is a float64 sum in numpy's
// order (add.reduce along the contiguous axis: sequential below 8 elements, else 8 interleaved partial sums combined
// pairwise plus a sequential tail) divided by n_obj; the Brain then sees float32(mean) (torch.Tensor(state), agent.py:176).
__global__ void quality_state_kernel(const float* __restrict__ scores, int n_obj, int n_frames, const float* __restrict__ counts,
                                     double* __restrict__ quality, float* __restrict__ state) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames) return;
    auto at = [&](int o) { return (double)scores[(size_t)o * n_frames + f]; };
    double sum;
    if (n_obj < 8) {
        sum = at(0);        // numpy's reduce starts from the first element (no 0.0 + x)
        for (int o = 1; o < n_obj; ++o) sum += at(o);
    } else {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = at(j);
        int i = 8;
        for (; i + 8 <= n_obj; i += 8)
            for (int j = 0; j < 8; ++j) r[j] += at(i + j);
        sum = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n_obj; ++i) sum += at(i);
    }
    const double q = sum / (double)n_obj;
    quality[f] = q;
    state[2 * f] = (float)q;
    state[2 * f + 1] = counts[f];
}

// Clamp + Adam with the step counter advanced by the SAME launch (a one-thread tick kernel in front of it cost a link of the
// step's launch chain, ~5 us): every thread reads the counter k left by the previous launch and evaluates step k+1's bias
// corrections itself (the float64 expressions of ivosw_clamp_adam: identical bits); the LAST workgroup to finish — a
// device-scope ticket, taken after the workgroup's own reads and writes — publishes k+1.  No other workgroup can still
// be reading the state at that point, and nothing but the ticket crosses workgroups, so no fence is needed.
template <bool VEC>
__global__ __launch_bounds__(1024) void clamp_adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                              float* __restrict__ v, int n, AdamDevState* __restrict__ st, float lr, float beta1,
                                                              float beta2, float eps, float wd, float clampv, float gscale) {
    const int step = st->step + 1;
    const double b1t = ipow((double)beta1, step), b2t = ipow((double)beta2, step);
    const float step_size = (float)((double)lr / (1.0 - b1t)), bc2_sqrt = (float)sqrt(1.0 - b2t);
    auto upd = [&](float gi, float pi, float& mi, float& vi) {
        return clamp_adam_elem(gi, pi, mi, vi, step_size, bc2_sqrt, beta1, beta2, eps, wd, clampv, gscale);
    };
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (VEC) {      // 16 bytes per lane and array; the n % 4 tail elements go to the first threads
        const int n4 = n >> 2;
        if (t < n4) {
            const float4 g4 = reinterpret_cast<const float4*>(g)[t];
            float4 p4 = reinterpret_cast<float4*>(p)[t], m4 = reinterpret_cast<float4*>(m)[t], v4 = reinterpret_cast<float4*>(v)[t];
            p4.x = upd(g4.x, p4.x, m4.x, v4.x); p4.y = upd(g4.y, p4.y, m4.y, v4.y);
            p4.z = upd(g4.z, p4.z, m4.z, v4.z); p4.w = upd(g4.w, p4.w, m4.w, v4.w);
            reinterpret_cast<float4*>(m)[t] = m4; reinterpret_cast<float4*>(v)[t] = v4; reinterpret_cast<float4*>(p)[t] = p4;
        } else if (t - n4 < (n & 3)) {
            const int i = 4 * n4 + (t - n4);
            float mi = m[i], vi = v[i];
            p[i] = upd(g[i], p[i], mi, vi);
            m[i] = mi; v[i] = vi;
        }
    } else if (t < n) {
        float mi = m[t], vi = v[t];
        p[t] = upd(g[t], p[t], mi, vi);
        m[t] = mi; v[t] = vi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (atomicAdd(&st->ticket, 1u) == gridDim.x - 1) {
            st->b1t = b1t; st->b2t = b2t; st->step = step; st->step_size = step_size; st->bc2_sqrt = bc2_sqrt;
            atomicExch(&st->ticket, 0u);
        }
    }
}

}  // namespace ivosw

using namespace ivosw;

extern "C" size_t ivosw_adam_state_bytes(void) { return sizeof(AdamDevState); }

extern "C" int ivosw_clamp_adam_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n, void* adam_state,
                                    float lr, float beta1, float beta2, float eps, float weight_decay, float clamp,
                                    float grad_scale, ivosw_stream_t stream) {
    IVOSW_REQUIRE(params && grads && exp_avg && exp_avg_sq && adam_state, "null pointer");
    IVOSW_ON_DEVICE_OF(params);
    IVOSW_REQUIRE(n > 0, "n must be positive");
    AdamDevState* sd = static_cast<AdamDevState*>(adam_state);
    // few, large workgroups: the tickets of one launch serialise on one address (708 of them took longer than the update)
    const bool vec = ((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads) | reinterpret_cast<uintptr_t>(exp_avg) |
                       reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15) == 0;
    if (vec)
        hipLaunchKernelGGL(clamp_adam_dev_kernel<true>, dim3((n / 4 + 3 + 1023) / 1024), dim3(1024), 0, as_stream(stream), params, grads,
                           exp_avg, exp_avg_sq, n, sd, lr, beta1, beta2, eps, weight_decay, clamp, grad_scale);
    else
        hipLaunchKernelGGL(clamp_adam_dev_kernel<false>, dim3((n + 1023) / 1024), dim3(1024), 0, as_stream(stream), params, grads,
                           exp_avg, exp_avg_sq, n, sd, lr, beta1, beta2, eps, weight_decay, clamp, grad_scale);
    IVOSW_CHECK_LAUNCH();
    return IVOSW_OK;
}

extern "C" int ivosw_quality_state(const float* scores, int n_obj, int n_frames, const float* counts, double* quality,
                                   float* state, ivosw_stream_t stream) {
    IVOSW_REQUIRE(scores && counts && quality && state, "null pointer");
    IVOSW_ON_DEVICE_OF(state);
    IVOSW_REQUIRE(n_obj > 0 && n_frames > 0, "n_obj and n_frames must be positive");
    hipLaunchKernelGGL(quality_state_kernel, dim3((n_frames + 63) / 64), dim3(64), 0, as_stream(stream), scores, n_obj, n_frames,
                       counts, quality, state);
    IVOSW_CHECK_LAUNCH();
    return IVOSW_OK;
}

extern "C" int ivosw_clamp_adam(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n, int step,
                                float lr, float beta1, float beta2, float eps, float weight_decay, float clamp,
                                float grad_scale, ivosw_stream_t stream) {
    IVOSW_REQUIRE(params && grads && exp_avg && exp_avg_sq, "null pointer");
    IVOSW_ON_DEVICE_OF(params);
    IVOSW_REQUIRE(n > 0 && step >= 1, "n must be positive and step >= 1");
    const double bc1 = 1.0 - ipow((double)beta1, step);
    const double bc2 = 1.0 - ipow((double)beta2, step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    hipLaunchKernelGGL(clamp_adam_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), params, grads, exp_avg,
                       exp_avg_sq, n, step_size, bc2_sqrt, beta1, beta2, eps, weight_decay, clamp, grad_scale);
    IVOSW_CHECK_LAUNCH();
    return IVOSW_OK;
}

extern "C" int ivosw_copy_f32(float* dst, const float* src, size_t n, ivosw_stream_t stream) {
    IVOSW_REQUIRE(dst && src, "null pointer");
    IVOSW_ON_DEVICE_OF(dst);
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
    IVOSW_ON_DEVICE_OF(state);
    IVOSW_REQUIRE(B > 0 && T > 0, "B and T must be positive");
    hipLaunchKernelGGL(replay_gather_kernel, dim3(B), dim3(64), 0, as_stream(stream), old_iou, new_iou, annotated,
                       next_annotated, action, reward_step, reward_done, idx, B, T, state, new_state, action_out,
                       reward_step_out, reward_done_out);
    IVOSW_CHECK_LAUNCH();
    return IVOSW_OK;
}

extern "C" size_t ivosw_replay_draw_state_bytes(void) { return sizeof(DrawState); }

extern "C" unsigned long long ivosw_replay_draw_index(unsigned long long seed, unsigned counter, unsigned slot, int n) {
    if (n <= 0) return 0;
    return (unsigned long long)(((unsigned __int128)draw_mix(seed, counter, slot) * (unsigned long long)n) >> 64);
}

extern "C" int ivosw_replay_draw_gather(const float* old_iou, const float* new_iou, const float* annotated,
                                        const float* next_annotated, const int64_t* action, const float* reward_step,
                                        const float* reward_done, void* draw_state, int n, int B, int T, int64_t* idx_out,
                                        float* state, float* new_state, int64_t* action_out, float* reward_step_out,
                                        float* reward_done_out, ivosw_stream_t stream) {
    IVOSW_REQUIRE(old_iou && new_iou && annotated && next_annotated && action && reward_step && reward_done && draw_state,
                  "null input pointer");
    IVOSW_REQUIRE(idx_out && state && new_state && action_out && reward_step_out && reward_done_out, "null output pointer");
    IVOSW_ON_DEVICE_OF(state);
    IVOSW_REQUIRE(n > 0 && B > 0 && T > 0, "n, B and T must be positive");
    hipLaunchKernelGGL(replay_draw_gather_kernel, dim3(B), dim3(64), 0, as_stream(stream), old_iou, new_iou, annotated,
                       next_annotated, action, reward_step, reward_done, static_cast<DrawState*>(draw_state), n, B, T, idx_out,
                       state, new_state, action_out, reward_step_out, reward_done_out);
    IVOSW_CHECK_LAUNCH();
    return IVOSW_OK;
}
