/* ivosw.h — C ABI of libivosw_hip.so: the MI355X (gfx950) hot path of IVOS-W.
 *
 * The reference (svip-lab/IVOS-W) is pure Python over PyTorch ops and has NO FFI/plugin interface
 * (SURVEY.md §8b); the boundary it offers is the Python class surface models.agent.{Brain,Agent},
 * models.assessment.{Encoder,AssessNet}.  Each entry point below therefore cites the reference
 * *method* whose arithmetic it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - extern "C", plain pointers + sizes.  No torch / HIP types in signatures: a stream is passed as
 *     void* (it is a hipStream_t; NULL = the null stream).
 *   - Every pointer is a DEVICE pointer owned by the caller unless a parameter says "host".
 *     Workspaces are sized by the *_ws_bytes queries and handed in by the caller; the library keeps no
 *     per-call state in them.  Two exceptions, both documented at their entry points: ivosw_p2p_alloc /
 *     ivosw_p2p_free allocate the fine-grained peer-to-peer arena (a kind of memory neither torch nor a
 *     caller can allocate), and ivosw_assess_forward keeps ONE helper stream + two events per device for
 *     the two-stream split of a batch (created on first use, never freed).
 *   - All calls are asynchronous on `stream`; nothing synchronises the device (except the entries that
 *     return a host value: ivosw_p2p_error, ivosw_profile_*).
 *   - Return 0 on success, negative on error; ivosw_last_error() gives a thread-local message.
 *   - Not thread-safe per workspace; distinct workspaces on distinct streams are independent.
 */
#ifndef IVOSW_H
#define IVOSW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IVOSW_OK 0
#define IVOSW_ERR_ARG (-1)      /* bad argument (null pointer, non-positive size, bad enum)   */
#define IVOSW_ERR_WS (-2)       /* workspace too small                                         */
#define IVOSW_ERR_LAUNCH (-3)   /* HIP launch / runtime error                                  */

#define IVOSW_F32 0             /* fp32 operands, fp32 accumulate (parity mode)                */
#define IVOSW_BF16 1            /* bf16 operands, fp32 accumulate (throughput mode)            */
#define IVOSW_F32X3 2           /* fp32 activations and weights, contractions as THREE bf16 MFMA passes: x = hi + lo with hi, lo in bf16,
                                 * a b ~ ah bh + ah bl + al bh, fp32 accumulate (error ~ 2^-17 per product: scores within 1e-4 rtol of the
                                 * reference like IVOSW_F32, at 16 / 3 of its matrix rate).  Layout and workspaces are IVOSW_F32's;
                                 * the packed arena holds the conv weights pre-split, so it must be packed with this dtype.            */

/* Brain parameter arena: the 10 tensors of Brain.state_dict() concatenated in state_dict order
 * (models/agent.py:13-31): encoder_fc1.{weight[128,2],bias[128]}, encoder_fc2.{weight[128,128],bias[128]},
 * lstm_cell.{weight_ih[512,128],weight_hh[512,128]}, decoder_fc1.{weight[128,256],bias[128]},
 * decoder_fc2.{weight[1,128],bias[1]}.                                                         */
#define IVOSW_BRAIN_NPARAMS 180993

typedef void* ivosw_stream_t;

const char* ivosw_last_error(void);
int ivosw_version(void);

/* ------------------------------------------------------------------ agent: Brain (K7) --------- */
/* Replaces Brain.forward (models/agent.py:33-64): x [N,T,2] fp32 -> q [N,T] fp32.
 * One shared bias-free LSTM cell runs forward and backward over the T frames from a zero state. */
size_t ivosw_brain_ws_bytes(int N, int T);
int ivosw_brain_forward(const float* params, const float* x, int N, int T, float* q,
                        void* ws, size_t ws_bytes, ivosw_stream_t stream);
/* Replaces Q.argmax() in Agent.action (models/agent.py:187-188): first maximum per row -> idx[N]. */
int ivosw_brain_argmax(const float* q, int N, int T, int64_t* idx, ivosw_stream_t stream);

/* ------------------------------------------------------------------ agent: DQN step (K8-K10) -- */
/* Replaces the arithmetic of Agent.update_agent (models/agent.py:128-155):
 *   a* = argmax policy(s'); Qn = target(s')[a*]; y1 = gamma*Qn + 0.1*r_step; y2 = 0.1*r_done;
 *   Qsa = policy(s)[action]; loss = mean((Qsa-y1)^2) + mean((Qsa-y2)^2); grads = dLoss/dpolicy.
 * state/new_state [B,T,2] fp32, action [B] int64, reward_* [B] fp32.  grads [NPARAMS] is overwritten,
 * *loss is a device float.  The grads are NOT clamped here so that a data-parallel caller can
 * all-reduce them first (RCCL) and clamp afterwards.                                             */
size_t ivosw_dqn_ws_bytes(int B, int T);
int ivosw_dqn_loss_grad(const float* policy, const float* target,
                        const float* state, const float* new_state, const int64_t* action,
                        const float* reward_step, const float* reward_done,
                        int B, int T, float gamma, float* grads, float* loss,
                        void* ws, size_t ws_bytes, ivosw_stream_t stream);
/* Replaces grad.clamp_(-1,1) + optim.Adam.step (models/agent.py:157-160, :101): g = clamp(grad*grad_scale);
 * g += wd*p; m,v update; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).  step = t >= 1.
 * grad_scale = 1/world_size after a sum all-reduce, 1 otherwise.                                 */
int ivosw_clamp_adam(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n,
                     int step, float lr, float beta1, float beta2, float eps, float weight_decay,
                     float clamp, float grad_scale, ivosw_stream_t stream);
/* The same update with Adam's step counter and bias corrections kept on the device (adam_state: ivosw_adam_state_bytes()
 * bytes, zero-initialised = step 0; the step counter is the int32 at byte offset 16, which is all a caller writes to resume
 * from step k), so that a HIP graph that captured the call replays correctly: every call (or replay) advances the step by one.                                                 */
size_t ivosw_adam_state_bytes(void);
int ivosw_clamp_adam_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n, void* adam_state,
                         float lr, float beta1, float beta2, float eps, float weight_decay, float clamp,
                         float grad_scale, ivosw_stream_t stream);
/* Replaces target_net.load_state_dict(policy_net.state_dict()) (models/agent.py:163-165).        */
int ivosw_copy_f32(float* dst, const float* src, size_t n, ivosw_stream_t stream);

/* ------------------------------------------------------------------ one-shot P2P all-reduce --- */
/* The data-parallel DQN step's gradient all-reduce (Agent.update_agent under torch.distributed; the reference is single
 * GPU) over xGMI peer-to-peer writes instead of a ring: every rank pushes its n floats into its slot of every peer's arena and
 * raises a flag there (7 links in parallel, one hop), then sums the `world` slots it received in rank order (bit-identical
 * on all ranks).  The arena is FINE-GRAINED device memory (coherent inside kernels across GPUs) — the one device allocation the
 * library makes itself: ivosw_p2p_alloc (returns the pointer and an IPC handle of ivosw_p2p_handle_bytes() bytes to send to
 * the peers) / ivosw_p2p_free; peers map it with ivosw_p2p_open / unmap with ivosw_p2p_close.  ivosw_p2p_allreduce:
 * arenas = HOST array of `world` device pointers (own arena at [rank]); epoch = 1, 2, 3, ... identical on every rank, one per
 * call; out may alias grads (16-byte aligned); a peer that does not arrive within timeout_ms sets the arena's error word
 * (ivosw_p2p_error) instead of hanging the GPU.  Two kernels per call, no host synchronisation.                           */
size_t ivosw_p2p_arena_bytes(int world, size_t n);
size_t ivosw_p2p_handle_bytes(void);
int ivosw_p2p_alloc(size_t bytes, void** arena, void* ipc_handle, size_t ipc_handle_bytes);
int ivosw_p2p_open(const void* ipc_handle, void** peer_arena);
int ivosw_p2p_close(void* peer_arena);
int ivosw_p2p_free(void* arena);
int ivosw_p2p_error(const void* arena, int* error);
int ivosw_p2p_allreduce(const float* grads, float* out, int n, int rank, int world, void* const* arenas, unsigned epoch,
                        int timeout_ms, ivosw_stream_t stream);
/* ivosw_p2p_allreduce followed by ivosw_clamp_adam(grad_scale = 1/world) as TWO launches: push, then wait + rank-ordered sum + clamp
 * + Adam (the reference's clamp / optim.Adam step, models/agent.py:157-160, on the gradient averaged over the ranks).  grads_out
 * (NULL allowed, may alias grads) receives the summed gradient; `step` is the 1-based Adam step.  On a timeout nothing is updated
 * (one verdict for the whole grid: workgroup 0 alone declares it) and the arena's error word is set (ivosw_p2p_error); once the
 * word is set every later call returns without touching its outputs until the arena is re-created.                                                                        */
int ivosw_p2p_allreduce_clamp_adam(const float* grads, float* grads_out, int n, int rank, int world, void* const* arenas,
                                   unsigned epoch, int timeout_ms, float* params, float* exp_avg, float* exp_avg_sq, int step,
                                   float lr, float beta1, float beta2, float eps, float weight_decay, float clamp,
                                   ivosw_stream_t stream);

/* ------------------------------------------------------------------ replay gather (K11) ------- */
/* Replaces DataLoader shuffle+collate of memory_pool.csv rows (datasets/agent_dataset.py:71-115,
 * train_agent.py:177-182) for a device-resident SoA replay buffer: columns [cap,T] fp32 and [cap]
 * scalars; idx [B] int64 -> state/new_state [B,T,2] and the [B] columns of the minibatch.        */
int ivosw_replay_gather(const float* old_iou, const float* new_iou, const float* annotated,
                        const float* next_annotated, const int64_t* action, const float* reward_step,
                        const float* reward_done, const int64_t* idx, int B, int T,
                        float* state, float* new_state, int64_t* action_out, float* reward_step_out,
                        float* reward_done_out, ivosw_stream_t stream);
/* The same gather with the minibatch indices DRAWN ON THE DEVICE (uniform over [0, n) with replacement — the role of the
 * DataLoader's shuffle, train_agent.py:177-182 — from a counter-based generator), so that a captured HIP graph of a training
 * step needs no host-side RNG launch: draw_state is ivosw_replay_draw_state_bytes() bytes on the device {uint64 seed at byte
 * 0, uint32 draw counter at byte 8, 4 bytes the library owns (zero)}; every call (or graph replay) uses the current counter
 * and advances it by one.  Slot b of draw c reads row ivosw_replay_draw_index(seed, c, b, n) — the host mirror of the device
 * arithmetic (integer only, bit-exact).  idx_out [B] int64 receives the rows that were drawn.                                */
size_t ivosw_replay_draw_state_bytes(void);
unsigned long long ivosw_replay_draw_index(unsigned long long seed, unsigned counter, unsigned slot, int n);
int ivosw_replay_draw_gather(const float* old_iou, const float* new_iou, const float* annotated,
                             const float* next_annotated, const int64_t* action, const float* reward_step,
                             const float* reward_done, void* draw_state, int n, int B, int T, int64_t* idx_out,
                             float* state, float* new_state, int64_t* action_out, float* reward_step_out,
                             float* reward_done_out, ivosw_stream_t stream);

/* One single-GPU training step of Agent.update_agent's loop (models/agent.py:128-160 minus the host coin of the target sync) as ONE
 * call and EIGHT launches: ivosw_replay_draw_gather + ivosw_dqn_loss_grad + ivosw_clamp_adam_dev with the minibatch draw and gather
 * folded into the encoder launch and the split-K slab reduction folded into clamp + Adam.  Same arithmetic, same orders of summation:
 * parameters, Adam state, gradient arena, loss, idx_out / state / new_state / action_out / reward_*_out and both device counters end
 * up bit-identical to the three separate calls.  `policy` is updated in place.  (When a tunable takes the step off the fused launch
 * chain the entry runs the three calls itself.)                                                                                  */
int ivosw_dqn_step_drawn(float* policy, const float* target, const float* old_iou, const float* new_iou, const float* annotated,
                         const float* next_annotated, const int64_t* action, const float* reward_step, const float* reward_done,
                         void* draw_state, int n, int B, int T, float gamma, int64_t* idx_out, float* state, float* new_state,
                         int64_t* action_out, float* reward_step_out, float* reward_done_out, float* grads, float* loss, void* ws,
                         size_t ws_bytes, float* exp_avg, float* exp_avg_sq, void* adam_state, float lr, float beta1, float beta2,
                         float eps, float weight_decay, float clamp, float grad_scale, ivosw_stream_t stream);

/* ------------------------------------------------------------------ assessment front end ------ */
/* Replaces (tp>0.5) + AssessNet.all2yxhw(scale=1.5) (models/assessment.py:165-166,110-161) with no D2H:
 * tp [B,H,W] fp32 -> yxhw [B,4] fp32 (y,x,h,w).  scratch: B*4 int32.                              */
int ivosw_mask_bbox(const float* tp, int B, int H, int W, float* yxhw, int32_t* scratch,
                    ivosw_stream_t stream);
/* Replaces get_ROI_grid + 2x F.grid_sample + the (f-mean)/std of Encoder.forward
 * (models/assessment.py:75-108,173-174,47): tf [B,3,H,W], tp [B,H,W] fp32 NCHW, yxhw [B,4] ->
 * roi [B,256,256,4] NHWC (R,G,B normalised, P raw) in `dtype`.                                    */
int ivosw_roi_sample(const float* tf, const float* tp, const float* yxhw, int B, int H, int W,
                     int dtype, void* roi, ivosw_stream_t stream);

/* ------------------------------------------------------------------ assessment network -------- */
/* Packed weights (BN folded in fp32, K-major repack, stem conv1|conv1_p concatenated along Cin).
 * `tensors` is a HOST array of 326 DEVICE pointers, one per entry of AssessNet.state_dict() in
 * state_dict order (models/assessment.py:12-71 + torchvision ResNet-50); fp32
 * tensors as stored, num_batches_tracked entries are ignored (may be NULL).                       */
#define IVOSW_ASSESS_NTENSORS 326
size_t ivosw_assess_packed_bytes(int dtype);
int ivosw_assess_pack(void* packed, int dtype, const void* const* tensors, int ntensors,
                      ivosw_stream_t stream);
/* Replaces AssessNet.forward (models/assessment.py:164-182): tf [B,3,H,W], tp [B,H,W] fp32 ->
 * scores [B] fp32.  Frames are processed in chunks of `chunk` frames at res2, doubling per stage
 * (<=0: library default, 256 bf16 / 128 fp32 modes); the chunk bounds the workspace (ivosw_assess_ws_bytes).
 * tap_stage/tap_out (debug, tests): 0 = none; 1 roi[.,256,256,4] 2 stem[.,128,128,64] 3 pool[.,64,64,64]
 * 4..7 res2..res5 outputs, NHWC in `dtype`; 8 pooled [.,2048] fp32.  Only with B <= chunk.        */
size_t ivosw_assess_ws_bytes(int dtype, int B, int H, int W, int chunk);
/* 1 when ivosw_assess_forward(dtype, B, chunk) runs the batch as two independent halves on two streams (bf16, default chunk,
 * B >= 64, tunables STREAMS2 / STREAMS2_MIN; the workspace query above already covers both halves), else 0.  Scores do not depend on it. */
int ivosw_assess_split(int dtype, int B, int chunk);
int ivosw_assess_forward(const void* packed, int dtype, const float* tf, const float* tp,
                         int B, int H, int W, float* scores, void* ws, size_t ws_bytes, int chunk,
                         int tap_stage, void* tap_out, ivosw_stream_t stream);
/* The same forward for the O objects of ONE n-frame video without replicating the frames (recommend_frame scores one object
 * at a time on the same all_F, utils/utils_agent.py:118-119): tf [n_frames,3,H,W]; unit u = obj * n_frames + frame reads
 * frame u % n_frames and the mask plane at masks + obj * mask_stride_obj + frame * mask_stride_frame (strides in elements;
 * for the reference's all_P [n,O+1,H,W]: masks = all_P + H*W, stride_frame = (O+1)*H*W, stride_obj = H*W).
 * scores [n_obj * n_frames] fp32, object-major.  Workspace: ivosw_assess_ws_bytes with B = n_obj * n_frames.            */
int ivosw_assess_forward_objects(const void* packed, int dtype, const float* tf, int n_frames, const float* masks,
                                 long mask_stride_frame, long mask_stride_obj, int n_obj, int H, int W, float* scores,
                                 void* ws, size_t ws_bytes, int chunk, ivosw_stream_t stream);
/* Replaces `mask_quality[:] = pred.mean(1); state = np.stack([mask_quality, counts], 1)` (utils/utils_agent.py:120-121) on
 * the device: scores [n_obj][n_frames] fp32 (as ivosw_assess_forward_objects writes them), counts [n_frames] fp32 ->
 * quality [n_frames] float64 (numpy's float64 mean of the float32 predictions, same summation order) and
 * state [n_frames,2] fp32 = (float32(quality), counts), the Brain's input.                                              */
int ivosw_quality_state(const float* scores, int n_obj, int n_frames, const float* counts, double* quality,
                        float* state, ivosw_stream_t stream);
/* Kernel-name patterns of the dominant kernel family (the tower's contraction kernels) for profiling. */
const char* ivosw_assess_dominant_kernel(int dtype);

/* ------------------------------------------------------------------ J / F metrics - */
/* Replaces davisinteractive.metrics.batched_jaccard / batched_f_measure as utils/misc.py:136-160
 * (sequence_metric) calls them.  gt, pred: device uint8 label maps [N,H,W]; obj_ids: HOST array of the
 * n_obj (<= 32) label values to score; bound_pix = the F-measure tolerance in pixels, as the reference
 * derives it (bound_th if >= 1 else ceil(bound_th * |(H,W)|_2), bound_th = 0.008; <= 32).  Writes the six
 * INTEGER counts per (frame, object) to the device array counts [N][n_obj][6] =
 *   { |gt & pred|, |gt | pred|, #pred-boundary, #gt-boundary, #pred-boundary within dil(gt-boundary),
 *     #gt-boundary within dil(pred-boundary) };
 * the host forms J and F from them with the reference's float64 expressions (ivos_w_amd/metrics.py).    */
size_t ivosw_jf_ws_bytes(int N, int H, int W, int n_obj);
int ivosw_jf_counts(const uint8_t* gt, const uint8_t* pred, int N, int H, int W, const uint8_t* obj_ids,
                    int n_obj, int bound_pix, int64_t* counts, void* ws, size_t ws_bytes, ivosw_stream_t stream);

/* ------------------------------------------------------------------ segmentation epilogue (8f-4) */
/* Replaces, per frame batch, F.interpolate(logits, (H,W), 'bilinear', align_corners=True) -> argmax(dim=1)
 * [-> .float()] and the final torch.softmax(cat(probs), 1) of utils/utils_manet.py:78-84,110-116,146-151,
 * 160-161.  logits: device [n,C,hs,ws] fp32.  Outputs (each optional, at least one): probs, element
 * (f,c,y,x) written at probs[f*probs_stride_n + c*probs_stride_c + y*W + x] (strides in elements: pass
 * C*H*W, H*W for the reference's [n,C,H,W]; n_total*H*W as the channel stride stores all_P object-major);
 * label_i64 / label_u8 / label_f32 [n,H,W] = argmax over channels (first maximum).                    */
int ivosw_seg_epilogue(const float* logits, int n, int C, int hs, int ws, int H, int W, float* probs,
                       long probs_stride_n, long probs_stride_c, int64_t* label_i64, uint8_t* label_u8,
                       float* label_f32, ivosw_stream_t stream);

/* ------------------------------------------------------------------ launch-sequence capture ---- */
/* HIP-graph capture of any sequence of the calls above on one stream (not in the reference: it replaces the ~40 host
 * launches per Agent.update_agent step, models/agent.py:128-160, by ONE hipGraphLaunch).  begin puts `stream` (non-null)
 * into capture; every ivosw_* call made on it until end is recorded instead of executed; end instantiates the graph and
 * reports its kernel-node count.  All device pointers passed during capture must stay valid for the life of the handle;
 * per-step scalars that change (Adam's step) live on the device (ivosw_clamp_adam_dev).  The handle is a host object.  */
typedef void* ivosw_graph_t;
int ivosw_graph_begin(ivosw_stream_t stream);
int ivosw_graph_end(ivosw_stream_t stream, ivosw_graph_t* out, int* kernel_nodes);
int ivosw_graph_launch(ivosw_graph_t g, ivosw_stream_t stream);
int ivosw_graph_destroy(ivosw_graph_t g);

/* ------------------------------------------------------------------ measurement hooks ---------- */
/* Not part of the reference surface: bench.py's roofline leg.  Between start and stop every launch of
 * the dominant kernel family (conv_igemm*, conv1x1_wide*, conv3x3_patch*, bneck*, stem_pool*) is bracketed by hipEvents on the launch stream; stop
 * synchronises those events and returns the summed kernel time (ms) and the launch count.          */
int ivosw_profile_start(void);
/* Span mode (what bench.py's `roofline` uses, inside the timed region): ONE event pair around each uninterrupted run of
 * tower launches (stem .. the last res5 kernel of a pass; the ROI sampler and the pool+fc kernel are outside).  stop
 * synchronises the events and returns the summed span time (ms), the number of spans and of family launches in them. */
int ivosw_profile_span_start(void);
int ivosw_profile_span_stop(double* total_ms, int* spans, int* launches);
int ivosw_profile_stop(double* total_ms, int* launches);
/* Text table (one line per distinct conv layer shape: calls, avg us, TFLOP/s, GB/s) of the launches recorded
 * since ivosw_profile_start; call before ivosw_profile_stop.  buf is a HOST buffer of `cap` bytes.       */
int ivosw_profile_report(char* buf, size_t cap);
/* Tuning/test hook: override a named integer tunable (otherwise read from the environment variable
 * IVOSW_TUNE_<KEY>).  Keys: FUSE (1 = whole-bottleneck fused kernels in bf16 mode, 0 = layer by layer), WS, NK. */
int ivosw_tune_set(const char* key, int value);
/* 1 when the library was built with -DIVOSW_ABLATION=1 (the ablation switches IVOSW_DEBUG_CONV / BDBG are compiled in and can
 * skip MFMAs, loads or stores); the default build returns 0 and contains none of them.  bench.py refuses to run on 1.   */
int ivosw_ablation_build(void);
/* Profiling aid (bench.py roofline.sclk_mhz): one wave spins for spin_us of wall time and writes {shader cycles, 100 MHz wall
 * ticks} of that interval to out2 (2 x uint64, device): launched on a side stream beside the measured work, cycles / ticks x 100 MHz
 * is the shader clock the chip ran at under that load.                                                                       */
int ivosw_clock_probe(unsigned long long* out2, int spin_us, ivosw_stream_t stream);

/* The forward entry points refuse an arena packed for another dtype.  The tag of an arena is known to this process when it packed the
 * arena itself, or from the arena's own 4-byte device tag read on the first forward call that sees the address.  Call this when the
 * memory behind `packed` is freed or re-filled (device copy) with an arena of another precision, so that the cached tag is dropped. */
int ivosw_assess_forget(const void* packed);

/* The tuning probes (single-kernel launches with phase stamps, micro-benchmark kernels) are not part of this ABI: include/ivosw_probe.h,
 * libivosw_probe.so.                                                                                                              */

#ifdef __cplusplus
}
#endif
#endif /* IVOSW_H */
