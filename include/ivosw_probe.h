/* Tuning probes of libivosw — NOT part of the product ABI (include/ivosw.h, libivosw_hip.so).
 *
 * libivosw_probe.so is a superset build of the same sources with -DIVOSW_PROBES: every entry of ivosw.h plus the entries below -
 * single-kernel launches with s_memtime phase stamps and the micro-benchmark contraction of round 5.  tools/ and one GPU test load it;
 * nothing a reference maintainer binds lives here.                                                                               */
#ifndef IVOSW_PROBE_H
#define IVOSW_PROBE_H
#include "ivosw.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Tuning probe: the fused Brain forwards that follow stamp s_memtime at four points (step start, MFMAs done, state update
 * done, barrier passed) of recurrence step T/2 (slots 0-3) and at kernel entry / weights in registers / last step done (4-6),
 * per workgroup, into ts [workgroups,8] uint64 (device); NULL = off.                                                     */
int ivosw_lstm_probe(unsigned long long* ts, unsigned long long* ts_bwd);   /* ts_bwd: the same for the BPTT kernel of ivosw_dqn_loss_grad */
/* Tuning probe: ONE fused bottleneck (wd/bd NULL: identity block, else the stride-1 downsample block)
 * (x [B,H,W,Cin] bf16 -> y [B,H,W,4*Cmid] bf16; weights packed
 * K-major bf16 with fp32 biases as ivosw_assess_pack lays them out) with s_memtime stamps at the phase
 * boundaries of every workgroup: ts [B*(H/16)*(W/16)][16] uint64 on the device (may be NULL).             */
int ivosw_bneck_probe(const void* x, void* y, const void* wa, const float* ba, const void* wb, const float* bb,
                      const void* wc, const float* bc, const void* wd, const float* bd, const void* zeros,
                      int B, int H, int W, int Cin, int Cmid,
                      unsigned long long* ts, ivosw_stream_t stream);

/* Tuning probe: ONE wide fused bottleneck (res4 identity block: H = W = 16, Cin = 1024, Cmid = 256), weights K-major
 * packed bf16; `frag` is device scratch for their fragment-ordered copies; ts [B][8] uint64 phase stamps or NULL.   */
int ivosw_bneck_wide_probe(const void* x, void* y, const void* wa, const float* ba, const void* wb, const float* bb,
                           const void* wc, const float* bc, void* frag, int B, int H, int W, int Cin, int Cmid,
                           unsigned long long* ts, ivosw_stream_t stream);

/* Tuning probe (round 5): the one-wave-per-SIMD, 4 x 4-register-tile contraction of the attainable-roof measurement
 * (csrc/gemm_bt.h; tools/ubench/gemm_tile_bench.hip times it): C [M][N] bf16 = act(A [M][K] . B [N][K]^T + bias [N]),
 * both operands K-major bf16, fp32 accumulation, act = ReLU when relu != 0.  M % 256 == 0, N % 256 == 0, K % 32 == 0, K >= 32.
 * No reference function stands behind it: the tower's 1x1 convolutions (models/assessment.py:58-61 through torchvision's
 * Bottleneck) are contractions of exactly this form, and the product path runs them in the 8-wave kernels that reach the same rate.
 * ts [M/256 * N/256][4] uint64 (may be NULL): s_memtime at start / after the K loop / at the end, s_memrealtime span.             */
int ivosw_gemm_bt_probe(const void* A, const void* B, const float* bias, void* C, int M, int N, int K, int relu,
                        unsigned long long* ts, ivosw_stream_t stream);

/* Tuning probe: ONE launch of the res2 stage kernel (the three bottlenecks of res2 + res3's forwarded conv1; reference
 * models/assessment.py:58-59) on x [B,64,64,64] bf16 with the weights of a packed bf16 arena (ivosw_assess_pack); y [B,64,64,256]
 * (y_s2 != 0: the even pixels, [B,32,32,256]), t1out [B,64,64,128]; ts [B*32][16] uint64 phase stamps or NULL.            */
int ivosw_res2_stage_probe(const void* packed, const void* x, void* y, void* t1out, int B, int y_s2, unsigned long long* ts,
                           ivosw_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* IVOSW_PROBE_H */
