#!/usr/bin/env python
"""bench.py — throughput of the IVOS-W hot path on MI355X (one process per GPU).

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

Default workload = BASELINE.json configs[1]: the assessment CNN (AssessNet.forward) on a batch of 256
synthetic 480p (frame, mask) pairs per GPU, bf16 operands / fp32 accumulate, inputs resident in HBM when
the clock starts.  One "step" = one AssessNet.forward over the batch.  `value` = assessed frames/s over all
ranks (frames are independent: weak scaling, no data-path collective).

The same JSON line also carries, as `dqn`, the second half of BASELINE's metric: Double-DQN agent steps/s
(configs[2]/[3]: replay 50k transitions, minibatch 128 per GPU, T=25; device-resident replay gather ->
3 forwards + loss + BPTT -> [RCCL all-reduce of the 724 KB gradient arena when N>1] -> clamp+Adam ->
target-sync coin flip).  `--workload dqn` makes that the headline `value` instead.

roofline: dominant kernel family = the tower's contraction kernels, conv_igemm* | conv1x1_wide* | conv3x3_patch* (layer by layer), stem_pool* and bneck* (whole
res2 bottlenecks fused) (bound: bf16 MFMA, 2.5 PFLOP/s dense).  achieved = algorithmic conv FLOPs per launch /
average launch duration, timed with HIP events on the launch stream inside the library
(ivosw_profile_start/stop) over extra steps that run right after the timed region, so the events do not perturb
`value`.  cpu_baseline: the oracle (torch-CPU restatement of the reference path, kind "port") on a bounded
sample, rank 0, N=1 only.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from ivos_w_amd import _lib as L            # noqa: E402
from ivos_w_amd import synth                # noqa: E402

GFLOP_PER_FRAME = 10.779365376              # 5 389 682 688 MAC x 2, conv stack + fc
CONV_LAUNCHES_PER_FRAME_CHUNK = 54           # stem + 53 tower convs, per chunk
PEAK_BF16_TFLOPS = 2500.0                    # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3


class AD(dict):
    __getattr__ = dict.__getitem__


def agent_cfg():
    return AD(phase="train", data=AD(subset="train"),
              agent=AD(memory_size=100000, gamma=0.95, eps_start=0.7, eps_end=0.25, eps_decay=500,
                       update_rate=0.05, lr=5e-6, weight_decay=5e-4))        # configs/config.yaml agent block


def dist_setup(n):
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    if n > 1 and world != n:
        raise SystemExit(f"--gpus {n} needs torch.distributed.run with nproc-per-node {n} (WORLD_SIZE={world})")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist_.init_process_group("nccl", device_id=dev)
        dist = dist_
    return rank, world, dev, dist


def timed(fn, steps, warmup, dev, dist):
    for _ in range(warmup):
        fn()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def build_assess(args, rank, dev):
    from ivos_w_amd.models.assessment import AssessNet
    net = AssessNet(precision=args.precision, chunk=args.chunk)
    sd = synth.assessnet_state_dict(0)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    net.to(dev).eval()
    # SURVEY §8(d) A2 inputs; generated per 32 frames to bound host memory, rank-offset seed
    tf = torch.empty(args.batch, 3, 480, 854, dtype=torch.float32, device=dev)
    tp = torch.empty(args.batch, 480, 854, dtype=torch.float32, device=dev)
    for i in range(0, args.batch, 32):
        n = min(32, args.batch - i)
        a, b = synth.assess_inputs(n, seed=1234 + 1000 * rank + i)
        tf[i:i + n].copy_(torch.from_numpy(a))
        tp[i:i + n].copy_(torch.from_numpy(b))
    return net, tf, tp


def bench_assess(args, rank, world, dev, dist):
    net, tf, tp = build_assess(args, rank, dev)
    out = {}

    def step():
        out["s"] = net(tf, tp)
    dt = timed(step, args.steps, args.warmup, dev, dist)
    assert torch.isfinite(out["s"]).all()
    fps = world * args.batch * args.steps / dt
    # roofline leg: HIP events around every conv launch, separate steps
    lib = L.lib()
    psteps = max(1, min(3, args.steps))
    lib.ivosw_profile_start()
    for _ in range(psteps):
        step()
    if args.layer_report and rank == 0:
        buf = ctypes.create_string_buffer(1 << 16)
        lib.ivosw_profile_report(buf, len(buf))
        with open(args.layer_report, "w") as f:
            f.write(f"# per-layer conv timing (HIP events, {psteps} steps, batch {args.batch}, chunk {args.chunk or 'default'}, {args.precision})\n")
            f.write(buf.value.decode())
    tot, cnt = ctypes.c_double(0), ctypes.c_int(0)
    lib.ivosw_profile_stop(ctypes.byref(tot), ctypes.byref(cnt))
    conv_ms = tot.value / psteps
    launches = cnt.value // psteps
    flops_step = GFLOP_PER_FRAME * 1e9 * args.batch
    achieved = flops_step / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else PEAK_F32_TFLOPS
    # HBM bytes per launch of the same kernel family: from the committed rocprofv3 --pmc passes of this very command
    # (tools/profile_round.sh -> profiles/pmc_traffic_latest.json); only quoted when the launch count still matches
    traffic, traffic_src, hbm_gbps = None, None, None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")
    if os.path.exists(tpath) and args.batch == 256 and args.precision == "bf16" and not args.chunk:
        tj = json.load(open(tpath))
        if abs(tj.get("launches_per_pass", -1) - launches) < 0.5:
            traffic, traffic_src = round(tj["bytes_per_launch"]), tj.get("source")
            hbm_gbps = round(tj["bytes_per_pass"] / (conv_ms * 1e-3) / 1e9, 1)
    dt_code = L.BF16 if args.precision == "bf16" else L.F32
    roof = {"bound": "mfma", "kernel": lib.ivosw_assess_dominant_kernel(dt_code).decode(), "achieved": round(achieved, 2),
            "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
            "hbm_GBps_of_family": hbm_gbps, "hbm_peak_GBps": 8000.0,
            "launches_per_step": launches, "avg_launch_us": round(conv_ms * 1e3 / max(launches, 1), 2),
            "flops_per_launch": flops_step / max(launches, 1), "kernel_ms_per_step": round(conv_ms, 3)}
    return fps, dt, roof


def build_dqn(args, rank, dev):
    from ivos_w_amd.models.agent import Agent
    from ivos_w_amd.models.momory_pool import DeviceReplay
    agent = Agent(dev, agent_cfg())
    for net, seed in ((agent.policy_net, 0), (agent.target_net, 0)):
        net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.brain_state_dict(seed).items()})
    replay = DeviceReplay(synth.replay_transitions(n=args.replay, T=25, seed=2019), dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(7 + 1000003 * rank)        # rank-offset minibatch stream
    np.random.seed(0)                          # target-sync coin: identical on every rank
    return agent, replay, gen


def bench_dqn(args, rank, world, dev, dist, steps, warmup):
    agent, replay, gen = build_dqn(args, rank, dev)
    B = args.minibatch

    def step():
        idx = torch.randint(0, len(replay), (B,), device=dev, generator=gen)
        agent.loss_and_grads(replay.sample(idx))
        if world > 1:
            dist.all_reduce(agent.policy_net.flat_grad)
            agent.optimizer.grad_scale = 1.0 / world
        agent.optimizer.step()
        if np.random.random() < agent.update_rate:
            agent.sync_target()
    dt = timed(step, steps, warmup, dev, dist)
    assert torch.isfinite(agent.policy_net.flat).all()
    return world * steps / dt, dt


def bench_jf(rank, world, dev, dist, steps=20, warmup=3, N=100, O=3):
    """DAVIS J and F of one 100-frame 480p sequence with 3 objects per step, label maps resident in
    HBM as uint8, counts left on the device (the host ratio step is a few hundred float64 divisions).  HBM-bound:
    algorithmic bytes = 2 label maps read once."""
    from ivos_w_amd import metrics
    gt, pr = synth.label_maps(N, 480, 854, O, seed=4 + rank)
    tg, tp = torch.from_numpy(gt).to(dev), torch.from_numpy(pr).to(dev)
    lib = L.lib()
    counts = torch.empty((N, O, 6), dtype=torch.int64, device=dev)
    nbytes = lib.ivosw_jf_ws_bytes(N, 480, 854, O)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    ids = bytes(range(1, O + 1))
    st = L.stream_ptr(dev)

    def step():
        L.check(lib.ivosw_jf_counts(L.dptr(tg), L.dptr(tp), N, 480, 854, ids, O, 8, L.dptr(counts), L.dptr(ws), nbytes, st), "jf")
    dt = timed(step, steps, warmup, dev, dist)
    fps = world * N * steps / dt
    out = {"metric": "jf_scored_frames_per_sec", "value": round(fps, 1), "unit": "frames/s (480p, 3 objects, J and F)",
           "us_per_sequence": round(dt / steps * 1e6, 1), "dtype": "u8",
           "roofline": {"bound": "hbm", "achieved": round(2 * N * 480 * 854 * steps / dt / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                        "frac": round(2 * N * 480 * 854 * steps / dt / 8e12, 4),
                        "note": "algorithmic bytes = both label maps read once (820 KB per frame); the boundary-match pass works on bit maps 8x smaller"}}
    if rank == 0 and world == 1:
        from oracle import jf_oracle as jo          # checker + CPU baseline only
        j, f = metrics.batched_j_and_f(tg[:4], tp[:4], nb_objects=O)
        assert np.array_equal(j, jo.batched_jaccard(gt[:4], pr[:4], nb_objects=O)) and np.array_equal(f, jo.batched_f_measure(gt[:4], pr[:4], nb_objects=O))
        t0 = time.perf_counter()
        jo.sequence_metric("J_AND_F", gt[:8], pr[:8], O)
        cdt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(8 / cdt, 2), "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": "8 frames x 3 objects, numpy/scipy oracle of davisinteractive's batched_jaccard + batched_f_measure"}
    return out


def bench_seg(rank, world, dev, dist, steps=20, warmup=3, n=100, C=4):
    """upsample + argmax + softmax epilogue for one 100-frame sequence per step (stride-4 logits of
    3 objects + background -> 480p), probabilities written once, object-major.  HBM-bound on the outputs."""
    from ivos_w_amd.utils import utils_manet
    g = torch.Generator(device=dev).manual_seed(5 + rank)
    x = torch.randn(n, C, 120, 214, device=dev, generator=g) * 3
    store = utils_manet.ProbStore(n, C, 480, 854, dev)

    def step():
        utils_manet.seg_epilogue(x, 480, 854, store, 0)
    dt = timed(step, steps, warmup, dev, dist)
    out_bytes = n * 480 * 854 * (C * 4 + 8 + 1 + 4)       # probs + int64 / uint8 / float labels
    res = {"metric": "seg_epilogue_frames_per_sec", "value": round(world * n * steps / dt, 1), "unit": "frames/s (480p, 4 channels)",
           "us_per_sequence": round(dt / steps * 1e6, 1), "dtype": "f32",
           "roofline": {"bound": "hbm", "achieved": round(out_bytes * steps / dt / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                        "frac": round(out_bytes * steps / dt / 8e12, 4),
                        "note": "algorithmic bytes = outputs written once (29 B per pixel); the stride-4 logits are 1/16 of a channel plane"}}
    if rank == 0 and world == 1:
        from oracle import seg_oracle as so           # checker + CPU baseline only
        xc = x[:4].cpu()
        t0 = time.perf_counter()
        up, lab = so.epilogue(xc, 480, 854)
        pc = torch.softmax(up, 1)
        cdt = time.perf_counter() - t0
        assert (store.all_P[:4].cpu() - pc).abs().max().item() < 2e-6
        res["cpu_baseline"] = {"value": round(4 / cdt, 2), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": "4 frames, torch-CPU interpolate + argmax + softmax (the reference's own calls)"}
    return res


def cpu_baseline_assess():
    from oracle import assess_oracle as ao
    sd = ao.to_torch_sd(synth.assessnet_state_dict(0))
    tf, tp = synth.assess_inputs(8, seed=1234)
    # oneDNN convolutions at 8 x 256^2 stop scaling past ~16 threads (measured on the 256-core GPU host: 8 thr
    # 24 fps, 16 thr 30 fps, 64 thr 10 fps), so the baseline uses min(16, cores) threads and says so in `cores`
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ao.assess_forward(sd, tf, tp)
    reps, t0 = 0, time.perf_counter()
    while reps < 3 or (time.perf_counter() - t0 < 10.0 and reps < 40):
        ao.assess_forward(sd, tf, tp)
        reps += 1
    dt = time.perf_counter() - t0
    return {"value": round(8 * reps / dt, 2), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{reps} x AssessNet.forward on 8 synthetic 480p pairs (BASELINE configs[0]), torch-CPU oracle"}


def cpu_baseline_dqn():
    from oracle import brain_oracle as bo
    tr = synth.replay_transitions(n=2000, T=25, seed=2019)
    P = synth.brain_state_dict(0)
    Pt = {k: v.copy() for k, v in P.items()}
    M = {k: np.zeros_like(v) for k, v in P.items()}
    V = {k: np.zeros_like(v) for k, v in P.items()}
    cfg = dict(gamma=0.95, lr=5e-6, weight_decay=5e-4, update_rate=0.05)
    reps, t0 = 0, time.perf_counter()
    while reps < 3 or (time.perf_counter() - t0 < 8.0 and reps < 200):
        batch = synth.collate_np(tr, synth.minibatch_indices(reps % 8, n=2000, B=128, seed=7))
        bo.dqn_step(P, Pt, M, V, reps + 1, batch, cfg, 1.0)
        reps += 1
    dt = time.perf_counter() - t0
    return {"value": round(reps / dt, 2), "unit": "steps/s", "cores": 1, "kind": "port",
            "sample": f"{reps} x update_agent, minibatch 128, T=25, numpy oracle"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["assess", "dqn"], default="assess")
    ap.add_argument("--batch", type=int, default=256, help="frames per GPU per step (assessment)")
    ap.add_argument("--precision", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--minibatch", type=int, default=128)
    ap.add_argument("--replay", type=int, default=50000)
    ap.add_argument("--dqn-steps", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--layer-report", default="", help="write a per-conv-layer timing table (HIP events) to this file")
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the hot path")
    rank, world, dev, dist = dist_setup(args.gpus)
    L.lib()

    line = {"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "data": "synthetic"}
    if args.workload == "assess":
        fps, dt, roof = bench_assess(args, rank, world, dev, dist)
        dqn_sps, dqn_dt = bench_dqn(args, rank, world, dev, dist, args.dqn_steps, 20)
        line.update({"metric": "assessed_frames_per_sec", "value": round(fps, 1), "unit": "frames/s",
                     "ms_per_step": round(dt / args.steps * 1e3, 3), "dtype": args.precision,
                     "config": {"workload": f"AssessNet.forward, batch {args.batch} x 480x854 frame+mask per GPU (BASELINE configs[1])",
                                "batch_per_gpu": args.batch, "chunk": args.chunk or "default", "parallelism": f"frames sharded x{world}"},
                     "roofline": roof,
                     "dqn": {"metric": "dqn_agent_steps_per_sec", "value": round(dqn_sps, 1), "unit": "minibatch-steps/s (all ranks)",
                             "transitions_per_sec": round(dqn_sps * args.minibatch, 1), "minibatch_per_gpu": args.minibatch,
                             "replay": args.replay, "T": 25, "steps": args.dqn_steps, "us_per_step": round(dqn_dt / args.dqn_steps * 1e6, 1),
                             "dtype": "f32", "collective": "rccl all_reduce(724KB)" if world > 1 else None}})
    else:
        sps, dt = bench_dqn(args, rank, world, dev, dist, args.steps, args.warmup)
        line.update({"metric": "dqn_agent_steps_per_sec", "value": round(sps, 1), "unit": "minibatch-steps/s",
                     "ms_per_step": round(dt / args.steps * 1e3, 4), "dtype": "f32",
                     "config": {"workload": f"Double-DQN update, replay {args.replay}, minibatch {args.minibatch}/GPU, T=25 (BASELINE configs[2]/[3])",
                                "parallelism": f"dp{world}"},
                     "roofline": {"bound": "mfma", "achieved": round(10.5e9 * sps / world / 1e12, 3), "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                                  "frac": round(10.5e9 * sps / world / 1e12 / PEAK_F32_TFLOPS, 5), "traffic": None,
                                  "note": "whole-step algorithmic 10.5 GFLOP / step time: the step is latency-bound (SURVEY §8d)"}})
    if args.workload == "assess":
        line["jf"] = bench_jf(rank, world, dev, dist)
        line["seg_epilogue"] = bench_seg(rank, world, dev, dist)
        if args.no_cpu_baseline:
            line["jf"].pop("cpu_baseline", None)
            line["seg_epilogue"].pop("cpu_baseline", None)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline_assess() if args.workload == "assess" else cpu_baseline_dqn()
        if args.workload == "assess":
            line["dqn"]["cpu_baseline"] = cpu_baseline_dqn()
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
