"""Single-node data parallelism: one process per GPU, torch.distributed (backend 'nccl' = RCCL over xGMI; 'gloo'
on CPU for tests).  Two shapes of work:

  * assessment — independent (frame, object) units: contiguous shards per rank, NO data-path collective; the
    per-frame scores (a few hundred floats) are all-gathered at the end.
  * DQN — synchronous data parallel: every rank samples its own minibatch (rank-offset RNG stream), gradients are
    summed with ONE all-reduce of the flat 180 993-float arena (724 KB), scaled by 1/world inside the fused
    clamp+Adam kernel (so the clamp sees the averaged gradient), and the target-sync coin comes from a
    shared-seed host RNG so replicas stay bit-identical.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def init(backend=None):
    """Initialise from the torchrun environment (RANK/WORLD_SIZE/LOCAL_RANK/MASTER_*). Returns (rank, world, device)."""
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    use_gpu = torch.cuda.is_available()
    device = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend or ("nccl" if use_gpu else "gloo"), rank=rank, world_size=world)
    return rank, world, device


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard_range(n, rank_, world_):
    """Contiguous, balanced [lo, hi) of n units for rank_ (first n % world ranks get one extra)."""
    q, r = divmod(n, world_)
    lo = rank_ * q + min(rank_, r)
    return lo, lo + q + (1 if rank_ < r else 0)


def gather_shards(local, n, device=None):
    """All-gather ragged 1-D float shards produced by ``shard_range`` back into the full [n] vector on every rank."""
    w = world()
    if w == 1:
        return local
    sizes = [shard_range(n, r, w)[1] - shard_range(n, r, w)[0] for r in range(w)]
    pad = max(sizes)
    buf = torch.zeros(pad, dtype=local.dtype, device=local.device)
    buf[:local.numel()] = local
    out = [torch.empty_like(buf) for _ in range(w)]
    dist.all_gather(out, buf)
    return torch.cat([o[:s] for o, s in zip(out, sizes)])


def allreduce_grads(flat_grad):
    """Sum the flat gradient arena over ranks in place; returns the scale (1/world) the optimizer must apply."""
    w = world()
    if w > 1:
        if flat_grad.is_cuda and dist.get_backend() == "gloo":
            # hosts without RCCL (and the two-ranks-on-one-GPU test): stage the 724 KB arena through pinned host memory
            h = flat_grad.to("cpu")
            dist.all_reduce(h)
            flat_grad.copy_(h)
        else:
            dist.all_reduce(flat_grad)        # RCCL over xGMI
    return 1.0 / w


def shared_coin(rng):
    """Target-sync coin flip: ``rng`` is a numpy RandomState seeded identically on every rank."""
    return rng.random_sample()


def rank_generator(seed, rank_, device="cpu"):
    """Rank-offset minibatch index stream."""
    g = torch.Generator(device=device)
    g.manual_seed(int(seed) + 1000003 * int(rank_))
    return g


def assess_sharded(assess_net, tf, tp):
    """Score the rank's contiguous shard of frames and return all scores on every rank ([n] float32)."""
    n = tf.shape[0]
    lo, hi = shard_range(n, rank(), world())
    local = assess_net(tf[lo:hi].contiguous(), tp[lo:hi].contiguous()).reshape(-1) if hi > lo else \
        torch.empty(0, dtype=torch.float32, device=tf.device)
    return gather_shards(local, n)
