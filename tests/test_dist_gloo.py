"""Multi-process (world_size 2, gloo, CPU) coverage of the N>1 host path: shard ranges + ragged all-gather for the
assessment shards, and the data-parallel DQN invariant — summing per-rank gradients of half batches and scaling by
1/world equals the single-process full-batch gradient, and clamp+Adam on it keeps replicas identical.  The oracle
stands in for the HIP kernels here (it is the checker; the thing under test is ivos_w_amd.parallel)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ivos_w_amd import parallel, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    from oracle import brain_oracle as bo
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, dev = parallel.init("gloo")
    assert (r, w) == (rank, world) and parallel.world() == world and parallel.rank() == rank
    # --- assessment shards: 7 frames over 2 ranks, ragged gather restores order
    n = 7
    lo, hi = parallel.shard_range(n, r, w)
    local = torch.arange(lo, hi, dtype=torch.float32) * 10
    full = parallel.gather_shards(local, n)
    assert torch.equal(full, torch.arange(n, dtype=torch.float32) * 10)
    # --- DQN: per-rank half batch -> all-reduce -> scale == full batch gradient
    tr = synth.replay_transitions(n=300, T=9, seed=5)
    P, Pt = synth.brain_state_dict(0), synth.brain_state_dict(1)
    idx = synth.minibatch_indices(0, n=300, B=16, seed=7)
    mine = idx[r * 8:(r + 1) * 8]
    _, G = bo.dqn_loss_and_grads(P, Pt, synth.collate_np(tr, mine), 0.95)
    flat = torch.from_numpy(synth.brain_flat(G).copy())
    scale = parallel.allreduce_grads(flat)
    assert scale == 0.5
    _, Gfull = bo.dqn_loss_and_grads(P, Pt, synth.collate_np(tr, idx), 0.95)
    want = synth.brain_flat(Gfull)
    got = flat.numpy() * scale
    np.testing.assert_allclose(got, want, rtol=2e-3, atol=2e-6 * np.abs(want).max())
    # --- replicas stay identical: same averaged grads + shared-seed coin on both ranks
    rng = np.random.RandomState(123)
    coin = parallel.shared_coin(rng)
    g0 = parallel.rank_generator(7, 0)
    g1 = parallel.rank_generator(7, 1)
    assert torch.randint(0, 1000, (4,), generator=g0).tolist() != torch.randint(0, 1000, (4,), generator=g1).tolist()
    q.put((rank, float(got.sum()), coin))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2]        # identical reduced gradient and coin on both ranks


def test_shard_range_properties():
    for n in (0, 1, 7, 256, 1000):
        for w in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
