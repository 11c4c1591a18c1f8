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


def test_rank_batch_sampler_deals_the_shuffled_minibatches_out():
    """parallel.RankBatchSampler: one shared permutation per epoch, cut into minibatches, rank r takes batches r, r + world, ...:
    the ranks' batches of a data-parallel step are disjoint, together they are `world` consecutive minibatches of the single-process
    order, every rank runs the same number of steps, and the deal is a pure function of the shared seed."""
    from ivos_w_amd.parallel import RankBatchSampler
    n, B, world = 103, 8, 2
    per_rank = [list(RankBatchSampler(n, B, r, world, seed=77)) for r in range(world)]
    assert len(per_rank[0]) == len(per_rank[1]) == len(RankBatchSampler(n, B, 0, world, 77)) == (-(-n // B)) // world
    g = torch.Generator()
    g.manual_seed(77)
    perm = torch.randperm(n, generator=g).tolist()
    single = [perm[i:i + B] for i in range(0, n, B)]
    for k in range(len(per_rank[0])):
        assert per_rank[0][k] == single[2 * k] and per_rank[1][k] == single[2 * k + 1]
        assert not set(per_rank[0][k]) & set(per_rank[1][k])
    assert list(RankBatchSampler(n, B, 1, world, seed=77)) == per_rank[1]
    assert list(RankBatchSampler(n, B, 1, world, seed=78)) != per_rank[1]
    # a DataLoader takes it as its batch sampler
    from torch.utils.data import DataLoader
    dl = DataLoader(list(range(n)), batch_sampler=RankBatchSampler(n, B, 0, world, 77))
    assert [b.tolist() for b in dl] == per_rank[0]
    second = [b.tolist() for b in dl]                        # a second pass over the same loader: a fresh permutation, again shared
    assert second != per_rank[0] and sorted(sum(second, [])) != [] and len(second) == len(per_rank[0])
    other = RankBatchSampler(n, B, 1, world, 77)
    list(other)
    assert not set(sum(second, [])) & set(sum(list(other), []))


def _forced_world1_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", IVOSW_FORCE_DIST="1")
    r, w, dev = parallel.init("gloo")
    assert (r, w) == (0, 1) and dist.is_initialized() and parallel.collective_active() and parallel.forced()
    g = torch.arange(1000, dtype=torch.float32) * 0.25 - 3.0
    want = g.clone()
    calls = []
    real = dist.all_reduce

    def counting(t, *a, **k):
        calls.append(t.data_ptr())
        return real(t, *a, **k)
    dist.all_reduce = counting
    scale = parallel.allreduce_grads(g)
    dist.all_reduce = real
    assert scale == 1.0 and torch.equal(g, want) and calls == [g.data_ptr()]        # the sum over one rank, through the backend, in place
    assert parallel.collective_path(g) == "backend"
    dist.destroy_process_group()
    os.environ["IVOSW_FORCE_DIST"] = "0"
    assert not parallel.collective_active() and parallel.collective_path(g) is None
    q.put("ok")


def test_forced_world_of_one_goes_through_the_backend_collective():
    """IVOSW_FORCE_DIST=1 (round 5): a single rank initialises the process group and its gradient arena takes the backend's all-reduce —
    the host logic behind tests/test_gpu_dist.py::test_rccl_world1... and `bench.py --gpus 1 --force-dist`, here on gloo / CPU."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_forced_world1_worker, args=(_free_port(), q))
    p.start()
    assert q.get(timeout=120) == "ok"
    p.join(60)
    assert p.exitcode == 0


@pytest.mark.parametrize("die", [False, True])
def test_bench_line_survives_a_process_that_dies_in_the_late_p2p_leg(die):
    """bench.py runs the peer-to-peer DQN leg as its LAST act under LineGuard: a watchdog child that holds the finished line and prints it
    iff the parent dies before disarming (abort() from inside the HIP runtime runs no Python handler).  Exactly one line either way."""
    import json
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent(f'''
        import importlib.util, json, os, sys
        sys.argv = ["bench.py"]
        spec = importlib.util.spec_from_file_location("bench", {os.path.join(root, "bench.py")!r})
        b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
        line = {{"metric": "assessed_frames_per_sec", "value": 1.0, "dqn": {{"collectives": {{"backend": {{"us_per_step": 200.0}}}}}}}}
        g = b.LineGuard(line, line["dqn"]["collectives"])
        assert "p2p" not in line["dqn"]["collectives"]
        if {die!r}:
            os.abort()
        line["dqn"]["collectives"]["p2p"] = {{"us_per_step": 150.0}}
        g.disarm()
        print(json.dumps(line), flush=True)
    ''')
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (r.stdout, r.stderr[-500:])
    d = json.loads(lines[0])
    assert d["value"] == 1.0 and d["dqn"]["collectives"]["backend"]["us_per_step"] == 200.0
    if die:
        assert r.returncode != 0 and "died inside the peer-to-peer leg" in d["dqn"]["collectives"]["p2p"]["error"]
    else:
        assert r.returncode == 0 and d["dqn"]["collectives"]["p2p"] == {"us_per_step": 150.0}
