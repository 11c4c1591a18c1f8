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


def forced():
    """IVOSW_FORCE_DIST=1: a world of ONE rank still initialises the process group and sends its gradient arena through the
    backend's all-reduce (the sum over one rank: bit-identical to the single-process step).  This is how the RCCL communicator,
    the collective on the compute stream and the clamp + Adam behind it are exercised on a one-GPU box (tests/test_gpu_dist.py,
    `bench.py --gpus 1 --force-dist`) instead of running for the first time on the 8-GPU node."""
    return os.environ.get("IVOSW_FORCE_DIST", "0") == "1"


def collective_active():
    """True when the data-parallel step runs its collective: more than one rank, or a forced world of one."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or forced())


def init(backend=None):
    """Initialise from the torchrun environment (RANK/WORLD_SIZE/LOCAL_RANK/MASTER_*). Returns (rank, world, device)."""
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("IVOSW_LOCAL_DEVICE", os.environ.get("LOCAL_RANK", 0)))      # tests put two ranks on one GPU
    use_gpu = torch.cuda.is_available()
    device = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(local)
    if (world > 1 or forced()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(so.getsockname()[1])
        backend = backend or os.environ.get("IVOSW_DIST_BACKEND") or ("nccl" if use_gpu else "gloo")
        if backend == "nccl" and use_gpu:
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
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


class P2PAllReduce:
    """One-shot all-reduce of a flat fp32 device vector over xGMI peer-to-peer writes (csrc/p2p.hip): every rank pushes its
    vector into its slot of every peer's fine-grained arena and sums the slots it received in rank order — two kernels per
    call, one hop of latency instead of the 2 (N-1) of a ring, the same bits on every rank.

    ``P2PAllReduce.create(n, device)`` is collective: every rank allocates its arena, the IPC handles go round with
    ``all_gather_object``, every rank maps the peers and the path is SELF-TESTED against ``dist.all_reduce`` on random data;
    unless every rank passes, all of them get ``None`` back (and use RCCL).

    OPT-IN (``IVOSW_P2P=1``): the cross-GPU write path has only ever run between two processes on one MI355X, so the product
    default is the backend's all-reduce (RCCL over xGMI) until an 8-GPU node has validated it; ``IVOSW_BENCH_P2P=1 bench.py --gpus N``
    times both.
    ``IVOSW_P2P_SELFTEST_FAIL=<rank>`` makes that rank report a failed self-test (fault injection: every rank must then end up
    on the backend path).  A peer that does not arrive within ``timeout_ms`` does NOT hang the GPU and is NOT silent either: the
    reduce kernels leave their outputs untouched and set the arena's error word, which ``check()`` reads (every call in
    ``Agent.update_agent``, which synchronises for its loss anyway; every ``every`` calls in the lean loops) and turns into a
    RuntimeError — after a timeout the parity double-buffering no longer holds, so the step cannot simply be retried."""

    def __init__(self):
        self.arena = None
        self.peers = []
        self.epoch = 0
        self.calls_since_check = 0
        self.collective = False         # create() reached its collective phase: teardown must be collective too

    @classmethod
    def create(cls, n, device, timeout_ms=2000):
        import ctypes as C
        from . import _lib as L
        w, r = world(), rank()
        device = torch.device(device)
        if w < 2 or device.type != "cuda" or os.environ.get("IVOSW_P2P", "0") != "1":
            return None
        self, ok = cls(), True
        self.collective = True
        self.n, self.rank, self.world, self.device, self.timeout_ms = int(n), r, w, device, int(timeout_ms)
        lib = L.lib()
        hb = lib.ivosw_p2p_handle_bytes()
        handle = (C.c_ubyte * hb)()
        try:
            with torch.cuda.device(device):
                ptr = C.c_void_p()
                ok = lib.ivosw_p2p_alloc(lib.ivosw_p2p_arena_bytes(w, n), C.byref(ptr), handle, hb) == 0
                if ok:
                    self.arena = ptr.value
        except Exception:
            ok = False
        handles = [None] * w
        dist.all_gather_object(handles, bytes(handle) if ok else None)      # collective on every rank, pass or fail
        ok = ok and all(h is not None for h in handles)
        table = (C.c_void_p * w)()
        if ok:
            try:
                with torch.cuda.device(device):
                    for s in range(w):
                        if s == r:
                            table[s] = self.arena
                            continue
                        p = C.c_void_p()
                        buf = (C.c_ubyte * hb).from_buffer_copy(handles[s])
                        if lib.ivosw_p2p_open(buf, C.byref(p)) != 0:
                            ok = False
                            break
                        self.peers.append(p.value)
                        table[s] = p.value
            except Exception:
                ok = False
        self.table = table
        ok = _all_ranks_agree(ok, device)
        if ok:
            # self-test: three rounds (both slot parities, re-use) against the reference collective
            g = torch.Generator(device="cpu").manual_seed(1234 + r)
            for _ in range(3):
                x = torch.randn(n, generator=g).to(device)
                want = x.clone()
                _reference_allreduce(want)                  # collective: every rank, every round, whatever happened before
                try:
                    got = x.clone()
                    self(got)
                    torch.cuda.synchronize(device)
                    ok = ok and self.error() == 0 and bool(torch.allclose(got, want, rtol=1e-5, atol=1e-5 * float(want.abs().max()) + 1e-30))
                except Exception:
                    ok = False
            if os.environ.get("IVOSW_P2P_SELFTEST_FAIL", "") == str(r):
                ok = False                                  # fault injection
            ok = _all_ranks_agree(ok, device)
        if not ok:
            self.close()
            return None
        return self

    def __call__(self, flat):
        """In-place sum over ranks of a contiguous fp32 CUDA vector of ``n`` elements, on the current stream."""
        from . import _lib as L
        assert flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous() and flat.numel() == self.n
        self.epoch += 1
        L.check(L.lib().ivosw_p2p_allreduce(L.dptr(flat), L.dptr(flat), self.n, self.rank, self.world, self.table, self.epoch,
                                            self.timeout_ms, L.stream_ptr(flat.device)), "p2p_allreduce")
        return flat

    def fused_step(self, brain, optimizer):
        """All-reduce of ``brain.flat_grad`` + clamp + Adam in TWO launches (push; wait + rank-ordered sum + clamp + Adam):
        ``ivosw_p2p_allreduce_clamp_adam``.  flat_grad is left holding the SUM over ranks; the optimizer's step counter advances."""
        from . import _lib as L
        optimizer._ensure()
        g = optimizer.param_groups[0]
        optimizer.state["step"] += 1
        self.epoch += 1
        flat, grad = brain.flat, brain.flat_grad
        L.check(L.lib().ivosw_p2p_allreduce_clamp_adam(
            L.dptr(grad), L.dptr(grad), self.n, self.rank, self.world, self.table, self.epoch, self.timeout_ms, L.dptr(flat),
            L.dptr(optimizer.state["exp_avg"]), L.dptr(optimizer.state["exp_avg_sq"]), optimizer.state["step"], g["lr"], g["betas"][0],
            g["betas"][1], g["eps"], g["weight_decay"], g["clamp"], L.stream_ptr(flat.device)), "p2p_allreduce_clamp_adam")

    def error(self):
        import ctypes as C
        from . import _lib as L
        e = C.c_int(0)
        L.check(L.lib().ivosw_p2p_error(C.c_void_p(self.arena), C.byref(e)), "p2p_error")
        return e.value

    def check(self, every=1):
        """Read the arena's error word every ``every`` calls (a 4-byte D2H copy, i.e. a stream synchronisation) and raise on a
        peer timeout.  The ranks that did arrive time out in turn on the missing peer and raise too."""
        self.calls_since_check += 1
        if self.calls_since_check < every:
            return
        self.calls_since_check = 0
        if self.error() != 0:
            raise RuntimeError(f"ivos-w P2P all-reduce: rank {self.rank} waited more than {self.timeout_ms} ms for a peer's gradient; "
                               "the replicas are no longer synchronised (nothing was applied for that step). Restart from the last "
                               "checkpoint, with IVOSW_P2P=0 to use the backend's all-reduce.")

    def close(self):
        """Collective whenever create() reached its collective phase, in the order synchronise -> barrier -> unmap peers -> free:
        no rank unmaps or frees while one of its own pushes, or a peer's, may still be writing."""
        from . import _lib as L
        lib = L.lib()
        if self.collective and dist.is_initialized():
            try:
                torch.cuda.synchronize(self.device)
                dist.barrier()
            except Exception:
                pass
        self.collective = False
        for p in self.peers:
            lib.ivosw_p2p_close(p)
        self.peers = []
        if self.arena:
            lib.ivosw_p2p_free(self.arena)
            self.arena = None


def _reference_allreduce(t):
    if t.is_cuda and dist.get_backend() == "gloo":
        h = t.to("cpu")
        dist.all_reduce(h)
        t.copy_(h)
    else:
        dist.all_reduce(t)


def _all_ranks_agree(ok, device):
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
    if dist.get_backend() != "gloo":
        flag = flag.to(device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(int(flag.item()) == 1)


_P2P = {}     # (n, device) -> P2PAllReduce or None (decided once per process, collectively)


def p2p_for(flat_grad):
    key = (flat_grad.numel(), str(flat_grad.device))
    if key not in _P2P:
        _P2P[key] = P2PAllReduce.create(flat_grad.numel(), flat_grad.device)
    return _P2P[key]


def _backend_allreduce(flat_grad):
    if flat_grad.is_cuda and dist.get_backend() == "gloo":
        # hosts without RCCL (and the two-ranks-on-one-GPU tests): stage the 724 KB arena through host memory
        h = flat_grad.to("cpu")
        dist.all_reduce(h)
        flat_grad.copy_(h)
    else:
        dist.all_reduce(flat_grad)            # RCCL over xGMI


def allreduce_grads(flat_grad, check_every=1):
    """Sum the flat gradient arena over ranks in place; returns the scale (1/world) the optimizer must apply.
    Path: the one-shot xGMI peer-to-peer all-reduce when it is enabled and every rank's self-test passed, else the backend's."""
    w = world()
    if w > 1:
        p2p = p2p_for(flat_grad) if flat_grad.is_cuda else None
        if p2p is not None:
            p2p(flat_grad)
            p2p.check(check_every)
        else:
            _backend_allreduce(flat_grad)
    elif collective_active():
        _backend_allreduce(flat_grad)         # forced world of one: the backend's all-reduce over a single rank
    return 1.0 / w


def collective_path(flat_grad):
    """'p2p' | 'backend' | None (single process): which all-reduce the data-parallel step of this process uses."""
    if world() < 2:
        return "backend" if collective_active() else None
    return "p2p" if (flat_grad.is_cuda and p2p_for(flat_grad) is not None) else "backend"


def data_parallel_step(brain, optimizer, check_every=1):
    """The exchange step of synchronous data parallelism + the optimizer step, on every rank: gradients summed over ranks, scaled by
    1/world INSIDE the clamp + Adam kernel (the clamp of models/agent.py:157-159 sees the averaged gradient, as one large batch
    would).  P2P path: two launches (push | wait + sum + clamp + Adam); backend path: all-reduce, then the fused clamp + Adam."""
    w = world()
    if w < 2:
        optimizer.grad_scale = allreduce_grads(brain.flat_grad, check_every) if collective_active() else 1.0
        optimizer.step()
        return
    grad = brain.flat_grad
    p2p = p2p_for(grad) if grad.is_cuda else None
    if p2p is not None and os.environ.get("IVOSW_P2P_FUSED", "1") != "0":
        p2p.fused_step(brain, optimizer)
        p2p.check(check_every)
    else:
        optimizer.grad_scale = allreduce_grads(grad, check_every)
        optimizer.step()


class RankBatchSampler:
    """The DataLoader's shuffled minibatches, dealt out to the ranks: ONE permutation of the dataset per epoch from a shared seed
    (identical on every rank), cut into batches of ``batch_size``; rank r takes batches r, r + world, ...  — so a data-parallel
    step consumes ``world`` consecutive minibatches of the single-process order, and every rank runs the same number of steps
    (a trailing group of fewer than ``world`` batches is dropped)."""

    def __init__(self, n, batch_size, rank_, world_, seed):
        self.n, self.batch_size, self.rank, self.world, self.seed = int(n), int(batch_size), int(rank_), int(world_), int(seed)
        self.epoch = 0

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.seed + 7919 * self.epoch)         # a fresh permutation per pass over the loader, the same on every rank
        self.epoch += 1
        perm = torch.randperm(self.n, generator=g).tolist()
        batches = [perm[i:i + self.batch_size] for i in range(0, self.n, self.batch_size)]
        for k in range(len(self)):
            yield batches[k * self.world + self.rank]

    def __len__(self):
        return -(-self.n // self.batch_size) // self.world


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
