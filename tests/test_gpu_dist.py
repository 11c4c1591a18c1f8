"""World-size-2 run of the PRODUCT path of the data-parallel DQN step (BASELINE configs[3]) on one MI355X: two processes,
both on cuda:0, gloo rendezvous with the gradient arena staged through host memory (two ranks cannot share one device
under RCCL).  Each rank runs Agent.update_agent on its half of a 128-transition minibatch for three steps.  Checks:
replicas stay bit-identical (same reduced gradient, same coin), the averaged gradient equals the single-process
full-batch gradient within fp32 summation tolerance, and so do the parameters after the three steps."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from ivos_w_amd import synth

pytestmark = pytest.mark.gpu
STEPS, B, LR = 3, 128, 5e-6


class AD(dict):
    __getattr__ = dict.__getitem__


def _cfg():
    return AD(phase="train", data=AD(subset="train"),
              agent=AD(memory_size=1000, gamma=0.95, eps_start=0.7, eps_end=0.25, eps_decay=500, update_rate=0.5, lr=LR,
                       weight_decay=5e-4))


def _agent(dev):
    from ivos_w_amd.models.agent import Agent
    a = Agent(dev, _cfg())
    for net, seed in ((a.policy_net, 0), (a.target_net, 1)):
        net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.brain_state_dict(seed).items()})
    return a


def _batch(tr, idx):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.collate_np(tr, idx).items()}


def _worker(rank, world, port, q, mode="backend"):
    import io
    import contextlib
    from ivos_w_amd import parallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      IVOSW_P2P="0" if mode == "backend" else "1")
    if mode == "p2p_fault":
        os.environ["IVOSW_P2P_SELFTEST_FAIL"] = "1"          # rank 1 reports a failed self-test: BOTH ranks must end up on the backend path
    r, w, dev = parallel.init("gloo")
    assert dev.type == "cuda" and w == 2
    tr = synth.replay_transitions(n=2000, T=25, seed=2019)
    agent = _agent(dev)
    np.random.seed(5)                                   # the shared target-sync coin
    out = []
    with contextlib.redirect_stdout(io.StringIO()):
        for s in range(STEPS):
            idx = synth.minibatch_indices(s, n=2000, B=B, seed=7)
            agent.update_agent(_batch(tr, idx[rank * (B // 2):(rank + 1) * (B // 2)]))
            out.append((agent.policy_net.flat_grad.cpu().numpy().copy(), agent.policy_net.flat.cpu().numpy().copy(),
                        agent.target_net.flat.cpu().numpy().copy()))
    path = parallel.collective_path(agent.policy_net.flat_grad)
    assert path == ("p2p" if mode == "p2p" else "backend"), (mode, path)
    if path == "backend":
        assert agent.optimizer.grad_scale == 0.5
    q.put((rank, out))
    for v in parallel._P2P.values():
        if v is not None:
            v.close()
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("mode", ["backend", "p2p", "p2p_fault"])
def test_world2_product_path_matches_single_process_full_batch(mode):
    """mode: 'backend' = the product default (torch.distributed all-reduce, then clamp + Adam with 1/world); 'p2p' = IVOSW_P2P=1: the
    one-shot peer-to-peer all-reduce FUSED with clamp + Adam (push | wait + rank-ordered sum + clamp + Adam: two launches);
    'p2p_fault' = P2P requested but rank 1's self-test is made to fail (IVOSW_P2P_SELFTEST_FAIL): every rank must then use the
    backend collective."""
    import io
    import contextlib
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    # replicas bit-identical after every step: reduced gradient arena, parameters, target net (coin agreed)
    for a, b in zip(res[0], res[1]):
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
    # single process, full batch
    dev = torch.device("cuda:0")
    tr = synth.replay_transitions(n=2000, T=25, seed=2019)
    agent = _agent(dev)
    start = agent.policy_net.flat.cpu().numpy().copy()
    np.random.seed(5)
    with contextlib.redirect_stdout(io.StringIO()):
        for s in range(STEPS):
            agent.update_agent(_batch(tr, synth.minibatch_indices(s, n=2000, B=B, seed=7)))
            if s == 0:
                want = agent.policy_net.flat_grad.cpu().numpy()
                got = res[0][0][0] * 0.5                # sum over ranks, scaled inside the Adam kernel
                np.testing.assert_allclose(got, want, rtol=2e-3, atol=2e-6 * np.abs(want).max())
    d1 = agent.policy_net.flat.cpu().numpy().astype(np.float64) - start
    d2 = res[0][-1][1].astype(np.float64) - start
    assert np.abs(d1).max() > 0.5 * LR                  # the steps moved the parameters
    close = np.abs(d2 - d1) <= 1e-2 * np.abs(d1) + 0.05 * LR * STEPS
    assert close.mean() > 0.999, close.mean()
    # the target-sync decisions (shared coin) are those of the single process
    assert np.array_equal(res[0][-1][2], res[0][-1][1]) == bool(torch.equal(agent.target_net.flat, agent.policy_net.flat))


@pytest.mark.parametrize("variant", ["default", "plain", "p2p_leg", "graph_no_p2p", "strong"])
def test_bench_multi_rank_branch_under_torchrun(tmp_path, variant):
    """bench.py's N > 1 branch exactly as the driver launches it (torch.distributed.run, one process per rank, barrier + max over
    ranks, rank 0 prints the line) — on this one-GPU box with both ranks on cuda:0 and the gloo rendezvous (two ranks cannot
    share a device under RCCL).  default: plain launches, the backend's all-reduce (dqn.value), then - as the LAST act of bench.py, the line
    complete and held by a watchdog child - the one-shot peer-to-peer all-reduce (IPC-mapped arenas of the two processes); p2p_leg: IVOSW_BENCH_P2P=0 opts out of it; graph_no_p2p: captured gradient graph + the collective of the backend (IVOSW_P2P=0: gloo staged through host memory).  Frames are sharded (weak scaling), the DQN leg all-reduces
    its gradient arena every step.  plain: `python bench.py --gpus 2 ...` with NO launcher in front — bench.py re-executes itself under
    torch.distributed.run (the driver's scaling run may be started that way) and still prints exactly one line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    env = dict(os.environ, IVOSW_BENCH_SAME_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    extra = []
    if variant == "default":                                 # (since round 6 the P2P leg is attempted by default; "default" checks the record's four fields)
        pass
    if variant == "p2p_leg":
        env["IVOSW_BENCH_P2P"] = "0"                         # the opt-out: the backend leg only
    if variant == "graph_no_p2p":
        env["IVOSW_P2P"] = "0"
        extra = ["--dqn-dp", "graph"]
    if variant == "strong":                                  # 16 frames IN TOTAL, 8 per rank ("256 total (strong)")
        env["IVOSW_P2P"] = "0"
        extra = ["--scaling", "strong"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--min-warm-s", "0",
           "--batch", "16", "--dqn-steps", "30", "--backend", "gloo", "--no-fp32"] + extra
    if variant == "plain":
        cmd = [sys.executable] + cmd[cmd.index(os.path.join(root, "bench.py")):]
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == ("strong" if variant == "strong" else "weak") and d["steps"] == 3 and d["value"] > 0 and d["checked"] is True
    assert d["config"]["parallelism"] == "frames sharded x2" and "cpu_baseline" not in d
    assert (d["config"]["total_batch"], d["config"]["batch_per_gpu"]) == ((16, 8) if variant == "strong" else (32, 16))
    assert abs(d["value"] - d["config"]["total_batch"] * 3 / (d["ms_per_step"] * 3e-3)) < 0.02 * d["value"]       # value = frames of ALL ranks / time
    assert d["dqn"]["value"] > 0 and "all-reduce" in d["dqn"]["collective"]
    if variant == "strong":
        return
    if variant in ("default", "plain", "p2p_leg"):
        legs = d["dqn"]["collectives"]                       # the timed collective paths; dqn.value is the backend leg (the product default)
        assert set(legs) == ({"backend"} if variant == "p2p_leg" else {"backend", "p2p"}), legs
        # ONE default invocation answers everything: the process group's own world size and backend string, the timed
        # backend leg, the P2P leg as a guarded late phase (timed, or skipped / failed WITH a reason - the line stands either way), and the
        # scaling against the committed single-GPU rate
        assert legs["backend"]["us_per_step"] > 0 and legs["backend"]["world_size"] == 2 and legs["backend"]["backend"] == "gloo", legs
        assert abs(d["dqn"]["us_per_step"] - legs["backend"]["us_per_step"]) < 0.2 and "_p2p_leg" not in d["dqn"]
        if "p2p" in legs:                                    # run as bench.py's last act, under the LineGuard watchdog
            assert legs["p2p"].get("us_per_step", 0) > 0 or "error" in legs["p2p"] or "skipped" in legs["p2p"], legs
            if "us_per_step" in legs["p2p"]:
                assert abs(legs["p2p"]["vs_backend"] - legs["backend"]["us_per_step"] / legs["p2p"]["us_per_step"]) < 0.01 and legs["p2p"]["scaling_vs_1gpu"] > 0
        sc = d["dqn"]["scaling_vs_1gpu"]
        assert sc["n_gpus"] == 2 and sc["value"] > 0 and sc["reference"]["steps_per_sec"] > 0
        assert abs(sc["value"] - d["dqn"]["value"] / sc["reference"]["steps_per_sec"]) < 0.01 * sc["value"] + 1e-3
        assert d["dqn"]["graph"] is False
    else:
        assert "gloo" in d["dqn"]["collective"] and d["dqn"]["graph"] is True


def _p2p_worker(rank, world, port, q):
    from ivos_w_amd import parallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", IVOSW_P2P="1")
    r, w, dev = parallel.init("gloo")
    n = 180993
    p2p = parallel.P2PAllReduce.create(n, dev)
    if p2p is None:
        q.put((rank, None))
    else:
        g = torch.Generator(device="cpu").manual_seed(77 + rank)
        outs = []
        for it in range(12):                                   # both slot parities, many re-uses, no host sync in between
            x = torch.randn(n, generator=g).to(dev) * (1 + it)
            p2p(x)
            outs.append(x)
        torch.cuda.synchronize(dev)
        assert p2p.error() == 0
        q.put((rank, [o.cpu().numpy() for o in outs]))
        p2p.close()
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_p2p_allreduce_two_ranks_on_one_device():
    """The one-shot peer-to-peer all-reduce (csrc/p2p.hip) with two processes on one MI355X: IPC-mapped fine-grained arenas,
    push + flag + reduce kernels, 12 back-to-back calls without host synchronisation.  Result = the rank-ordered fp32 sum of the
    two inputs, bit-identical on both ranks.  (Across GPUs the same kernels write over xGMI; that part cannot run here.)"""
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_p2p_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert res[0] is not None and res[1] is not None, "P2PAllReduce.create fell back (IPC mapping or self-test failed)"
    gens = [torch.Generator(device="cpu").manual_seed(77 + r) for r in range(2)]
    for it in range(12):
        xs = [torch.randn(180993, generator=g).numpy() * np.float32(1 + it) for g in gens]
        want = xs[0] + xs[1]                                   # rank order
        np.testing.assert_array_equal(res[0][it], want)
        np.testing.assert_array_equal(res[1][it], want)


def _timeout_worker(rank, world, port, q):
    from ivos_w_amd import parallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", IVOSW_P2P="1")
    r, w, dev = parallel.init("gloo")
    n = 180993
    p2p = parallel.P2PAllReduce.create(n, dev, timeout_ms=200)
    outcome = "no p2p"
    if p2p is not None:
        x = torch.ones(n, device=dev) * (rank + 1)
        keep = x.clone()
        if rank == 0:                                          # rank 1 never shows up for this step
            p2p(x)
            try:
                p2p.check()
                outcome = "silent"
            except RuntimeError as e:
                outcome = "raised: " + str(e)[:60]
            assert torch.equal(x, keep)                        # nothing was applied: the local gradient is NOT passed off as the sum
        else:
            outcome = "absent"
    q.put((rank, outcome))
    torch.distributed.barrier()
    if p2p is not None:
        p2p.close()
    torch.distributed.destroy_process_group()


def test_p2p_timeout_is_loud():
    """A peer that does not arrive within the timeout: the reduce kernel leaves its output untouched and sets the arena's error
    word; P2PAllReduce.check() turns it into a RuntimeError (round 2's path returned silently and kept the un-reduced gradient)."""
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_timeout_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert res[0].startswith("raised: ivos-w P2P all-reduce"), res
    assert res[1] == "absent"


def test_train_agent_data_parallel_under_torchrun(tmp_path):
    """train_agent.py (the entry point, not the bench) as a 2-rank data-parallel job exactly as torch.distributed.run launches it —
    both ranks on this box's one GPU (IVOSW_LOCAL_DEVICE=0, gloo rendezvous).  Every rank plays the same episodes; the update
    loop deals the shuffled minibatches out to the ranks and all-reduces the gradients.  Rank 0 writes the checkpoint and the
    summary, which records that the replicas' parameter bits were compared and are identical."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    work = str(tmp_path)
    env = dict(os.environ, IVOSW_LOCAL_DEVICE="0", IVOSW_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    args = ["with", "synthetic=1", f"ckpt_dir={work}/ckpt", f"agent.save_result_dir={work}/results", "num_epochs=1", "synth.n_sequences=2",
            "synth.n_frames=26", "synth.height=120", "synth.width=216", "agent.train_batch_size=16", "agent.update_rate=0.3"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "train_agent.py")] + args
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    hist = json.load(open(os.path.join(work, "results", "train_summary.json")))
    assert hist and hist[-1]["world"] == 2 and hist[-1]["replicas_identical"] is True and hist[-1]["collective"] == "backend"
    assert hist[-1]["updates"] > 0
    assert os.path.exists(os.path.join(work, "ckpt", "agent.pt"))
    assert os.path.isdir(os.path.join(work, "results", "rank1")) and not os.path.exists(os.path.join(work, "results", "rank1", "train_summary.json"))


def _rccl_world1_worker(port, q, forced):
    """forced: ONE rank, backend nccl (= RCCL): communicator creation, dist.all_reduce of the 724 KB gradient arena on the compute stream,
    the fused clamp + Adam behind it — parallel.data_parallel_step exactly as a rank of the 8-GPU job runs it.  not forced: the plain
    single-process steps, in a fresh process as well (the pytest process may carry tunables of earlier tests)."""
    import io
    import contextlib
    from ivos_w_amd import parallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", IVOSW_FORCE_DIST="1" if forced else "0",
                      IVOSW_P2P="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r, w, dev = parallel.init("nccl")
    assert dev.type == "cuda" and w == 1
    tr = synth.replay_transitions(n=2000, T=25, seed=2019)
    agent = _agent(dev)
    calls = []
    if forced:
        assert torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl" and parallel.collective_active()
        assert parallel.collective_path(agent.policy_net.flat_grad) == "backend"
        real = torch.distributed.all_reduce

        def counting(t, *a, **k):
            calls.append((tuple(t.shape), t.device.type, t.data_ptr()))
            return real(t, *a, **k)
        torch.distributed.all_reduce = counting
    else:
        assert not torch.distributed.is_initialized() and not parallel.collective_active()
    np.random.seed(5)
    out = []
    with contextlib.redirect_stdout(io.StringIO()):
        for s in range(STEPS):
            agent.update_agent(_batch(tr, synth.minibatch_indices(s, n=2000, B=B, seed=7)))
            out.append((agent.policy_net.flat_grad.cpu().numpy().copy(), agent.policy_net.flat.cpu().numpy().copy(),
                        agent.target_net.flat.cpu().numpy().copy()))
    if forced:
        torch.distributed.all_reduce = real
        # every step sent the flat gradient arena itself (no staging copy) through the backend, on the device
        assert len(calls) == STEPS and all(c == ((agent.policy_net.flat_grad.numel(),), "cuda", agent.policy_net.flat_grad.data_ptr()) for c in calls), calls
    assert agent.optimizer.grad_scale == 1.0
    q.put(out)
    if forced:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def test_rccl_world1_three_steps_are_bit_identical_to_the_single_process_step():
    """the RCCL path had never executed, not even at world size 1.  One rank initialises `nccl`, runs three
    `update_agent` steps through parallel.data_parallel_step (all-reduce over the one rank, then clamp + Adam with scale 1) and must
    reproduce the plain single-process steps BIT FOR BIT: gradients, parameters, target syncs."""
    ctx = mp.get_context("spawn")
    got = {}
    for forced in (True, False):
        s_ = socket.socket()
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
        s_.close()
        q = ctx.Queue()
        p = ctx.Process(target=_rccl_world1_worker, args=(port, q, forced))
        p.start()
        got[forced] = q.get(timeout=600)
        p.join(120)
        assert p.exitcode == 0
    moved = 0.0
    for a, b in zip(got[True], got[False]):
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
        moved = max(moved, float(np.abs(a[0]).max()))
    assert moved > 0                                     # real gradients went through the collective


def test_bench_force_dist_runs_the_multi_rank_branch_on_rccl():
    """`bench.py --gpus 1 --force-dist`: the N > 1 branch (process group, barrier + max over ranks around the timed regions, the
    all-reduce inside every DQN step) on the nccl backend with one rank — what the driver's 2 / 4 / 8-GPU runs execute per rank."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("IVOSW_FORCE_DIST", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "3", "--warmup", "1", "--batch", "32",
                        "--min-warm-s", "0", "--dqn-steps", "60", "--no-fp32", "--no-cpu-baseline", "--no-clock-probe", "--no-live-traffic"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and "force_dist" in line and "nccl" in line["force_dist"]
    assert line["value"] > 0 and line["dqn"]["value"] > 0
    assert line["dqn"]["collective"].startswith("RCCL all-reduce")
    assert line["dqn"]["collectives"]["backend"]["us_per_step"] > 0
