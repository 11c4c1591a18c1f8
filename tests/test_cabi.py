"""The C-ABI shared library loads and exports every symbol include/ivosw.h declares (no compute, no GPU)."""
import ctypes
import os
import re

import pytest

from ivos_w_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="ivosw.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ivosw_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def handle():
    if not L.available():
        import __graft_entry__ as g
        g.build()
    return ctypes.CDLL(L.LIB_PATH)


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(L.SIGNATURES)
    assert declared_symbols("ivosw_probe.h") == sorted(L.PROBE_SIGNATURES)
    assert not set(L.SIGNATURES) & set(L.PROBE_SIGNATURES)


def test_probes_are_not_part_of_the_product_library(handle):
    """a reference maintainer who binds include/ivosw.h sees hot-path entries and profiling aids only.  The
    tuning probes (single-kernel launches with phase stamps, the round-5 micro-benchmark contraction) are declared in
    include/ivosw_probe.h and exported by libivosw_probe.so alone - a superset build of the same sources (-DIVOSW_PROBES)."""
    assert not [s for s in L.PROBE_SIGNATURES if hasattr(handle, s)]
    assert not [s for s in declared_symbols() if s.endswith("_probe") and s != "ivosw_clock_probe"]
    probe = ctypes.CDLL(L.PROBE_LIB_PATH)
    missing = [s for s in list(L.SIGNATURES) + list(L.PROBE_SIGNATURES) if not hasattr(probe, s)]
    assert not missing, missing


def test_every_declared_symbol_is_exported(handle):
    missing = [s for s in declared_symbols() if not hasattr(handle, s)]
    assert not missing, missing


def test_host_only_queries(handle):
    lib = L.lib()
    assert lib.ivosw_version() >= 100
    assert lib.ivosw_brain_ws_bytes(128, 25) > 0 and lib.ivosw_brain_ws_bytes(0, 25) == 0
    assert lib.ivosw_dqn_ws_bytes(128, 25) > lib.ivosw_brain_ws_bytes(128, 25)
    # 23.5 M parameters: fp32 arena ~ 4 B each (+ folded downsample copies); the bf16 arena holds the K-major weights AND
    # their MFMA-fragment-ordered copies for the fused / wide kernels, so it is 2-3 x 47 MB
    nb16, nb32 = lib.ivosw_assess_packed_bytes(L.BF16), lib.ivosw_assess_packed_bytes(L.F32)
    assert 94e6 < nb32 < 130e6 and 47e6 < nb16 < 150e6
    assert lib.ivosw_assess_packed_bytes(7) == 0
    assert lib.ivosw_assess_packed_bytes(L.F32X3) == nb32      # the three-pass mode keeps the fp32 layout (weights pre-split in place)
    assert lib.ivosw_assess_ws_bytes(L.F32X3, 64, 480, 854, 0) == lib.ivosw_assess_ws_bytes(L.F32, 64, 480, 854, 0)
    assert lib.ivosw_assess_ws_bytes(L.BF16, 256, 480, 854, 0) > 0
    fam = lib.ivosw_assess_dominant_kernel(L.BF16).decode().split("|")
    assert "conv_igemm*" in fam and "bneck*" in fam           # kernel-name patterns of the tower's contraction kernels
    assert lib.ivosw_assess_dominant_kernel(L.F32).decode() == "conv_igemm*"
    assert lib.ivosw_tune_set(b"FUSE", 1) == 0 and lib.ivosw_tune_set(None, 1) != 0
    # the table holds every key the sources read, all set at once (round 3's held 32 of 47: the 33rd set failed and the tests
    # did not look), and L.tune_set — what tests and bench.py call — raises instead of returning a status nobody reads
    # (in a child process: the switches set here must not leak into other tests of this process)
    import subprocess
    import sys
    code = r"""
import os, re, sys
sys.path.insert(0, sys.argv[1])
from ivos_w_amd import _lib as L
lib = L.lib()
csrc = os.path.join(os.path.dirname(L.LIB_PATH), "csrc")
keys = sorted({k for f in os.listdir(csrc) if f.endswith((".hip", ".h", ".cpp"))
               for k in re.findall(r'tune_get\("([A-Z0-9_]+)"', open(os.path.join(csrc, f)).read())})
assert len(keys) >= 40, len(keys)
for i, k in enumerate(keys + ["CABI_TEST_%d" % j for j in range(40)]):
    assert lib.ivosw_tune_set(k.encode(), 7) == 0, (i, k)
try:
    L.tune_set("K" * 40, 1)
except RuntimeError as e:
    assert "bad key" in str(e)
else:
    raise SystemExit("L.tune_set did not raise")
print("tunables ok", len(keys))
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code, root], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "tunables ok" in r.stdout, r.stdout + r.stderr
    # the minibatch draw's host mirrors (C and Python) are the same integer function
    from ivos_w_amd.models.momory_pool import draw_indices
    assert lib.ivosw_replay_draw_state_bytes() == 16
    for seed, counter, n in ((0, 0, 1), (2019, 0, 50000), (0xFFFF_FFFF_FFFF_FFFF, 0xFFFF_FFFF, 3000), (0x1234_5678_9ABC_DEF1, 77, 2 ** 31 - 1)):
        py = draw_indices(seed, counter, 40, n)
        assert py.min() >= 0 and py.max() < n
        assert [int(lib.ivosw_replay_draw_index(seed, counter, b, n)) for b in range(40)] == [int(v) for v in py]


def test_argument_errors_do_not_touch_the_gpu(handle):
    lib = L.lib()
    assert lib.ivosw_brain_forward(None, None, 1, 1, None, None, 0, None) == -1
    assert b"null" in lib.ivosw_last_error()
    assert lib.ivosw_clamp_adam(None, None, None, None, 10, 1, 0.0, 0.9, 0.999, 1e-8, 0.0, 1.0, 1.0, None) == -1
    assert lib.ivosw_assess_forward(None, 1, None, None, 1, 480, 854, None, None, 0, 0, 0, None, None) == -1


def test_product_path_has_no_cpu_fallback():
    import torch
    from ivos_w_amd.models.agent import Brain
    from ivos_w_amd.models.assessment import AssessNet
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Brain()(torch.zeros(1, 4, 2))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        AssessNet().eval()(torch.zeros(1, 3, 32, 32), torch.zeros(1, 32, 32))


def test_product_code_never_imports_the_oracle():
    bad = []
    for base in (os.path.join(ROOT, "ivos-w_amd"),):
        for dp, _, fs in os.walk(base):
            for f in fs:
                if f.endswith(".py") and re.search(r"^\s*(from|import)\s+oracle\b", open(os.path.join(dp, f)).read(), re.M):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_stage_kernel_isa_has_no_use_of_an_in_flight_asm_ds_read(tmp_path):
    """res2_stage.hip reads its MFMA fragments with inline-asm ds_reads and counted lgkmcnt waits.  hipcc believes an asm's output is valid
    at once, so under register pressure it may spill / copy / reuse the destination BEFORE the data has landed (the persistent variant of
    the kernel computed garbage that way).  Build check: compile the file to gfx950 ISA and let tools/asm_inflight_scan.py look for any
    instruction that touches the destination of a ds_read still in flight."""
    import shutil
    import subprocess
    import sys
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this host")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    asm = str(tmp_path / "res2_stage.s")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-S", "--cuda-device-only",
                        os.path.join(root, "ivos-w_amd", "csrc", "res2_stage.hip"), "-o", asm], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    s = subprocess.run([sys.executable, os.path.join(root, "tools", "asm_inflight_scan.py"), asm], capture_output=True, text=True, timeout=300)
    assert s.returncode == 0 and "res2_stage_kernel" in s.stdout, s.stdout[-2000:]


def test_shipped_kernels_scratch_budget(handle):
    """scratch spills in the hot kernels mean a compiler update can move their timing unnoticed.  The figures are read
    from the code objects inside the SHIPPED library (tools/kernel_resources.py: no compile).  Every kernel must be spill-free except the
    ones listed here with their present counts (all of them loop-invariant addresses parked in the prologue and reloaded at phase
    boundaries, never inside a k-loop) - a RATCHET: a count may shrink, never grow, and no new kernel may start spilling.  The one-wave-per-SIMD
    kernels must really have their accumulators in AGPRs."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(root, "tools", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    if not all(os.path.exists(os.path.join(kr.LLVM, t)) for t in ("llvm-objdump", "llvm-readelf")):
        pytest.skip("no ROCm LLVM tools on this host")
    assert os.path.exists(L.LIB_PATH) and os.path.exists(L.PROBE_LIB_PATH)         # (the `handle` fixture built them if they were missing)
    table = kr.kernel_table(L.LIB_PATH)
    assert len(table) > 40
    # the product library carries no micro-benchmark kernel (they live in libivosw_probe.so: VERDICT round 5, item 10)
    assert not [n for n in table if "gemm_bt_kernel" in n or "gemm_bt_persist" in n]
    allowed = {                                       # substring of the mangled name -> ceiling
        "res2_stage_kernelILb1ELi0": 10, "res2_stage_kernelILb0ELi0": 10, "stage_first_kernel": 23, "bneck_wide_stage_kernelILi256ELi16ELi1": 25,
        "bneck_wide_kernelILi256ELi16ELi1": 1, "bneck_halo64s_kernelILb1ELb0ELi64ELb0": 1, "bneck_halo128s_kernel": 31,
        "conv_igemm_ws_kernelIfLi4ELi2ELi2ELi2ELi3ELi8ELb1": 1,
    }
    over = {}
    for name, r in table.items():
        cap = max([v for k, v in allowed.items() if k in name] or [0])
        if r["spill"] > cap:                          # (SGPR spills go to VGPR lanes, not to memory: not counted)
            over[name] = (r["spill"], cap)
        assert r["lds"] <= 163840 and r["vgpr"] <= 512
    assert not over, over
    # the one-wave-per-SIMD contraction (gemm_bt.h, probe library) really keeps its 4 x 4 accumulator tiles in the AGPR half of the file
    ptable = kr.kernel_table(L.PROBE_LIB_PATH)
    bt = [r for n, r in ptable.items() if "gemm_bt_kernel" in n]
    assert bt and all(r["agpr"] == 256 and r["spill"] == 0 and r["scratch"] == 0 and r["lds"] == 131072 for r in bt), bt
    # the register-chained res2 kernel (round 6): one wave per SIMD, no scratch, the whole LDS map inside 160 KB
    rc = [r for n, r in table.items() if "res2_chain_kernel" in n]
    assert len(rc) == 4 and all(r["spill"] == 0 and r["scratch"] == 0 and r["vgpr"] > 256 and r["lds"] <= 163840 for r in rc), rc
    # the default-path kernels this round touched: spill-free
    for key in ("bneck_halo_kernelILi128ELb1", "bneck_halo_kernelILi128ELb0", "conv1x1_wide_kernel"):
        hits = [r for n, r in table.items() if key in n]
        assert hits and all(r["spill"] == 0 and r["scratch"] == 0 for r in hits), key
