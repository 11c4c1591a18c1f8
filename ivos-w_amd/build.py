"""Builds libivosw_hip.so (gfx950) in-tree with hipcc; cross-compiles without a GPU."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libivosw_hip.so")
PROBE_LIB = os.path.join(HERE, "libivosw_probe.so")     # superset build with the tuning probes of include/ivosw_probe.h (tools/, one GPU test)
PROBE_SOURCES = ["assess.hip", "bottleneck.hip", "bottleneck_wide.hip", "brain.hip"]      # the translation units that hold a probe entry
SOURCES = ["capi.cpp", "brain.hip", "dqn.hip", "assess_front.hip", "conv.hip", "bottleneck.hip", "bottleneck_wide.hip", "res2_stage.hip", "res2_chain.hip", "gemm_8phase.hip", "stage_first.hip", "stem.hip", "assess.hip", "metrics.hip", "seg_epilogue.hip", "p2p.hip"]  # missing files are skipped
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
# per-file flags: res2_chain.hip runs one wave per SIMD (512 registers) and wants its MFMA accumulators in VGPRs (no v_accvgpr_read per
# epilogue value: - 10 % on the block body, tools/ubench/chain_bench.hip)
EXTRA = {"res2_chain.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}
if os.environ.get("IVOSW_ABLATION") == "1":      # tuning builds only: compiles the ablation switches in (common.h)
    FLAGS.append("-DIVOSW_ABLATION=1")


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "ivosw.h"))
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs = []
    for s in srcs:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s + ".o")
        if force or _stale(obj, [src] + headers):
            cmd = [_hipcc()] + FLAGS + EXTRA.get(s, []) + (["-x", "hip"] if s.endswith(".cpp") else []) + ["-c", src, "-o", obj]
            jobs.append(cmd)
        if s in PROBE_SOURCES:                      # the same unit once more with the probe entries compiled in
            pobj = os.path.join(OBJ, s + ".probe.o")
            if force or _stale(pobj, [src] + headers):
                jobs.append([_hipcc()] + FLAGS + EXTRA.get(s, []) + ["-DIVOSW_PROBES=1", "-c", src, "-o", pobj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r
    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s + ".o") for s in srcs]
    if force or jobs or _stale(LIB, objs):
        run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    pobjs = [os.path.join(OBJ, s + (".probe.o" if s in PROBE_SOURCES else ".o")) for s in srcs]
    if force or jobs or _stale(PROBE_LIB, pobjs):
        run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", PROBE_LIB] + pobjs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
