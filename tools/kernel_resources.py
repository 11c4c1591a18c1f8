"""Register / scratch / LDS figures of every kernel in the SHIPPED library (libivosw_hip.so), read from the gfx950 code objects
embedded in it (llvm-objdump --offloading + llvm-readelf --notes on a temporary copy: nothing is compiled).
usage: python tools/kernel_resources.py [path/to/libivosw_hip.so]      (tests/test_cabi.py imports `kernel_table`)"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
HERE = os.path.dirname(os.path.abspath(__file__))


def kernel_table(lib=None):
    lib = lib or os.path.join(os.path.dirname(HERE), "ivos-w_amd", "libivosw_hip.so")
    out = {}
    with tempfile.TemporaryDirectory() as td:
        so = os.path.join(td, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], cwd=td, capture_output=True, check=True)
        for f in sorted(os.listdir(td)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(td, f)], capture_output=True, text=True).stdout
            for blk in notes.split("- .agpr_count:")[1:]:
                get = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
                # the kernel's own name: the metadata's `.symbol:` (minus `.kd`), not the first `.name:` (an argument name, should the compiler emit them)
                sym = re.search(r"\.symbol:\s+(\S+?)\.kd\b", blk)
                name = sym.group(1) if sym else re.findall(r"\.name:\s+(\S+)", blk)[-1]
                out[name] = dict(agpr=int(re.match(r"\s*(\d+)", blk).group(1)), vgpr=get("vgpr_count"), sgpr=get("sgpr_count"),
                                 spill=get("vgpr_spill_count"), sspill=get("sgpr_spill_count"), scratch=get("private_segment_fixed_size"),
                                 lds=get("group_segment_fixed_size"))
    return out


def demangle(names):
    for tool in (os.path.join(LLVM, "llvm-cxxfilt"), "c++filt"):
        if shutil.which(tool):
            return subprocess.run([tool], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return list(names)


if __name__ == "__main__":
    t = kernel_table(sys.argv[1] if len(sys.argv) > 1 else None)
    names = sorted(t)
    print(f"{'vgpr':>5} {'agpr':>5} {'spill':>6} {'scratch':>8} {'lds':>7}  kernel")
    for n, d in zip(names, demangle(names)):
        r = t[n]
        print(f"{r['vgpr']:5d} {r['agpr']:5d} {r['spill']:6d} {r['scratch']:8d} {r['lds']:7d}  {d[:150]}")
