"""Build check for res2_stage.hip: inline-asm ds_reads are invisible to hipcc, which may spill / copy / reuse their destination
registers before the data has landed.  Scans the gfx950 ISA (hipcc -S --cuda-device-only) for any instruction that touches the
destination of a ds_read still in flight (not yet retired by an s_waitcnt lgkmcnt).  The scan is linear (it does not follow
branches: a hit right behind a label may belong to another path) and models lgkmcnt as an in-order queue (LDS stores and scalar
loads occupy slots).  usage: asm_inflight_scan.py file.s   (exit status 1 when anything is flagged)"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
# per kernel
start = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and "ivosw" in l]
for si, s0 in enumerate(start):
    try:
        e0 = next(i for i in range(s0, len(lines)) if "s_endpgm" in lines[i])
    except StopIteration:
        continue
    name = lines[s0].split(":")[0]
    pending = []   # list of (regset, order) in issue order
    bad = 0
    seen_barrier, smem_late = False, 0
    def regs(tok):
        m = re.match(r"v\[(\d+):(\d+)\]", tok)
        if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.match(r"v(\d+)$", tok)
        if m: return {int(m.group(1))}
        return set()
    for i in range(s0, e0):
        l = lines[i].strip()
        if not l or l.startswith(";") or l.startswith("."): continue
        op = l.split()[0]
        toks = [t.strip(",") for t in l.split()[1:]]
        if op.startswith("ds_read") or op.startswith("ds_load"):
            pending.append(regs(toks[0]))
            continue
        if op == "s_barrier":
            seen_barrier = True
        if (op.startswith("s_load") or op.startswith("s_buffer_load")) and seen_barrier and "res2_stage" in name:
            smem_late += 1                          # a scalar load behind the prologue: lgkmcnt(N) waits of the fragment pipeline would miscount
            print(name[-40:], "line", i - s0, ":", l[:100], "scalar load behind the first barrier")
        if op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load"):
            pending.append(set())                   # occupies a slot of lgkmcnt without a vector destination (LDS stores / atomics, scalar loads)
            continue
        m = re.match(r"s_waitcnt.*lgkmcnt\((\d+)\)", l)
        if m:
            n = int(m.group(1))
            pending = pending[len(pending) - n:] if n < len(pending) else pending
            if n == 0: pending = []
            continue
        if op in ("s_barrier",): continue
        used = set()
        for t in toks:
            used |= regs(t)
        for pr in pending:
            if used & pr:
                bad += 1
                if bad <= 12: print(name[-40:], "line", i - s0, ":", l[:100], "touches pending", sorted(pr)[:4])
    # The prologue's counted VMEM wait (res2_stage.hip: wait_vmcnt<21> between the halo's LDS-DMA and the first barrier) is only right
    # while EXACTLY that many vector-memory instructions sit between the last halo DMA and the wait: with fewer (loads merged, sunk or
    # CSE'd by another hipcc) the barrier would release before the halo has landed.  Check the first counted wait behind an LDS-DMA.
    vm_bad = 0
    if "res2_stage" in name:
        last_dma, n_vmem, checked = None, 0, False
        for i in range(s0, e0):
            l = lines[i].strip()
            if not l or l.startswith(";") or l.startswith("."): continue
            op = l.split()[0]
            is_dma = (op.startswith("global_load_lds") or (op.startswith("buffer_load") and " lds" in l))
            if is_dma:
                last_dma, n_vmem = i, 0
                continue
            if op.startswith(("global_load", "buffer_load", "global_store", "buffer_store", "global_atomic", "flat_")):
                n_vmem += 1
                continue
            m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", l)
            if m and last_dma is not None:
                n = int(m.group(1))
                if n != 0 and n != n_vmem:
                    vm_bad += 1
                    print(name[-40:], "line", i - s0, ":", l[:60], "but", n_vmem, "vector-memory instructions follow the last LDS-DMA")
                checked = True
                break
            if op == "s_barrier":
                break
        if not checked:
            vm_bad += 1
            print(name[-40:], ": no s_waitcnt vmcnt between the halo LDS-DMA and the first barrier")
    print(name, "suspicious:", bad, "late scalar loads:", smem_late, "prologue vmcnt mismatches:", vm_bad)
    total = globals().get("total", 0) + bad + smem_late + vm_bad

sys.exit(1 if globals().get("total", 0) else 0)
