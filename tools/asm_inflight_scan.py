"""Build check for res2_stage.hip: inline-asm ds_reads are invisible to hipcc, which may spill / copy / reuse their destination
registers before the data has landed.  Scans the gfx950 ISA (hipcc -S --cuda-device-only) for any instruction that touches the
destination of a ds_read still in flight (not yet retired by an s_waitcnt lgkmcnt).  usage: asm_inflight_scan.py file.s"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
# per kernel
start = [i for i, l in enumerate(lines) if re.match(r"^_ZN5ivosw17res2_stage_kernel.*:", l)]
for si, s0 in enumerate(start):
    e0 = next(i for i in range(s0, len(lines)) if "s_endpgm" in lines[i])
    name = lines[s0].split(":")[0]
    pending = []   # list of (regset, order) in issue order
    bad = 0
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
        if op.startswith("ds_read"):
            pending.append(regs(toks[0]))
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
    print(name, "suspicious:", bad)
