"""Condensed view of one kernel of a hipcc -S listing: memory instructions, waits, barriers, branches, loop labels, and counts of MFMA / VALU
between them -- to read where s_waitcnt vmcnt(N) sits in a pipelined loop.   python tools/isa_digest.py FILE.s 'kernel-name-substring' [start end]"""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
hi = int(sys.argv[4]) if len(sys.argv) > 4 else 10 ** 9
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(";")[0].strip().endswith(":"))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
keep = re.compile(r"s_waitcnt|buffer_load|global_load|buffer_store|global_store|s_barrier|ds_write|ds_read|s_cbranch|^\.LBB|scratch_")
mf = va = 0
for n, l in enumerate(lines[start:end]):
    if not (lo <= n <= hi):
        continue
    s = l.strip()
    if s.startswith("v_mfma"):
        mf += 1
    elif s.startswith("v_"):
        va += 1
    elif keep.search(s):
        if mf or va:
            print(f"        [{mf} mfma, {va} valu]")
            mf = va = 0
        print(f"{n:5d} " + re.sub(r"\s+", " ", s.split(";")[0])[:90])
