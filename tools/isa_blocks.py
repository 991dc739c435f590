"""Basic blocks of one kernel in a hipcc -S listing with instruction counts per class (VALU / SALU / LDS / VMEM / other)
and the branch structure: where a VALU-bound kernel's instructions are.   python tools/isa_blocks.py file.s <symbol substring>"""
import re, sys, collections
src, sym = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sym in l and l.rstrip().split(":")[0].endswith(l.split(":")[0]))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
def cls(op):
    if op.startswith(("v_pk_fma", "v_fma", "v_fmac", "v_pk_mul", "v_pk_add")): return "fma"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_"): return "salu"
    return "other"
blocks, cur = [], None
for i in range(start + 1, end + 1):
    l = lines[i].split(";")[0].rstrip()
    if not l.strip(): continue
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        cur = {"name": m.group(1), "c": collections.Counter(), "br": [], "line": i + 1}
        blocks.append(cur); continue
    if cur is None:
        cur = {"name": "entry", "c": collections.Counter(), "br": [], "line": i + 1}; blocks.append(cur)
    t = l.split()
    if not t or t[0].startswith("."): continue
    # inline-asm statements may hold several instructions per line? (they are printed one per line)
    c = cls(t[0]); cur["c"][c] += 1
    if c == "branch": cur["br"].append((t[0], t[-1]))
idx = {b["name"]: k for k, b in enumerate(blocks)}
tot = collections.Counter()
for k, b in enumerate(blocks):
    back = [f"{op}->{tg}{'(BACK)' if idx.get(tg, 1 << 30) <= k else ''}" for op, tg in b["br"]]
    c = b["c"]; tot.update(c)
    print(f"{b['name']:>10} L{b['line']:<6} fma {c['fma']:3d} valu {c['valu']:4d} salu {c['salu']:4d} lds {c['lds']:3d} vmem {c['vmem']:3d} wait {c['wait']:2d} bar {c['barrier']} | {' '.join(back)}")
print("total", dict(tot))
