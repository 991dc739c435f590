"""Static issue-cost estimate of a gfx950 kernel's ISA (from `hipcc -save-temps`), per region between s_barrier
instructions.  Costs per wave-instruction from profiles/r02_gfx950_issue_costs.txt (clocks per SIMD)."""
import re
import sys

FAST = {"v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_add_u32", "v_sub_u32",
        "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_accvgpr_write_b32"}
QUARTER = {"v_rcp_f32", "v_exp_f32", "v_log_f32", "v_rsq_f32", "v_sqrt_f32", "v_rcp_iflag_f32"}


def cost(op, line):
    base = re.sub(r"_e32$|_e64$", "", op)
    if "dpp" in line or "row_" in line or "quad_perm" in line:
        return 4.6, "dpp"
    if base in QUARTER:
        return 8.2, "trans"
    if base.startswith("v_pk_"):
        return 4.5, "pk"
    if base in FAST and not re.search(r"\bs\d+|\bs\[", line.split(None, 1)[1] if " " in line.strip() else ""):
        return 2.25, "fast"
    if base.startswith("v_"):
        return 4.3, "slow"
    return 0.0, "other"


def main(path, pattern=""):
    text = open(path).read()
    kernels = re.split(r"\n(?=_Z\w+:)", text)
    for k in kernels:
        name = k.split(":", 1)[0]
        if not name.startswith("_Z") or (pattern and pattern not in name):
            continue
        lines = k.split("\n")
        regions, cur = [], []
        for l in lines:
            cur.append(l)
            if "s_barrier" in l or "s_endpgm" in l:
                regions.append(cur)
                cur = []
        print(name[:110])
        for i, reg in enumerate(regions):
            tot = {"fast": 0, "slow": 0, "dpp": 0, "pk": 0, "trans": 0}
            clk = 0.0
            salu = ds = vmem = 0
            for l in reg:
                t = l.strip().split()
                if not t or t[0].startswith((";", ".")) or t[0].endswith(":"):
                    continue
                op = t[0]
                if op.startswith("v_"):
                    c, cls = cost(op, l)
                    tot[cls] += 1
                    clk += c
                elif op.startswith("s_") and op not in ("s_waitcnt", "s_nop", "s_barrier"):
                    salu += 1
                elif op.startswith("ds_"):
                    ds += 1
                elif op.startswith(("global_", "buffer_", "scratch_")):
                    vmem += 1
            print(f"  region {i:2d}: {len(reg):5d} lines  VALU fast {tot['fast']:4d} slow {tot['slow']:4d} dpp {tot['dpp']:4d} pk {tot['pk']:3d} "
                  f"trans {tot['trans']:3d} -> {clk:7.0f} clk | SALU {salu:4d} DS {ds:4d} VMEM {vmem:3d}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
