"""CPU (hipcc cross-compiles gfx950 without a GPU): properties of the COMPILED three-product kernels that their speed rests on and that a
source edit or a compiler update can silently lose (hazards 17-22; profiles/r05_gemm_negative_results_v2.txt):
  * the pipelined loop of gemm_f16x3_tile has counted vector-memory waits -- no `s_waitcnt vmcnt(0)` between its barriers (hipcc drains
    every outstanding load when a load of the loop sits under a branch) -- and no scratch;
  * the operand split is the two-instruction form (v_fma_mix{lo,hi}_f16), the A-fragment LDS reads of linear_f16x3 carry immediate
    offsets, and the hot instantiations do not spill more than a handful of registers."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "univs_amd", "csrc")
HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")


def listing(tmp_path_factory, source):
    out = tmp_path_factory.mktemp("isa") / (source + ".s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
           os.path.join(CSRC, source), "-o", str(out), "-Rpass-analysis=kernel-resource-usage"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    usage = {}
    name = None
    for line in p.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            usage[name] = {}
        for key in ("VGPRs", "VGPRs Spill", "ScratchSize [bytes/lane]"):
            m = re.search(re.escape(key) + r": (\d+)", line)
            if m and name:
                usage[name][key] = int(m.group(1))
    return out.read_text().splitlines(), usage


def kernel_body(lines, mangled_part):
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and mangled_part in l and l.split(";")[0].strip().endswith(":"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    return [l.strip() for l in lines[start:end]]


@pytest.fixture(scope="module")
def tile(tmp_path_factory):
    return listing(tmp_path_factory, "gemm_f16x3_tile.hip")


@pytest.fixture(scope="module")
def resident(tmp_path_factory):
    return listing(tmp_path_factory, "linear_f16x3.hip")


@pytest.mark.parametrize("inst", ["ILi4ELi2ELi2ELi2E", "ILi5ELi3ELi4ELi1E", "ILi3ELi3ELi2ELi2E", "ILi4ELi4ELi4ELi1E"])
def test_tile_kernel_loop_has_counted_waits_and_no_scratch(tile, inst):
    lines, usage = tile
    name = next(n for n in usage if "gemm_f16x3_tile" + inst in n)
    assert usage[name]["VGPRs Spill"] == 0 and usage[name]["ScratchSize [bytes/lane]"] == 0, usage[name]
    if inst.endswith("Li2E"):                                  # two workgroups per CU need <= 128 registers
        assert usage[name]["VGPRs"] <= 128, usage[name]
    body = kernel_body(lines, "gemm_f16x3_tile" + inst)
    barriers = [i for i, l in enumerate(body) if l.startswith("s_barrier")]
    assert len(barriers) >= 2                                  # the k loop is unrolled by the number of load slots: one barrier per k-step
    loop = body[barriers[0]:barriers[-1]]
    waits = [int(m.group(1)) for l in loop for m in [re.match(r"s_waitcnt vmcnt\((\d+)\)", l)] if m]
    assert waits and min(waits) >= 3, waits                    # the loads of the next k-steps stay in flight across every wait
    assert not any(l.startswith("scratch_") for l in loop)
    assert sum(l.startswith("v_mfma_f32_16x16x32_f16") for l in loop) > 0


def test_split_is_two_mixed_precision_fmas_per_value(tile):
    lines, _ = tile
    body = kernel_body(lines, "gemm_f16x3_tileILi4ELi2ELi2ELi2E")
    barriers = [i for i, l in enumerate(body) if l.startswith("s_barrier")]
    loop = body[barriers[0]:barriers[1]]                       # one k-step: one column tile of 16 x 32 values per wave = 8 values per lane
    mix = sum(l.startswith("v_fma_mixlo_f16") or l.startswith("v_fma_mixhi_f16") for l in loop)
    assert mix == 16, mix
    assert not any(l.startswith("v_cvt_f32_f16") for l in loop)  # no convert-back of the plain expression's chain


def test_resident_kernel_reads_fragments_with_immediate_offsets(resident):
    lines, usage = resident
    for inst, max_spill in (("ILi8ELi4ELb1E", 8), ("ILi6ELi4ELb1E", 2), ("ILi6ELi3ELb1E", 2)):
        name = next(n for n in usage if "linear_f16x3" + inst in n)
        assert usage[name]["VGPRs Spill"] <= max_spill, (inst, usage[name])
    body = kernel_body(lines, "linear_f16x3ILi8ELi4ELb1E")
    reads = [l for l in body if l.startswith("ds_read_b128")]
    with_imm = [l for l in reads if "offset:" in l]
    assert len(with_imm) >= 64, (len(reads), len(with_imm))   # 16 reads per k-step x 4 unrolled k-steps (the rest: bias / scales in the epilogue)
    # the k loop's waits are counted (16 x loads in flight per wave)
    waits = [int(m.group(1)) for l in body for m in [re.match(r"s_waitcnt vmcnt\((\d+)\)", l)] if m]
    assert sum(w >= 12 for w in waits) >= 12, waits


def test_library_has_no_packed_f32_low_result_from_a_high_half():
    """gfx950 hazard across waves (csrc/common.h: fma_single; tools/race_probe8.py measured it): v_pk_{fma,mul,add}_f32 whose LOW result
    selects the HIGH half of src1 / src2 reads 0 there in lanes 48..63 while another wave of the SIMD issues v_mfma_f32_16x16x32_{f16,bf16}
    -- a kernel on another stream is enough.  hipcc forms such instructions from scalar code, so the BUILT library is scanned (every
    gfx950 code object of libunivs_hip.so, disassembled): a source edit or a compiler update that brings the form back fails here."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import isa_scan
    finally:
        sys.path.pop(0)
    if not os.path.exists(os.path.join(isa_scan.LLVM, "llvm-objdump")) or shutil.which("objcopy") is None:
        pytest.skip("llvm-objdump / objcopy not found")
    from univs_amd import build
    lib = build.build()                                        # (a no-op when the binary is newer than the sources)
    objects = isa_scan.code_objects(lib)
    assert len(objects) >= 20, len(objects)                    # one code object per .hip file
    found = {}
    kernels = 0
    for co in objects:
        text = isa_scan.disassemble(co)
        kernels += len(re.findall(r"^[0-9a-f]+ <_Z\w+>:", text, flags=re.M))
        found.update(isa_scan.vulnerable_by_kernel(text))
    assert kernels > 100, kernels                              # (the scan saw the kernels)
    assert not found, {k: v[:3] for k, v in found.items()}
    # the scanner itself: the measured form, its src1 variant, and forms measured immune
    probe = ("0000 <_Z1kv>:\n v_pk_fma_f32 v[0:1], v[0:1], v[4:5], v[4:5] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n"
             " v_pk_add_f32 v[8:9], v[8:9], v[8:9] op_sel:[0,1] op_sel_hi:[1,0]\n v_pk_mul_f32 v[2:3], v[2:3], v[4:5] op_sel:[1,0]\n"
             " v_pk_fma_f32 v[2:3], v[2:3], v[4:5], v[6:7] op_sel_hi:[1,1,0]\n")
    assert len(isa_scan.vulnerable_by_kernel(probe)["_Z1kv"]) == 2
