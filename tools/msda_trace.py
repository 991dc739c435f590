"""Phase timing of the LDS-tiled MSDA kernel from inside the kernel (run on the GPU box).

`python tools/msda_trace.py --build` (here, no GPU) compiles the product sources with -DUNIVS_MSDA_TRACE
into tools/_trace/libunivs_hip_trace.so; `python tools/msda_trace.py` (GPU box) loads that build through
UNIVS_HIP_LIB, launches the kernel at the BASELINE config-2 geometry and prints, per phase, the mean /
p90 s_memtime delta over all workgroups plus the kernel's wall time, so that stamps can be converted
to a share of the launch."""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "_trace", "libunivs_hip_trace.so")
NAMES = {0: "prologue(box math)", 1: "qglob+sync", 2: "issue(0) issued", 3: "L0 top", 4: "L0 commit+sync",
         5: "L0 phaseB", 6: "L1 barrier", 7: "L1 commit+sync", 8: "L1 phaseB", 9: "L2 barrier",
         10: "L2 commit+sync", 11: "L2 phaseB", 15: "stores"}


def build():
    sys.path.insert(0, ROOT)
    from univs_amd import build as b
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    srcs = [os.path.join(b.CSRC, s) for s in b.SOURCES]
    cmd = [b._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DUNIVS_MSDA_TRACE", *os.environ.get("UNIVS_TRACE_DEFS", "").split(),
           "-I", os.path.join(ROOT, "include"), *srcs, "-o", OUT]
    subprocess.check_call(cmd)
    print("built", OUT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--T", type=int, default=5)
    args = ap.parse_args()
    if args.build:
        return build()
    os.environ["UNIVS_HIP_LIB"] = OUT
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from tests import cases      # development tool: the tests' input builders
    from univs_amd import _lib, ops
    dev = torch.device("cuda:0")
    T = args.T
    shapes = [(23, 40), (46, 80), (92, 160)]
    case = dict(name="kb", shapes=shapes, N=T, M=8, D=32, P=4, encoder=True, far=False)
    value, shapes, lsi, loc, attn = cases.msda_inputs(case)
    value, loc, attn = value.to(dev), loc.to(dev), attn.to(dev)
    ops.msda_set_impl(2)
    for _ in range(5):
        ops.ms_deform_attn_forward(value, shapes, lsi, loc, attn)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    ops.ms_deform_attn_forward(value, shapes, lsi, loc, attn)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b)
    nblk = int(os.environ.get("UNIVS_MSDA_GRID", "256"))
    gen2 = os.environ.get("UNIVS_MSDA_TILED", "2") != "1"
    buf = np.zeros((nblk, 16), dtype=np.uint64)
    lib = _lib.load()
    fn = lib.univs_msda_trace2_read if gen2 else lib.univs_msda_trace_read
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    rc = fn(buf.ctypes.data, nblk)
    assert rc == 0, rc
    t = buf.astype(np.int64)
    # stamps 13 / 14: workgroup start / end; 0, 3..11, 15: phases of the workgroup's SECOND item (steady state)
    span = t[:, 14].max() - t[:, 13].min()
    life = t[:, 14] - t[:, 13]
    print(f"kernel (1 launch, events incl. launch overhead): {ms * 1e3:.1f} us;  stamp span {span} ticks "
          f"-> {span / (ms * 1e3):.1f} ticks/us")
    print(f"workgroup lifetime: mean {life.mean():.0f}  min {life.min()}  max {life.max()} ticks "
          f"(start skew {t[:, 13].max() - t[:, 13].min()}, end skew {t[:, 14].max() - t[:, 14].min()})")
    tot = t[:, 15] - t[:, 0]
    print(f"steady-state item: mean {tot.mean():.0f}  p10 {np.percentile(tot, 10):.0f}  p90 {np.percentile(tot, 90):.0f} ticks")
    names = {3: "top (decode, acc=0, barrier)", 4: "L0 commit+sync", 5: "L0 loads+gathers", 6: "L1 barrier", 7: "L1 commit+sync",
             8: "L1 loads+gathers", 9: "L2 barrier", 10: "L2 commit(+qglob)+sync", 11: "L2 prefetch+gathers", 15: "reduce+stores"}
    if gen2:   # one barrier per step: stamps are (step top, after the barrier, after the gathers)
        names = {3: "top (item turnover)", 4: "step0 commit + issue loads", 5: "step0 barrier wait", 6: "-", 7: "step1 commit + issue loads",
                 8: "step1 barrier wait", 9: "-", 10: "step2 commit + issue loads", 11: "step2 barrier wait", 15: "end"}
    order = [0, 3, 4, 5, 6, 7, 8, 9, 10, 11, 15]
    if gen2:
        print(f"  consumers (wave 8): gathers step0 {np.mean(t[:, 2] - t[:, 1]):.0f}, barrier wait {np.mean(t[:, 6] - t[:, 2]):.0f}, "
              f"gathers step1 {np.mean(t[:, 9] - t[:, 6]):.0f}, barrier wait {np.mean(t[:, 12] - t[:, 9]):.0f}")
        print("  producers (wave 0):")
    for i, j in zip(order[:-1], order[1:]):
        d = t[:, j] - t[:, i]
        print(f"  {names[j]:<30s} mean {d.mean():8.0f}  p10 {np.percentile(d, 10):8.0f}  p90 {np.percentile(d, 90):8.0f}  "
              f"share {d.sum() / tot.sum() * 100:5.1f} %")


if __name__ == "__main__":
    main()
