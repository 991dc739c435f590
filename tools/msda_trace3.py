"""Phase timing of msda_fwd_tiled3 / msda_fwd_tiled4 (--gen 4) from inside the kernel (run on the GPU box).
`python tools/msda_trace3.py --build` (no GPU needed) compiles the product sources with -DUNIVS_MSDA_TRACE into
tools/_trace/libunivs_hip_trace.so; `python tools/msda_trace3.py [--ablate N]` loads that build through UNIVS_HIP_LIB, launches
the kernel at the BASELINE config-2 geometry and prints the mean s_memtime deltas (shader clocks) of the second item of
every workgroup: gather wave 0 and fill wave 0, per level step."""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "_trace", "libunivs_hip_trace.so")


def build():
    sys.path.insert(0, ROOT)
    from univs_amd import build as b
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    srcs = [os.path.join(b.CSRC, s) for s in b.SOURCES]
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DUNIVS_MSDA_TRACE",
                           "-I", os.path.join(ROOT, "include"), *srcs, "-o", OUT])
    print("built", OUT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--ablate", type=int, default=0)
    ap.add_argument("--variant", type=int, default=1)
    ap.add_argument("--gen", type=int, default=3)
    args = ap.parse_args()
    if args.build:
        return build()
    os.environ["UNIVS_HIP_LIB"] = OUT
    os.environ["UNIVS_MSDA_TILED"] = str(args.gen)
    os.environ["UNIVS_MSDA_T3_VARIANT"] = str(args.variant)
    os.environ["UNIVS_MSDA_ABLATE"] = str(args.ablate)
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from tests import cases      # development tool: the tests' input builders
    from univs_amd import _lib, ops
    dev = torch.device("cuda:0")
    T = 5
    case = dict(name="kb", shapes=[(23, 40), (46, 80), (92, 160)], N=T, M=8, D=32, P=4, encoder=True, far=False)
    value, shapes, lsi, loc, attn = cases.msda_inputs(case)
    value, loc, attn = value.to(dev), loc.to(dev), attn.to(dev)
    for _ in range(5):
        ops.ms_deform_attn_forward(value, shapes, lsi, loc, attn)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    ops.ms_deform_attn_forward(value, shapes, lsi, loc, attn)
    b.record()
    torch.cuda.synchronize()
    assert ops.msda_last_tiled_generation() == args.gen
    n = 256
    buf = (ctypes.c_ulonglong * (32 * n))()
    lib = _lib.load()
    reader = getattr(lib, f"univs_msda_trace{args.gen}_read")
    reader.restype = ctypes.c_int
    assert reader(buf, n) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(n, 32).astype(np.int64)
    print(f"gen {args.gen} ablate {args.ablate} variant {args.variant}: launch {a.elapsed_time(b) * 1e3:.1f} us")
    if args.gen == 4:
        names = {2: "level 0 (+ next inputs)", 3: "level 1 (+ next rows)", 4: "level 2 (+ lists)", 5: "(to barrier A)", 6: "barrier A wait",
                 7: "wait for loads + commit rows", 8: "reduce + store + barrier B"}
        prev = t[:, 0]
        for i in (2, 3, 4, 5, 6, 7, 8):
            print(f"  {names[i]:38s} {(t[:, i] - prev).mean():8.0f} clk")
            prev = t[:, i]
        print(f"  item: {(t[:, 8] - t[:, 0]).mean():.0f} clk")
        life = t[:, 29] - t[:, 28]
        print(f"  workgroup lifetime: mean {life.mean():.0f}, min {life.min()}, max {life.max()} clk")
        return
    g, f = t[:, :15].reshape(n, 3, 5), t[:, 16:28].reshape(n, 3, 4)
    names_g = ["entry+records" if args.gen == 3 else "rows+records", "gathers", "miss/out", "barrier wait"]
    names_f = ["entry+commit", "issue loads", "barrier wait"]
    for k in range(3):
        dg = np.diff(g[:, k], axis=1)
        df = np.diff(f[:, k], axis=1)
        step = (g[:, k, 4] - g[:, k, 0]).mean()
        print(f"  step {k}: {step:7.0f} clk | gather wave: " + ", ".join(f"{nm} {dg[:, i].mean():6.0f}" for i, nm in enumerate(names_g))
              + (f" | rows: wait {(f[:, k, 0] - g[:, k, 0]).mean():6.0f}, commit {(f[:, k, 1] - f[:, k, 0]).mean():6.0f}, request {(f[:, k, 2] - f[:, k, 1]).mean():6.0f}, records {(g[:, k, 1] - f[:, k, 2]).mean():6.0f}"
                 if args.gen == 4 else " | fill wave: " + ", ".join(f"{nm} {df[:, i].mean():6.0f}" for i, nm in enumerate(names_f))))
    print(f"  item: {(g[:, 2, 4] - g[:, 0, 0]).mean():.0f} clk")
    if args.gen == 4:
        life = t[:, 29] - t[:, 28]
        span = t[:, 29].max() - t[:, 28].min()
        print(f"  workgroup lifetime: mean {life.mean():.0f}, min {life.min()}, max {life.max()} clk; first start -> last end {span} clk "
              f"= {span / (a.elapsed_time(b) * 1e3):.0f} clk/us of launch time")
        print(f"  item top (issue the next item's query list + inputs) -> step 0: {(g[:, 0, 0] - t[:, 15]).mean():.0f} clk")


if __name__ == "__main__":
    main()
