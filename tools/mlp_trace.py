"""In-kernel timeline of the phase-shifted fused MLP (csrc/mlp_f16x3.hip: mlp_f16x3_ps) on the encoder FFN's shape (GPU box; needs
UNIVS_HIP_LIB=univs_amd/libunivs_hip_mlp_trace.so, built by `python -m univs_amd.build --ablate mlp_trace`): per wave of workgroup 0 and slot,
s_memtime clocks of: fragments requested -> GEMM done -> activation done -> (waiting for the weight stream) -> (barrier)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univs_amd import _lib, ops, synth  # noqa: E402

SLOTS, ST = 24, 6


def main():
    dev = torch.device("cuda:0")
    M, C, Hd = 96600, 256, 1024
    x = synth.normal("mt/x", (M, C)).to(dev)
    w1 = synth.normal("mt/w1", (Hd, C), std=C ** -0.5).to(dev)
    b1 = synth.normal("mt/b1", (Hd,)).to(dev)
    w2 = synth.normal("mt/w2", (C, Hd), std=Hd ** -0.5).to(dev)
    b2 = synth.normal("mt/b2", (C,)).to(dev)
    for _ in range(3):
        y = ops.mlp_fused(x, w1, b1, w2, b2, act="relu", residual=x)
    assert y is not None
    torch.cuda.synchronize()
    lib = _lib.load()
    fn = lib.univs_debug_mlp_trace
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p]
    buf = (ctypes.c_ulonglong * (8 * SLOTS * ST))()
    assert fn(buf) == 0
    d = list(buf)
    t0 = min(v for v in d if v)
    print("wave slot | start(+clk from first stamp)  gemm  act  wait_w  barrier | slot total")
    for w in range(8):
        for s in range(2, 14):
            st = d[(w * SLOTS + s) * ST:(w * SLOTS + s + 1) * ST]
            if not st[3]:
                continue
            start = st[0] if st[0] else st[3]
            print(f"  w{w} s{s:2d} | {start - t0:8d}  {st[1] - st[0] if st[0] else 0:6d} {st[2] - st[1] if st[0] else 0:6d} {st[4] - st[3]:6d} {st[5] - st[4]:6d} | {st[5] - start:6d}")


if __name__ == "__main__":
    main()
