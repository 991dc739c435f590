"""In-kernel phase timeline of linear_f16x3 / gemm_f16x3_stream (GPU box; needs UNIVS_HIP_LIB=univs_amd/libunivs_hip_trace*.so,
built by `python -m univs_amd.build --ablate trace`).  For a few (workgroup, wave) pairs: clocks from kernel entry to the end of
the W staging, per tile to the end of the k loop and of the epilogue; entry skew and total in 100-MHz real-time ticks."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univs_amd import _lib, ops, synth  # noqa: E402

SLOTS, STAMPS = 48, 64


def read(fn_name, clear=False):
    lib = _lib.load()
    fn = getattr(lib, fn_name)
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    buf = (ctypes.c_ulonglong * (SLOTS * STAMPS))()
    rc = fn(buf, 1 if clear else 0)
    assert rc == 0, rc
    return list(buf)


def show(name, data):
    print(f"== {name}")
    base_real = min(data[s * STAMPS + 62] for s in range(SLOTS) if data[s * STAMPS + 0])
    for s in range(SLOTS):
        st = data[s * STAMPS:(s + 1) * STAMPS]
        if not st[0]:
            continue
        wgx, wgy, wave = s // 16, (s // 8) % 2, s % 8
        if wave not in (0, 7):
            continue
        t0 = st[0]
        nt = st[63]
        rel = lambda v: (v - t0) if v else None   # noqa: E731
        tiles = []
        prev = st[2]
        for t in range(min(nt, 14)):
            k, e = st[3 + 2 * t], st[4 + 2 * t]
            if not k:
                break
            tiles.append(f"{k - prev}+{e - k}")
            prev = e
        print(f"  wg x{wgx} y{wgy} wave{wave}: entry +{(st[62] - base_real) * 10} ns, total {(st[61] - st[62]) * 10} ns | loads issued {rel(st[1])} clk, "
              f"W staged {rel(st[2])} clk, tiles {nt}: kloop+epilogue clk " + " ".join(tiles))


def main():
    dev = torch.device("cuda:0")
    T = 5
    for name, M, K, N, act, res in (("s3_qkv 18400x384->1152", T * 3680, 384, 1152, None, False), ("s3_fc1", T * 3680, 384, 1536, "gelu", False),
                                    ("s3_fc2 (stream)", T * 3680, 1536, 384, None, True), ("s3_proj (stream)", T * 3680, 384, 384, None, True),
                                    ("enc_value 96600x256->256", T * 19320, 256, 256, None, False), ("dec_kv_l8 73600x256->768", T * 14720, 256, 768, None, False),
                                    ("s4_qkv (stream)", T * 920, 768, 2304, None, False)):
        x = synth.normal(f"gt/x/{M}x{K}", (M, K)).to(dev)
        w = synth.normal(f"gt/w/{N}x{K}", (N, K), std=K ** -0.5).to(dev)
        b = synth.normal(f"gt/b/{N}", (N,)).to(dev)
        r = synth.normal(f"gt/r/{M}x{N}", (M, N)).to(dev) if res else None
        for _ in range(3):
            ops.linear_fused(x, w, b, act=act, residual=r)
        torch.cuda.synchronize()
        read("univs_debug_gemm_trace_linear", clear=True)
        read("univs_debug_gemm_trace_stream", clear=True)
        a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a_.record()
        ops.linear_fused(x, w, b, act=act, residual=r)
        b_.record()
        torch.cuda.synchronize()
        print(f"{name}: {a_.elapsed_time(b_) * 1e3:.1f} us by events (one launch)")
        for fn in ("univs_debug_gemm_trace_linear", "univs_debug_gemm_trace_stream"):
            d = read(fn)
            if any(d):
                show(fn.rsplit("_", 1)[1], d)


if __name__ == "__main__":
    main()
