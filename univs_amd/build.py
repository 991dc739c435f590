"""Builds libunivs_hip.so (gfx950) in-tree with hipcc.  `python -m univs_amd.build [--force]`.

The shared library is the drop-in boundary (include/univs_hip.h); it is rebuilt only when a source
is newer than the binary.  hipcc cross-compiles for gfx950 without a GPU present.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libunivs_hip.so")
SOURCES = ["capi.hip", "msda_fwd.hip", "msda_tiled2.hip", "msda_strips.hip", "msda_bwd.hip", "msda_prepare.hip", "mask_decode.hip", "linear_split.hip", "linear_f16x3.hip", "gemm_f16x3_stream.hip", "mlp_f16x3.hip", "small_linear.hip", "cross_attn.hip", "window_attn.hip", "window_attn_f16.hip", "resample.hip", "layer_norm.hip", "group_norm.hip", "softmax.hip", "transpose.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "msda_common.h"), os.path.join(CSRC, "msda_geometry.h"), os.path.join(CSRC, "msda_dev.h"), os.path.join(CSRC, "msda_strips_geom.h"), os.path.join(CSRC, "config.h"), os.path.join(CSRC, "window_attn.h"), os.path.join(CSRC, "f16x3.h"),
           os.path.join(HERE, "..", "include", "univs_hip.h")]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-fgpu-rdc" if False else "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
           "-I", os.path.join(HERE, "..", "include"), *srcs, "-o", LIB + ".tmp"]
    if verbose:
        print("[univs_amd.build]", " ".join(cmd), file=sys.stderr, flush=True)
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
