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
SOURCES = ["capi.hip", "msda_fwd.hip", "msda_tiled2.hip", "msda_strips.hip", "msda_heads.hip", "msda_bwd.hip", "msda_prepare.hip", "mask_decode.hip", "linear_split.hip", "linear_f16x3.hip", "gemm_f16x3_stream.hip", "gemm_f16x3_tile.hip", "mlp_f16x3.hip", "small_linear.hip", "cross_attn.hip", "window_attn.hip", "window_attn_f16.hip", "resample.hip", "layer_norm.hip", "group_norm.hip", "softmax.hip", "proca_attn.hip", "prompt_sampler.hip", "transpose.hip", "mask_stats.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "msda_common.h"), os.path.join(CSRC, "msda_geometry.h"), os.path.join(CSRC, "msda_dev.h"), os.path.join(CSRC, "msda_strips_geom.h"), os.path.join(CSRC, "msda_heads_geom.h"), os.path.join(CSRC, "config.h"), os.path.join(CSRC, "window_attn.h"), os.path.join(CSRC, "f16x3.h"),
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


def _flags(defines=()):
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
            "-I", os.path.join(HERE, "..", "include"), *[f"-D{d}" for d in defines]]


def build(force=False, verbose=True, lib=LIB, defines=(), tag=""):
    """One object per source (compiled in parallel, rebuilt only when the source or a header is newer), then one link.
    `defines` / `tag` / `lib`: instrumented builds of the same sources for timing experiments (tools/gpu_runs)."""
    if not force and not defines and not needs_build():
        return lib
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(HERE, "_obj" + tag)
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    hdr_t = max(os.path.getmtime(h) for h in HEADERS)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]

    def compile_one(s):
        src, obj = os.path.join(CSRC, s), os.path.join(objdir, s.replace(".hip", ".o"))
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_t):
            return obj
        cmd = [hipcc, *_flags(defines), "-c", src, "-o", obj]
        if verbose:
            print("[univs_amd.build]", " ".join(cmd), file=sys.stderr, flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-fno-gpu-rdc", *objs, "-o", lib + ".tmp"]
    if verbose:
        print("[univs_amd.build]", " ".join(cmd), file=sys.stderr, flush=True)
    subprocess.run(cmd, check=True)
    os.replace(lib + ".tmp", lib)
    return lib


ABLATIONS = {"plainsplit": ("UNIVS_SPLIT_PLAIN",), "nosplit": ("UNIVS_ABLATE_NOSPLIT",), "nomfma": ("UNIVS_ABLATE_NOMFMA",),
             "nosplit_nomfma": ("UNIVS_ABLATE_NOSPLIT", "UNIVS_ABLATE_NOMFMA"), "trace": ("UNIVS_TRACE_GEMM",), "vtpad8": ("UNIVS_WH_VT_PAD=8", "UNIVS_XA_VS=40"), "gs256": ("UNIVS_GS_THREADS=256",),
             "trace_nosplit_nomfma": ("UNIVS_TRACE_GEMM", "UNIVS_ABLATE_NOSPLIT", "UNIVS_ABLATE_NOMFMA"),
             # csrc/msda_heads.hip (S6_ABLATE bits): no gather stream / no row movement / no reduction + stores / no records
             "heads_nostream": ("S6_ABLATE=1",), "heads_norows": ("S6_ABLATE=2",), "heads_noreduce": ("S6_ABLATE=4",),
             "heads_norecords": ("S6_ABLATE=8",), "heads_nostream_norows": ("S6_ABLATE=3",), "heads_onlyrows": ("S6_ABLATE=13",),
             "heads_skeleton": ("S6_ABLATE=15",), "heads_trace": ("S6_TRACE",),
             # csrc/mlp_f16x3.hip: the weight stream staged through registers (round 5's form) instead of LDS-DMA
             "mlp_regstage": ("UNIVS_MLP_REGSTAGE",), "mlp_trace": ("UNIVS_TRACE_MLP",),
             "mlp_trace_nomfma": ("UNIVS_TRACE_MLP", "ML_PS_ABL=1"), "mlp_trace_noreads": ("UNIVS_TRACE_MLP", "ML_PS_ABL=2")}


def build_ablation(name, verbose=False):
    """libunivs_hip_<name>.so beside the product library: the same sources with a timing-experiment define (csrc/f16x3.h).
    Loaded through UNIVS_HIP_LIB by tools/kbench.py; never by the product."""
    lib = os.path.join(HERE, f"libunivs_hip_{name}.so")
    return build(force=False, verbose=verbose, lib=lib, defines=ABLATIONS[name], tag="_" + name)


if __name__ == "__main__":
    if "--ablate" in sys.argv:
        print(build_ablation(sys.argv[sys.argv.index("--ablate") + 1]))
    else:
        build(force="--force" in sys.argv)
        print(LIB)
