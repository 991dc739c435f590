"""Process-level runtime switches of the MI355X build.

`enable_tuned_gemms()` points PyTorch's TunableOp at the GEMM-algorithm table shipped with the package
(`univs_amd/tuning/*.csv`): for every (transpose, M, N, K, leading-dimension) GEMM shape of the config-2 hot
path it names the fastest hipBLASLt / rocBLAS solution found on an MI355X (tools: run bench.py once with
`UNIVS_GEMM_TUNE=1`).  Arithmetic stays fp32 on the same libraries -- only the algorithm choice changes
(+9 % frames/s at config 2).  The table carries validators (PyTorch, HIP, hipBLASLt, rocBLAS versions, GCN
arch); on any mismatch PyTorch ignores it and the default heuristics are used, as they are for shapes
that are not in the table.
"""
import glob
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))


def default_table():
    files = sorted(glob.glob(os.path.join(_HERE, "tuning", "tunableop_gfx950_*.csv")))
    return files[0] if files else None


def enable_tuned_gemms(path=None, tune=None):
    """Returns a short description of what was enabled (for logs / the bench JSON)."""
    if not torch.cuda.is_available():
        return "tunableop: no GPU"
    tune = (os.environ.get("UNIVS_GEMM_TUNE", "0") == "1") if tune is None else tune
    path = path or os.environ.get("UNIVS_GEMM_TABLE") or default_table()
    tn = torch.cuda.tunable
    if tune:
        out = os.environ.get("UNIVS_GEMM_TABLE_OUT", "tunableop_results.csv")
        tn.enable(True)
        tn.tuning_enable(True)
        tn.set_max_tuning_duration(20)
        tn.set_filename(out)
        return f"tunableop: tuning -> {out}"
    if not path or not os.path.exists(path):
        return "tunableop: no table"
    tn.enable(True)
    tn.tuning_enable(False)
    if hasattr(tn, "write_file_on_exit"):
        tn.write_file_on_exit(False)          # read-only use: no per-rank result files in the working directory
    ok = tn.read_file(path)
    if not ok:
        tn.enable(False)
        return f"tunableop: table {os.path.basename(path)} rejected by its validators (library versions differ)"
    return f"tunableop: {os.path.basename(path)}"
