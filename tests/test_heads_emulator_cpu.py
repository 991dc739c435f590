"""The data flow of csrc/msda_heads.hip (generation 6) emulated on the host (tools/heads_emulate.cpp, built with hipcc's host
compiler over the SAME geometry header the kernel includes, csrc/msda_heads_geom.h): tile tables, cold / entering-row piece lists
with their packed masks, the circular row windows in LDS, the sample record with its lane-specific corner / chunk order, the FMA
division, the segment schedules of both policies -- against an fp64 evaluation of ms_deform_attn_forward
(ops/src/cuda/ms_deform_im2col_cuda.cuh:242-304) on ten geometries incl. slices of BASELINE config 2 and 5.  No GPU: this is the
part of the MSDA kernel that is host logic."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    return None


@pytest.mark.skipif(_hipcc() is None, reason="hipcc not found")
def test_heads_data_flow_and_schedules_on_the_host(tmp_path):
    exe = str(tmp_path / "heads_emulate")
    subprocess.run([_hipcc(), "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tools", "heads_emulate.cpp"), "-o", exe], check=True, capture_output=True, timeout=300)
    out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=300).stdout
    rows = [l for l in out.splitlines() if "max err" in l]
    assert len(rows) >= 10 and out.strip().endswith("all ok"), out
    for l in rows:
        m = re.search(r"max err ([0-9.e+-]+), bank conflicts (\d+), stale reads (\d+), unwritten outputs (\d+), tiles not covered once (\d+), "
                      r"inexact divisions (\d+)", l)
        assert m, l
        assert float(m.group(1)) < 2e-5 and all(int(m.group(i)) == 0 for i in range(2, 7)), l
    assert any(l.startswith("cfg2-slice") for l in rows) and any(l.startswith("cfg5-slice") for l in rows)
    sched = [l for l in out.splitlines() if l.startswith("schedule")]
    assert len(sched) >= 10
    for l in sched:      # every schedule covers every tile once and is balanced to one tile
        m = re.search(r": ok, steps total (\d+) \(want (\d+)\), max per workgroup (\d+) \(ideal ([0-9.]+)\)", l)
        assert m and m.group(1) == m.group(2) and int(m.group(3)) <= float(m.group(4)) + 1.0, l
