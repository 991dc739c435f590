"""Drop-in for the reference's compiled extension module `MultiScaleDeformableAttention`
(pybind registration at mask2former/modeling/pixel_decoder/ops/src/vision.cpp:18-21; imported by
ops/functions/ms_deform_attn_func.py:21-29).  Put this directory on PYTHONPATH instead of building the
reference's CUDA extension; the calls land in libunivs_hip.so through univs_amd.ops."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from univs_amd.ops import ms_deform_attn_backward, ms_deform_attn_forward  # noqa: E402,F401
