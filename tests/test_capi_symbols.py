"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/univs_hip.h declares
(no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import pytest

from univs_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "univs_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(univs_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    build.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 8
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/univs_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes SIGNATURES table out of sync with the header"


def test_version_and_error_paths_without_gpu():
    lib = _lib.load()
    assert lib.univs_version().decode().startswith("univs_hip ")
    assert lib.univs_msda_set_impl(7) == _lib.ERR_INVALID_ARGUMENT
    assert b"impl=7" in lib.univs_last_error()
    assert lib.univs_msda_set_impl(0) == _lib.OK
    # backward validates its arguments without touching the device (no levels -> invalid argument)
    rc = lib.univs_msda_backward_f32(None, None, None, None, None, None, 0, 0, 0, 0, 0, 0, 0, None, None, None, None)
    assert rc == _lib.ERR_INVALID_ARGUMENT


def test_product_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from univs_amd import ops
    v = torch.zeros(1, 4, 1, 4)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        ops.ms_deform_attn_forward(v, [(2, 2)], [0], torch.zeros(1, 4, 1, 1, 1, 2), torch.zeros(1, 4, 1, 1, 1))
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        ops.mask_decode(torch.zeros(1, 2, 4), torch.zeros(1, 4, 2, 2))


def test_compat_module_exposes_reference_entry_points():
    """B2: the module name and the two functions the reference's autograd Function calls
    (ops/functions/ms_deform_attn_func.py:36,45)."""
    import importlib
    import os
    import sys
    sys.path.insert(0, os.path.join(ROOT, "univs_amd", "compat"))
    try:
        m = importlib.import_module("MultiScaleDeformableAttention")
    finally:
        sys.path.pop(0)
    assert callable(m.ms_deform_attn_forward) and callable(m.ms_deform_attn_backward)


def test_config_surface_loads_reference_style_yaml(tmp_path):
    from univs_amd.config import load_cfg
    base = tmp_path / "Base.yaml"
    base.write_text("MODEL:\n  BACKBONE:\n    NAME: build_resnet_backbone\n  MASK_FORMER:\n    NUM_OBJECT_QUERIES: 200\n"
                    "INPUT:\n  CROP:\n    SIZE: (600, 1024)\n")
    child = tmp_path / "swint.yaml"
    child.write_text("_BASE_: Base.yaml\nMODEL:\n  BACKBONE:\n    NAME: D2SwinTransformer\n  SWIN:\n    EMBED_DIM: 96\n"
                     "INPUT:\n  SAMPLING_FRAME_NUM: 4\n")
    cfg = load_cfg(str(child), ["MODEL.MASK_FORMER.NUM_OBJECT_QUERIES", "100", "INPUT.MIN_SIZE_TEST", "720"])
    assert cfg.MODEL.BACKBONE.NAME == "D2SwinTransformer"
    assert cfg.MODEL.MASK_FORMER.NUM_OBJECT_QUERIES == 100 and cfg.INPUT.MIN_SIZE_TEST == 720
    assert cfg.INPUT.CROP.SIZE == (600, 1024) and cfg.INPUT.SAMPLING_FRAME_NUM == 4
    assert cfg.MODEL.SEM_SEG_HEAD.PIXEL_DECODER_NAME == "MSDeformAttnPixelDecoder"   # default kept


def test_configure_round_trip_without_a_gpu():
    """UnivsConfig (include/univs_hip.h): the settings object is plain host state -- set by name, read back, validated."""
    from univs_amd import ops
    base = ops.get_config()
    try:
        prev = ops.configure(msda_impl=1, msda_halo=5)
        assert prev == base and ops.get_config()["msda_impl"] == 1 and ops.get_config()["msda_halo"] == 5
        with pytest.raises(RuntimeError):
            ops.configure(mask_decode_ct=3)
        with pytest.raises(KeyError):
            ops.configure(nope=1)
        assert ops.get_config()["msda_impl"] == 1
    finally:
        ops.configure()
    assert all(v == 0 for v in ops.get_config().values())
