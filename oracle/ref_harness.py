"""Dev-container-only harness that imports the *reference* UniVS hot-path modules.

TEST INFRASTRUCTURE -- never imported by the product (`univs_amd/`), never shipped to the GPU box
as a dependency: `/root/reference` does not exist there.  It exists to (i) generate the golden
fixtures under `tests/golden/` (see `oracle/gen_golden.py`) and (ii) validate the CPU restatement
in `oracle/torch_ref.py` against the real reference while developing.

How the import works:
  * the reference's package `__init__` files import cv2 / kornia / pycocotools / torchvision, which are
    absent here, so we register empty *parent packages* (with `__path__` pointing at the reference
    directories) and then import only the leaf modules of the hot path;
  * the handful of third-party symbols those leaf modules need at import / construction time
    (detectron2 `configurable`, `Conv2d(norm=, activation=)`, `get_norm`, `ShapeSpec`, registries,
    `point_sample`; fvcore `c2_xavier_fill`; timm `DropPath/to_2tuple/trunc_normal_`) are provided
    as small behavioural stand-ins written from the public documentation of those libraries;
  * the compiled extension `MultiScaleDeformableAttention` is replaced by a module forwarding to the
    reference's own `ms_deform_attn_core_pytorch` (ops/functions/ms_deform_attn_func.py:52-72), the
    swap the reference itself documents at ops/modules/ms_deform_attn.py:118-119 and uses as the
    oracle in ops/test.py:35-63.
"""
import importlib
import math
import os
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REF_ROOT = os.environ.get("UNIVS_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "univs"))


def _pkg(name, path=None):
    m = types.ModuleType(name)
    m.__path__ = [path] if path else []
    m.__package__ = name
    sys.modules[name] = m
    return m


class _Registry:
    def __init__(self, name):
        self._name = name
        self._map = {}

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self._map[o.__name__] = o
                return o
            return deco
        self._map[obj.__name__] = obj
        return obj

    def get(self, name):
        return self._map[name]


def _configurable(init_func=None, *, from_config=None):
    # explicit-kwargs construction only (the oracle never builds from a cfg)
    if init_func is not None:
        return init_func

    def wrapper(f):
        return f
    return wrapper


class _ShapeSpec:
    def __init__(self, channels=None, height=None, width=None, stride=None):
        self.channels, self.height, self.width, self.stride = channels, height, width, stride


class _Conv2d(nn.Conv2d):
    """detectron2.layers.Conv2d: conv -> optional norm -> optional activation."""

    def __init__(self, *args, **kwargs):
        norm = kwargs.pop("norm", None)
        activation = kwargs.pop("activation", None)
        super().__init__(*args, **kwargs)
        self.norm = norm
        self.activation = activation

    def forward(self, x):
        x = F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        return x


def _get_norm(norm, out_channels):
    if norm is None or norm == "":
        return None
    if norm == "GN":
        return nn.GroupNorm(32, out_channels)
    if norm == "LN":
        return nn.LayerNorm(out_channels)
    raise ValueError(norm)


def _point_sample(input, point_coords, **kwargs):
    add_dim = False
    if point_coords.dim() == 3:
        add_dim = True
        point_coords = point_coords.unsqueeze(2)
    output = F.grid_sample(input, 2.0 * point_coords - 1.0, **kwargs)
    if add_dim:
        output = output.squeeze(3)
    return output


def _c2_xavier_fill(module):
    nn.init.kaiming_uniform_(module.weight, a=1)
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)


class _DropPath(nn.Module):
    def __init__(self, p=0.0):
        super().__init__()
        self.p = p

    def forward(self, x):
        return x  # eval only


def _to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


def _trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0):
    return nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)


_INSTALLED = False


def install():
    """Register parent packages + third-party stand-ins, then make reference leaf modules importable."""
    global _INSTALLED
    if _INSTALLED:
        return
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REF_ROOT} (this harness only works in the dev container)")

    R = REF_ROOT
    # --- third-party stand-ins -------------------------------------------------------------
    d2 = _pkg("detectron2")
    cfgm = _pkg("detectron2.config"); cfgm.configurable = _configurable
    cfgm.CfgNode = dict
    lay = _pkg("detectron2.layers")
    lay.Conv2d, lay.ShapeSpec, lay.get_norm = _Conv2d, _ShapeSpec, _get_norm
    lay.DeformConv = object
    mod = _pkg("detectron2.modeling")
    mod.BACKBONE_REGISTRY = _Registry("BACKBONE")
    mod.SEM_SEG_HEADS_REGISTRY = _Registry("SEM_SEG_HEADS")
    mod.META_ARCH_REGISTRY = _Registry("META_ARCH")
    mod.Backbone = nn.Module
    mod.ShapeSpec = _ShapeSpec
    ut = _pkg("detectron2.utils")
    reg = _pkg("detectron2.utils.registry"); reg.Registry = _Registry
    _pkg("detectron2.projects")
    _pkg("detectron2.projects.point_rend")
    pf = _pkg("detectron2.projects.point_rend.point_features"); pf.point_sample = _point_sample
    d2.config, d2.layers, d2.modeling, d2.utils = cfgm, lay, mod, ut

    fv = _pkg("fvcore"); fvnn = _pkg("fvcore.nn"); wi = _pkg("fvcore.nn.weight_init")
    wi.c2_xavier_fill = _c2_xavier_fill
    wi.c2_msra_fill = _c2_xavier_fill
    fv.nn = fvnn; fvnn.weight_init = wi

    tm = _pkg("timm"); tmm = _pkg("timm.models"); tml = _pkg("timm.models.layers")
    tml.DropPath, tml.to_2tuple, tml.trunc_normal_ = _DropPath, _to_2tuple, _trunc_normal_
    tm.models = tmm; tmm.layers = tml

    _pkg("torchvision")

    # --- reference parent packages (no __init__ executed) -----------------------------------
    sys.path.insert(0, R)
    _pkg("mask2former", f"{R}/mask2former")
    _pkg("mask2former.modeling", f"{R}/mask2former/modeling")
    _pkg("mask2former.modeling.backbone", f"{R}/mask2former/modeling/backbone")
    _pkg("mask2former.modeling.meta_arch", f"{R}/mask2former/modeling/meta_arch")
    _pkg("mask2former.modeling.pixel_decoder", f"{R}/mask2former/modeling/pixel_decoder")
    _pkg("mask2former.modeling.transformer_decoder", f"{R}/mask2former/modeling/transformer_decoder")
    _pkg("univs", f"{R}/univs")
    _pkg("univs.modeling", f"{R}/univs/modeling")
    _pkg("univs.modeling.transformer_decoder", f"{R}/univs/modeling/transformer_decoder")
    _pkg("univs.modeling.prompt_encoder", f"{R}/univs/modeling/prompt_encoder")
    _pkg("univs.utils", f"{R}/univs/utils")
    langm = _pkg("univs.modeling.language")
    langm.pre_tokenize_expression = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("no tokenizer in oracle"))
    _pkg("datasets", f"{R}/datasets")
    _pkg("datasets.concept_emb", f"{R}/datasets/concept_emb")

    # --- the compiled op -> the reference's own pure-PyTorch core ---------------------------
    msda = types.ModuleType("MultiScaleDeformableAttention")
    sys.modules["MultiScaleDeformableAttention"] = msda

    def _fwd(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
        core = sys.modules["mask2former.modeling.pixel_decoder.ops.functions.ms_deform_attn_func"].ms_deform_attn_core_pytorch
        return core(value, spatial_shapes, sampling_loc, attn_weight)

    def _bwd(*a, **k):
        raise NotImplementedError
    msda.ms_deform_attn_forward, msda.ms_deform_attn_backward = _fwd, _bwd

    # prompt_encoder's __init__ re-exports; make `from univs.modeling.prompt_encoder import X` work
    pe_leaf = importlib.import_module("univs.modeling.prompt_encoder.prompt_encoder")
    pe_pkg = sys.modules["univs.modeling.prompt_encoder"]
    for n in ("TextPromptEncoder", "VisualPromptEncoder", "VisualPromptSampler"):
        setattr(pe_pkg, n, getattr(pe_leaf, n))
    _INSTALLED = True


def ref():
    """Namespace of the reference classes/functions on the hot path."""
    install()
    ns = types.SimpleNamespace()
    f = importlib.import_module("mask2former.modeling.pixel_decoder.ops.functions.ms_deform_attn_func")
    ns.ms_deform_attn_core_pytorch = f.ms_deform_attn_core_pytorch
    m = importlib.import_module("mask2former.modeling.pixel_decoder.ops.modules.ms_deform_attn")
    ns.MSDeformAttn = m.MSDeformAttn
    pd = importlib.import_module("mask2former.modeling.pixel_decoder.msdeformattn")
    ns.MSDeformAttnPixelDecoder = pd.MSDeformAttnPixelDecoder
    ns.MSDeformAttnTransformerEncoderLayer = pd.MSDeformAttnTransformerEncoderLayer
    sw = importlib.import_module("mask2former.modeling.backbone.swin")
    ns.SwinTransformer = sw.SwinTransformer
    ns.WindowAttention = sw.WindowAttention
    pe2 = importlib.import_module("mask2former.modeling.transformer_decoder.position_encoding")
    ns.PositionEmbeddingSine = pe2.PositionEmbeddingSine
    pe3 = importlib.import_module("univs.modeling.transformer_decoder.position_encoding")
    ns.PositionEmbeddingSine3D = pe3.PositionEmbeddingSine3D
    ns.PositionEmbeddingSine3DArbitraryT = pe3.PositionEmbeddingSine3DArbitraryT
    dec = importlib.import_module("univs.modeling.transformer_decoder.video_mask2former_transformer_decoder_univs")
    ns.Decoder = dec.VideoMultiScaleMaskedTransformerDecoderUniVS
    pr = importlib.import_module("univs.modeling.prompt_encoder.prompt_encoder")
    ns.VisualPromptSampler = pr.VisualPromptSampler
    ns.VisualPromptEncoder = pr.VisualPromptEncoder
    hd = importlib.import_module("mask2former.modeling.meta_arch.mask_former_head")
    ns.MaskFormerHead = hd.MaskFormerHead
    cm = importlib.import_module("univs.utils.comm")
    ns.comm = cm
    ns.ShapeSpec = _ShapeSpec
    return ns


# ---------------------------------------------------------------------------------------------
# The reference's clip loop (univs/inference/inference_video_entity.py) -- dev container only.
# Its module imports cv2 / kornia / pycocotools / matplotlib / torchvision.ops / detectron2 data
# structures and the whole `univs` package; none of that is exercised by `inference_video` up to the
# `targets` updates we pin (tests/golden/g11_*), so those names are registered as inert stand-ins.
def _install_inference_stubs():
    if "univs.inference.inference_video_entity" in sys.modules:
        return
    R = REF_ROOT

    class _Inert:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            raise RuntimeError("inert stand-in called")

    for name in ("cv2", "kornia", "pycocotools", "matplotlib"):
        if name not in sys.modules:
            _pkg(name)
    sys.modules["kornia"].color = types.SimpleNamespace()
    pm = _pkg("pycocotools.mask"); sys.modules["pycocotools"].mask = pm
    pp = _pkg("matplotlib.pyplot"); sys.modules["matplotlib"].pyplot = pp
    tv = sys.modules["torchvision"]
    tvo = _pkg("torchvision.ops"); tvb = _pkg("torchvision.ops.boxes")
    tvb.batched_nms = _Inert()
    tvb.box_area = lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    tv.ops = tvo; tvo.boxes = tvb

    d2data = _pkg("detectron2.data")

    class _Meta:
        @staticmethod
        def get(name):
            return types.SimpleNamespace(name=name, thing_dataset_id_to_contiguous_id={})
    d2data.MetadataCatalog = _Meta
    pp2 = _pkg("detectron2.modeling.postprocessing"); pp2.sem_seg_postprocess = _Inert()
    st = _pkg("detectron2.structures")
    st.Boxes = st.ImageList = st.Instances = st.BitMasks = _Inert
    mem = _pkg("detectron2.utils.memory"); mem.retry_if_cuda_oom = lambda f: f

    _pkg("mask2former.utils", f"{R}/mask2former/utils")
    u = sys.modules["univs"]
    for n in ("VideoSetCriterionUni", "VideoHungarianMatcherUni", "BoxVISTeacherSetPseudoMask", "TextPromptEncoder",
              "build_clip_language_encoder", "Clips", "FastOverTracker_DET", "MDQE_OverTrackerEfficient"):
        setattr(u, n, _Inert)
    _pkg("univs.data")
    dd = _pkg("univs.data.datasets")
    dd._get_vspw_vss_metadata = dd._get_vipseg_panoptic_metadata_val = lambda *a, **k: {}
    pt = _pkg("univs.prepare_targets"); pt.PrepareTargets = _Inert
    vz = _pkg("univs.utils.visualizer"); vz.VisualizerFrame = _Inert
    _pkg("univs.inference", f"{R}/univs/inference")
    vq = _pkg("univs.inference.visualization"); vq.visualization_query_embds = _Inert


def ref_inference():
    """The reference's `InferenceVideoEntity` class and its helper module (`univs/inference/comm.py`)."""
    install()
    _install_inference_stubs()
    ns = types.SimpleNamespace()
    ns.comm = importlib.import_module("univs.inference.comm")
    m = importlib.import_module("univs.inference.inference_video_entity")
    ns.InferenceVideoEntity = m.InferenceVideoEntity
    ns.module = m
    return ns


def ref_inference_vos():
    """The reference's `InferenceVideoVOS` (univs/inference/inference_video_vos.py); same inert stand-ins as for the
    entity loop (PIL is real: the reference writes its results as PNG files, which the golden generator reads back)."""
    install()
    _install_inference_stubs()
    ns = types.SimpleNamespace()
    m = importlib.import_module("univs.inference.inference_video_vos")
    ns.InferenceVideoVOS = m.InferenceVideoVOS
    ns.module = m
    return ns


# ---------------------------------------------------------------------------------------------
# The reference's CLIP text encoder + tokenizer (univs/modeling/language/) -- dev container only.
def ref_language():
    """Namespace with the reference's `TextEncoder` module, `clip_prompt_utils` module and `TextPromptEncoder`.
    `ftfy` is not installed here: it is replaced by an identity `fix_text` (the expressions used are plain ASCII)."""
    import importlib.util
    install()
    if "ftfy" not in sys.modules:
        f = types.ModuleType("ftfy")
        f.fix_text = lambda t: t
        sys.modules["ftfy"] = f
    ns = types.SimpleNamespace()
    for attr, fname in (("TextEncoder", "TextEncoder.py"), ("clip_prompt_utils", "clip_prompt_utils.py")):
        name = f"_ref_language_{attr}"
        if name not in sys.modules:
            spec = importlib.util.spec_from_file_location(name, f"{REF_ROOT}/univs/modeling/language/{fname}")
            m = importlib.util.module_from_spec(spec)
            sys.modules[name] = m
            spec.loader.exec_module(m)
        setattr(ns, attr, sys.modules[name])
    pe_leaf = importlib.import_module("univs.modeling.prompt_encoder.prompt_encoder")
    pe_leaf.pre_tokenize_expression = ns.clip_prompt_utils.pre_tokenize_expression
    ns.TextPromptEncoder = pe_leaf.TextPromptEncoder
    ns.bpe_path = f"{REF_ROOT}/univs/modeling/language/bpe_simple_vocab_16e6.txt.gz"
    return ns


def ref_prepare_targets():
    """The reference's `PrepareTargets` class (univs/prepare_targets.py), loaded from its file under a private module name
    (the `univs.prepare_targets` entry of sys.modules is an inert stand-in for the inference-loop imports)."""
    import importlib.util
    install()
    _install_inference_stubs()
    name = "_ref_prepare_targets"
    if name not in sys.modules:
        spec = importlib.util.spec_from_file_location(name, f"{REF_ROOT}/univs/prepare_targets.py")
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
    return sys.modules[name].PrepareTargets


def ref_evaluators():
    """The reference's `VPSEvaluator` / `VSSEvaluator` classes (univs/evaluation/vps_evaluation.py, vss_evaluation.py) for their
    `process` methods -- the on-disk result formats.  Stand-ins: detectron2's DatasetEvaluator / PathManager / MetadataCatalog /
    comm (inert bases), tqdm, the metric modules the files import at the top (never called here), and `panopticapi.utils`, which is
    absent from this image: `IdGenerator` / `rgb2id` are the restatement in univs_amd/inference/results.py (published algorithm) --
    so the golden pins everything `process` does EXCEPT the library's own colour rule."""
    import importlib.util
    install()
    from univs_amd.inference import results as ours
    pa = _pkg("panopticapi")
    pu = _pkg("panopticapi.utils")
    pu.IdGenerator, pu.rgb2id = ours.IdGenerator, ours.rgb2id
    pa.utils = pu
    for name in ("detectron2.utils.comm", "detectron2.config", "detectron2.data", "detectron2.evaluation", "detectron2.utils.file_io"):
        if name not in sys.modules:
            _pkg(name)
    sys.modules["detectron2.config"].CfgNode = getattr(sys.modules["detectron2.config"], "CfgNode", type("CfgNode", (), {}))
    if not hasattr(sys.modules["detectron2.data"], "MetadataCatalog"):
        sys.modules["detectron2.data"].MetadataCatalog = types.SimpleNamespace(get=lambda n: types.SimpleNamespace(name=n))
    sys.modules["detectron2.evaluation"].DatasetEvaluator = type("DatasetEvaluator", (), {})

    class _PM:
        @staticmethod
        def get_local_path(p):
            return p

        @staticmethod
        def mkdirs(p):
            os.makedirs(p, exist_ok=True)
    sys.modules["detectron2.utils.file_io"].PathManager = _PM
    if "tqdm" not in sys.modules:
        try:
            import tqdm  # noqa: F401
        except ImportError:
            t = _pkg("tqdm")
            t.tqdm = lambda x, **k: x
    ev = _pkg("_ref_evaluation", f"{REF_ROOT}/univs/evaluation")
    for leaf, names in (("eval_vpq_vps", ("vpq_compute_parallel",)), ("eval_stquality_vps", ("STQuality",)), ("eval_utils_vss", ("Evaluator",))):
        m = _pkg(f"_ref_evaluation.{leaf}")
        for n in names:
            setattr(m, n, None)
        setattr(ev, leaf, m)
    ns = types.SimpleNamespace()
    for attr, fname in (("VPSEvaluator", "vps_evaluation"), ("VSSEvaluator", "vss_evaluation")):
        name = f"_ref_evaluation.{fname}"
        spec = importlib.util.spec_from_file_location(name, f"{REF_ROOT}/univs/evaluation/{fname}.py")
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        setattr(ns, attr, getattr(m, attr))
    return ns
