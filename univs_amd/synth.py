"""Closed-form deterministic tensors (weights and synthetic inputs), keyed on a *name*.

There is no network on either box, so no checkpoint can be fetched, and the reference's Python cannot
travel to the GPU box.  Everything that needs "the same weights on both sides" (golden generation from
the imported reference in the dev container, the oracle, the HIP build, bench.py) therefore derives
its tensors from this file: a counter-based integer hash (FNV-1a of the name -> splitmix64 per
element), evaluated in uint64/float64 numpy, so the values are bit-identical on every machine and do
not depend on any RNG implementation or torch version ("closed-form generator").

Shape/scale rules (`fill_state_dict`) are chosen so that activations stay O(1) through 12 Swin
blocks + 6 encoder layers + 10 decoder layers and mask logits come out O(1..10):
  * >=2-D weights (Linear / Conv / in_proj):     uniform, std = 1/sqrt(fan_in)
  * embeddings and task prompts (query_feat, query_embed, level_embed, prompt_*): uniform[-1, 1]
  * 1-D "*.weight" (LayerNorm / GroupNorm scales): 1 + 0.1 u
  * 1-D "*.bias":                                  0.05 u
  * `sampling_offsets.bias`:                       2.0 u   (pixels; keeps MSDA sampling local but non-trivial)
  * `cls_temp` / `reid_temp`:                      log(1/0.07), the reference's init value
      (video_mask2former_transformer_decoder_univs.py:234-236)
  * `relative_position_bias_table`:                0.5 u
  * frozen-BN `running_var`:                       1 + 0.1 u
Integer buffers (e.g. Swin's `relative_position_index`) are left untouched.
"""
import math

import numpy as np
import torch

_MASK64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK64
        z = z ^ (z >> np.uint64(31))
    return z


def hash_u01(name: str, n: int) -> np.ndarray:
    """n float64 values in [0, 1), a pure function of (name, index)."""
    seed = np.uint64(_fnv1a64(name))
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        bits = _splitmix64(seed ^ _splitmix64(idx))
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def uniform(name: str, shape, lo=-1.0, hi=1.0, dtype=torch.float32) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    v = lo + (hi - lo) * hash_u01(name, n)
    return torch.from_numpy(v.reshape(tuple(shape))).to(dtype)


def normal(name: str, shape, std=1.0, dtype=torch.float32) -> torch.Tensor:
    """Box-Muller on two hashed uniforms (float64), for inputs that should look Gaussian."""
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = np.maximum(hash_u01(name + "/u1", n), 1e-300)
    u2 = hash_u01(name + "/u2", n)
    v = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2) * std
    return torch.from_numpy(v.reshape(tuple(shape))).to(dtype)


_EMBED_KEYS = ("query_feat", "query_embed", "level_embed", "prompt_detection", "prompt_sot",
               "prompt_grounding", "absolute_pos_embed")


def make_param(name: str, shape, gain: float = 1.0) -> torch.Tensor:
    shape = tuple(shape)
    if "cls_temp" in name or "reid_temp" in name:
        return torch.full(shape, math.log(1.0 / 0.07), dtype=torch.float32)
    if "relative_position_bias_table" in name:
        return uniform(name, shape) * 0.5
    if any(k in name for k in _EMBED_KEYS):
        return uniform(name, shape)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        return uniform(name, shape) * (gain * math.sqrt(3.0) / math.sqrt(fan_in))
    if name.endswith("sampling_offsets.bias"):
        return uniform(name, shape) * 2.0
    if name.endswith("running_var"):          # frozen BatchNorm statistics must stay positive
        return 1.0 + 0.1 * uniform(name, shape)
    if name.endswith("weight"):
        return 1.0 + 0.1 * uniform(name, shape)
    return 0.05 * uniform(name, shape)


def fill_state_dict(state_dict, prefix: str = "") -> dict:
    """Return {key: tensor} with a closed-form value for every floating-point entry of `state_dict`
    (shapes taken from it).  `prefix` is prepended to the key before hashing so that the same module
    class embedded at different places (or tested stand-alone) can be given the values it would have
    inside the full model."""
    out = {}
    for k, v in state_dict.items():
        if torch.is_floating_point(v):
            out[k] = make_param(prefix + k, v.shape).to(v.dtype)
        else:
            out[k] = v.clone()
    return out


def load_synthetic(module: torch.nn.Module, prefix: str = "") -> torch.nn.Module:
    sd = fill_state_dict(module.state_dict(), prefix)
    module.load_state_dict(sd, strict=True)
    return module


def synthetic_frames(T: int, H: int, W: int, seed_name: str = "frames/seed0") -> torch.Tensor:
    """uint8-valued float frames U{0..255}, [T, 3, H, W]."""
    v = np.floor(hash_u01(seed_name, T * 3 * H * W) * 256.0)
    return torch.from_numpy(v.reshape(T, 3, H, W)).to(torch.float32)


PIXEL_MEAN = (123.675, 116.28, 103.53)   # configs/univs_inf/vids/Base.yaml:6
PIXEL_STD = (58.395, 57.12, 57.375)      # configs/univs_inf/vids/Base.yaml:7
