"""Generate the golden fixtures under tests/golden/ by running the REAL reference (imported from
/root/reference through oracle/ref_harness.py).  Dev-container only; the fixtures it writes are plain
data (inputs where they are not closed-form, and the reference's outputs) and travel with the repo.

    python oracle/gen_golden.py [name ...]        # no names = all

Inputs and weights come from univs_amd/synth.py (closed-form, name-keyed), so most fixtures only store
the reference's OUTPUTS; tests rebuild the inputs from the same names.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_harness as rh  # noqa: E402
from univs_amd import synth  # noqa: E402
from tests import cases  # noqa: E402  (shared input builders: the tests use the very same functions)

OUT = os.path.join(ROOT, "tests", "golden")
GENERATORS = {}


def gen(fn):
    GENERATORS[fn.__name__] = fn
    return fn


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


# ---------------------------------------------------------------------------------------------------
@gen
def g0_msda_kat():
    """The reference's own known-answer shapes (ops/test.py:24-31,35-63): N1 M2 D2 Lq2 L2 P2,
    shapes (6,4),(3,2), torch.manual_seed(3).  Inputs are stored (torch RNG is version dependent)."""
    R = rh.ref()
    N, M, D, Lq, L, P = 1, 2, 2, 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    torch.manual_seed(3)
    out = {}
    for tag in ("double", "float"):  # same call order as ops/test.py
        value = torch.rand(N, S, M, D) * 0.01
        loc = torch.rand(N, Lq, M, L, P, 2)
        attn = torch.rand(N, Lq, M, L, P) + 1e-5
        attn /= attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
        if tag == "double":
            o = R.ms_deform_attn_core_pytorch(value.double(), shapes, loc.double(), attn.double())
        else:
            o = R.ms_deform_attn_core_pytorch(value, shapes, loc, attn)
        out.update({f"value_{tag}": value, f"loc_{tag}": loc, f"attn_{tag}": attn, f"out_{tag}": o})
    save("g0_msda_kat", shapes=shapes, level_start_index=lsi, **out)


@gen
def g1_msda_encoder_geometry():
    """MSDA at the config-1 encoder geometry incl. out-of-range locations;
    also a ragged / non-2x pyramid and an L=4 case.  Inputs closed-form (tests/cases.py)."""
    R = rh.ref()
    arrays = {}
    for case in cases.MSDA_CASES:
        value, shapes, lsi, loc, attn = cases.msda_inputs(case)
        o = R.ms_deform_attn_core_pytorch(value, torch.as_tensor(shapes), loc, attn)
        sub = cases.msda_query_subset(case, o.shape[1])
        arrays[f"{case['name']}/out_subset"] = o[:, sub]
        arrays[f"{case['name']}/sum"] = o.double().sum()
        arrays[f"{case['name']}/abs_sum"] = o.double().abs().sum()
    save("g1_msda_geometry", **arrays)


@gen
def g_window_attention():
    """Reference WindowAttention.forward (swin.py:131-171) with closed-form weights, with and without
    the shifted-window mask, window 7 (49 tokens) and window 12 (144 tokens)."""
    R = rh.ref()
    arrays = {}
    for case in cases.WINATTN_CASES:
        mod = R.WindowAttention(case["dim"], (case["win"], case["win"]), case["heads"]).eval()
        synth.load_synthetic(mod, prefix=case["name"] + ".")
        x, mask = cases.winattn_inputs(case)
        with torch.no_grad():
            y = mod(x, mask=mask)
        arrays[f"{case['name']}/out"] = y
    save("g_window_attention", **arrays)


def main():
    names = sys.argv[1:] or list(GENERATORS)
    for n in names:
        print(f"== {n}")
        with torch.no_grad():
            GENERATORS[n]()


if __name__ == "__main__":
    main()
