"""Closed-form test inputs shared by oracle/gen_golden.py (which feeds them to the real reference) and
by the tests (which feed the very same tensors to the oracle and to the HIP path)."""
import math

import numpy as np
import torch

from univs_amd import synth

# ---------------------------------------------------------------------------------------------------
# MSDeformAttn forward
# ---------------------------------------------------------------------------------------------------
MSDA_CASES = [
    # config-1 encoder geometry (256x448 input -> res5/res4/res3)
    dict(name="cfg1", shapes=[(8, 14), (16, 28), (32, 56)], N=2, M=8, D=32, P=4, encoder=True),
    # non-2x pyramid with odd sizes (ragged tiles, clipped windows)
    dict(name="ragged", shapes=[(5, 7), (9, 13), (17, 25)], N=1, M=8, D=32, P=4, encoder=True),
    # four levels, finest level first (tile grid must follow the largest level wherever it sits)
    dict(name="L4", shapes=[(32, 48), (16, 24), (8, 12), (4, 6)], N=1, M=8, D=32, P=4, encoder=True),
    # single level, single head
    dict(name="L1", shapes=[(20, 33)], N=2, M=1, D=32, P=4, encoder=True),
    # decoder-style: Lq != S, runtime L/P path of the vec4 kernel
    dict(name="generic", shapes=[(6, 4), (3, 2)], N=2, M=4, D=16, P=3, encoder=False, Lq=10),
    # head dim not a multiple of 4 -> scalar kernel
    dict(name="oddD", shapes=[(7, 5), (4, 3)], N=1, M=3, D=5, P=2, encoder=False, Lq=9),
]


def level_start_index(shapes):
    st, acc = [], 0
    for h, w in shapes:
        st.append(acc)
        acc += h * w
    return st, acc


def msda_inputs(case, dtype=torch.float32):
    """-> value [N,S,M,D], shapes list, level_start list, loc [N,Lq,M,L,P,2], attn [N,Lq,M,L,P]."""
    shapes = case["shapes"]
    N, M, D, P = case["N"], case["M"], case["D"], case["P"]
    L = len(shapes)
    lsi, S = level_start_index(shapes)
    nm = "msda/" + case["name"]
    value = synth.normal(nm + "/value", (N, S, M, D))
    if case["encoder"]:
        Lq = S
        # reference points = pixel centres of every level (msdeformattn.py:143-158, valid_ratio == 1)
        refs = []
        for (h, w) in shapes:
            ys = (torch.arange(h, dtype=torch.float32) + 0.5) / h
            xs = (torch.arange(w, dtype=torch.float32) + 0.5) / w
            yy, xx = torch.meshgrid(ys, xs, indexing="ij")
            refs.append(torch.stack([xx.reshape(-1), yy.reshape(-1)], -1))
        ref = torch.cat(refs, 0)  # [S, 2]
        off = synth.normal(nm + "/off", (N, Lq, M, L, P, 2), std=2.0)  # pixels of the target level
        # every 7th query gets far offsets (out of the tile halo, partly out of the image)
        if case.get("far", True):
            far = (torch.arange(Lq) % 7 == 3).view(1, Lq, 1, 1, 1, 1)
            off = torch.where(far, off * 6.0, off)
        norm = torch.tensor([[w, h] for (h, w) in shapes], dtype=torch.float32).view(1, 1, 1, L, 1, 2)
        loc = ref.view(1, Lq, 1, 1, 1, 2) + off / norm
        if not case.get("far", True):
            loc = loc.clamp(-0.05, 1.05)  # SURVEY.md section 8d "realistic locality" micro-benchmark input
    else:
        Lq = case["Lq"]
        loc = synth.uniform(nm + "/loc", (N, Lq, M, L, P, 2), -0.15, 1.15)
    logits = synth.normal(nm + "/attn", (N, Lq, M, L * P))
    attn = torch.softmax(logits, -1).view(N, Lq, M, L, P)
    return value.to(dtype), shapes, lsi, loc.to(dtype).contiguous(), attn.to(dtype).contiguous()


def msda_query_subset(case, Lq):
    return np.arange(0, Lq, 3)


# ---------------------------------------------------------------------------------------------------
# Swin window attention (module level: x -> qkv -> core -> proj)
# ---------------------------------------------------------------------------------------------------
WINATTN_CASES = [
    dict(name="winattn/w7_noshift", dim=96, heads=3, win=7, nW=6, batch=2, shift=False),
    dict(name="winattn/w7_shift", dim=96, heads=3, win=7, nW=6, batch=2, shift=True),
    dict(name="winattn/w12_shift", dim=128, heads=4, win=12, nW=4, batch=1, shift=True),
]


def winattn_inputs(case):
    """-> x [batch*nW, win*win, dim], mask [nW, Ntok, Ntok] or None (0 / -100, swin.py:437-440)."""
    ntok = case["win"] ** 2
    x = synth.normal(case["name"] + "/x", (case["batch"] * case["nW"], ntok, case["dim"]))
    mask = None
    if case["shift"]:
        # a plausible shifted-window mask: tokens carry a region id, different ids cannot attend
        ids = torch.floor(synth.uniform(case["name"] + "/ids", (case["nW"], ntok), 0.0, 3.0))
        ids[0] = 0.0  # first window: single region (all-zero mask row block)
        diff = ids[:, None, :] - ids[:, :, None]
        mask = torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))
    return x, mask


# ---------------------------------------------------------------------------------------------------
# Mask decode
# ---------------------------------------------------------------------------------------------------
MASKDEC_CASES = [
    dict(name="maskdec/cfg1", T=2, Q=20, C=256, H=64, W=112),
    dict(name="maskdec/ragged", T=3, Q=37, C=256, H=23, W=41),     # HW not a multiple of 32, Q not of 32
    dict(name="maskdec/bigQ", T=1, Q=150, C=256, H=16, W=28),      # more than one 128-row tile
    dict(name="maskdec/smallC", T=2, Q=5, C=64, H=8, W=9),
]


def maskdec_inputs(case):
    e = synth.normal(case["name"] + "/embed", (case["T"], case["Q"], case["C"]), std=0.5)
    f = synth.normal(case["name"] + "/feat", (case["T"], case["C"], case["H"], case["W"]), std=0.5)
    return e, f
