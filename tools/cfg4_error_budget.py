"""BASELINE config 4 at full size on the GPU under each numerics switch: where does the mask-logit error against the
reference's CPU run (tests/golden/g14) come from?  Prints one line per variant (run on the GPU box)."""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(tag, split_linear, maskdec_impl, tuned):
    os.environ["UNIVS_SPLIT_LINEAR"] = str(split_linear)
    for m in [k for k in sys.modules if k.startswith("univs_amd.layers")]:
        importlib.reload(sys.modules[m])
    import univs_amd.layers as layers
    layers._SPLIT_LINEAR_LEVEL = split_linear
    layers._SPLIT_LINEAR = split_linear != 0
    from tests import cases      # development tool: the tests' input builders
    from tests import helpers
    from univs_amd import ops, runtime
    if tuned:
        runtime.enable_tuned_gemms()
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g14_cfg4_full_size.npz"))
    case = cases.CFG4
    swin = helpers.build_swin(dev, variant=cases.SWIN_B)
    head = helpers.build_head(case, dev, return_aux=False, **cases.CFG4_DECODER)
    x = cases.preprocess(cases.cfg2_frames()).to(dev)
    ops.mask_decode_set_impl(maskdec_impl)
    tv = [{k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in cases.cfg4_targets(case)[0].items()}]
    with torch.no_grad():
        feats = swin(x)
        out = head(feats, targets=tv)
    ops.mask_decode_set_impl(0)
    ferr = {k: float(np.abs(v[:, ::16, ::4, ::4].cpu().numpy() - g["feat_" + k + "_s"]).max()) for k, v in feats.items()}
    fmag = {k: float(np.abs(g["feat_" + k + "_s"]).max()) for k in feats}
    ref_s = g["pred_masks_s"]
    got_s = out["pred_masks"][0, :, :, ::16, ::16].cpu().numpy()
    err = np.abs(got_s - ref_s)
    q = np.unravel_index(err.argmax(), err.shape)
    print(f"{tag:34s} pred_masks max-abs-err {err.max():.3e} (|ref| there {abs(ref_s[q]):.2f}, |ref|max {np.abs(ref_s).max():.2f}), "
          f"rms {np.sqrt((err ** 2).mean()):.2e}, p99.9 {np.quantile(err, 0.999):.2e}; logits {np.abs(out['pred_logits'].cpu().numpy() - g['pred_logits']).max():.2e}; "
          f"feat err " + " ".join(f"{k}:{ferr[k]:.1e}/{fmag[k]:.0f}" for k in sorted(ferr)), flush=True)


if __name__ == "__main__":
    run("default (split linear, bf16x6 dec)", 1, 0, False)
    run("library GEMMs only", 0, 0, False)
    run("library GEMMs + exact f32 decode", 0, 1, False)
    run("split linear + exact f32 decode", 1, 1, False)
    run("default + tuned GEMM table", 1, 0, True)
