"""CPU: caller-side target preparation (univs_amd/prepare_targets.py) against the reference's `PrepareTargets.process_inference`
(oracle/gen_golden.py: g18) -- detection with a known / raw / semantic dataset, grounding (expressions through the CLIP text
tower + tokenizer), grounding without expressions, 'sot', custom text prompts."""
import json
import os

import numpy as np
import pytest
import torch

from oracle.cpu_path import cpu_ops
from tests import cases
from tests.test_language_cpu import BPE, needs_vocab
from univs_amd.modeling.prompt_encoder import TextPromptEncoder
from univs_amd.prepare_targets import PrepareTargets


def describe(out):
    plain, tensors = {}, {}
    for k, v in out.items():
        (tensors if isinstance(v, torch.Tensor) else plain)[k] = v
    return json.loads(json.dumps(plain, sort_keys=True, default=list)), tensors


@needs_vocab
@pytest.mark.parametrize("name", list(cases.prepare_targets_inputs()))
def test_process_inference_matches_reference(golden_dir, name, monkeypatch):
    monkeypatch.setenv("UNIVS_BPE_VOCAB", BPE)
    from univs_amd.modeling.language import tokenizer as tk
    tk._shared_tokenizer.cache_clear()
    g = np.load(os.path.join(golden_dir, "g18_prepare_targets.npz"))
    over, batched = cases.prepare_targets_inputs()[name]
    tpe = TextPromptEncoder(cases.build_text_encoder(cases.TEXT_SMALL), num_frames=3, device="cpu")
    pt = PrepareTargets(num_frames=3, clip_class_embed_path=cases.clip_table(), **over)
    with cpu_ops(), torch.no_grad():
        out = pt.process_inference(batched, (64, 96), torch.device("cpu"), tpe, (60, 90))
    assert len(out) == 1
    plain, tensors = describe(out[0])
    assert plain == json.loads(str(g[f"{name}_0_plain"]))
    assert batched[0]["prompt_type"] == str(g[f"{name}_input_prompt_type"])
    want_tensors = {k[len(name) + 3:] for k in g.files if k.startswith(name + "_0_") and not k.endswith("_plain")}
    for k, v in tensors.items():
        if f"{name}_0_{k}_shape" in g.files:                    # large table: shape, sample and checksum
            assert list(v.shape) == g[f"{name}_0_{k}_shape"].tolist()
            assert np.array_equal(v[::97, ::16].numpy(), g[f"{name}_0_{k}_sample"])
            assert abs(float(v.double().sum()) - float(g[f"{name}_0_{k}_sum"])) < 1e-6
            want_tensors -= {k + "_shape", k + "_sample", k + "_sum"}
        else:
            ref = torch.from_numpy(g[f"{name}_0_{k}"])
            assert tuple(v.shape) == tuple(ref.shape) and (v.float() - ref.float()).abs().max().item() < 2e-5, k
            want_tensors -= {k}
    assert not want_tensors, want_tensors


def test_training_branches_are_loud():
    pt = PrepareTargets(num_frames=2, clip_class_embed_path=cases.clip_table())
    with pytest.raises(NotImplementedError):
        pt.process([], None, "cpu")
    with pytest.raises(NotImplementedError):
        pt.preprocess_text_prompt(None, [], [], is_train=True)
    with pytest.raises(AssertionError):
        PrepareTargets(clip_class_embed_path=cases.clip_table(), custom_videos_text=[["a"], ["b"]])
