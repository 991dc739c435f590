"""CPU: the CLIP tokenizer and text tower (univs_amd/modeling/language) against goldens of the REAL reference
(oracle/gen_golden.py: g16a-c).  The tokenizer needs OpenAI CLIP's public merge table, which is data that is not shipped
here: set UNIVS_BPE_VOCAB (the dev container finds the reference's copy); without it the tokenizer tests are skipped."""
import os

import numpy as np
import pytest
import torch

from oracle.cpu_path import cpu_ops
from tests import cases

_CANDIDATES = (os.environ.get("UNIVS_BPE_VOCAB"), "/root/reference/univs/modeling/language/bpe_simple_vocab_16e6.txt.gz")
BPE = next((p for p in _CANDIDATES if p and os.path.isfile(p)), None)
needs_vocab = pytest.mark.skipif(BPE is None, reason="CLIP BPE merge table not available")


@needs_vocab
def test_tokenizer_matches_reference(golden_dir):
    from univs_amd.modeling.language import tokenizer as tk
    g = np.load(os.path.join(golden_dir, "g16a_tokenizer.npz"))
    tok = tk.SimpleTokenizer(BPE)
    exprs = [str(e) for e in g["expressions"]]
    ids = tk.pre_tokenize_expression(exprs, tok)
    assert ids.dtype == torch.long and tuple(ids.shape) == (len(exprs), 81, 77)
    assert torch.equal(ids, torch.from_numpy(g["ids"]).long())
    assert torch.equal(tk.pre_tokenize(["person", "traffic light"], tok), torch.from_numpy(g["class_ids"]).long())
    assert [tok.decode(tok.encode(e)) for e in exprs[:6]] == [str(d) for d in g["decoded"]]
    # the over-long text is cut at 77 tokens by the template expansion ...
    assert int((ids[-1, 0] != 0).sum()) == 77
    # ... and refused by tokenize()
    os.environ["UNIVS_BPE_VOCAB"] = BPE
    tk._shared_tokenizer.cache_clear()
    assert torch.equal(tk.tokenize(exprs[:6]), torch.from_numpy(g["plain"]).long())
    with pytest.raises(RuntimeError):
        tk.tokenize(exprs[-1])
    assert tk.clean_strings(["Traffic_light(1)", "a man's hat - red/blue!"]) == [str(c) for c in g["cleaned"]]


def test_missing_vocab_is_loud(monkeypatch, tmp_path):
    from univs_amd.modeling.language import tokenizer as tk
    monkeypatch.setenv("UNIVS_BPE_VOCAB", str(tmp_path / "nope.gz"))
    if not os.path.isfile(os.path.join(os.path.dirname(tk.__file__), "bpe_simple_vocab_16e6.txt.gz")):
        with pytest.raises(FileNotFoundError):
            tk.default_bpe()


def check_text_encoder(g, cfg, device, tol, stride=1):
    from univs_amd.modeling.prompt_encoder import TextPromptEncoder
    enc = cases.build_text_encoder(cfg, device)
    sample = torch.from_numpy(g["sample"]).long().to(device)
    x_word, x_eot = enc.encode_text(sample, only_eot=False)
    assert torch.equal(enc.encode_text(sample, only_eot=True), x_eot)
    e1 = (x_eot.cpu() - torch.from_numpy(g["x_eot"])).abs().max().item()
    e2 = (x_word.cpu()[:, ::stride] - torch.from_numpy(g["x_word"])).abs().max().item()
    assert e1 < tol and e2 < tol, (e1, e2)
    tokens = torch.from_numpy(g["tokens"]).long()
    E = tokens.shape[0]
    tpe = TextPromptEncoder(enc, num_frames=2, device=device)
    exprs = ["w " * (int(n) - 5) for n in g["exp_word_len"]]             # only the word count of the text is used here
    exprs = [e.strip() for e in exprs]
    w, s, n = tpe.get_expression_prompt(exprs, device, tokens=tokens, max_batch=100)   # 100: chunks straddle expressions
    assert tuple(w.shape) == (E, 77, 2, cfg["embed_dim"]) and tuple(s.shape) == (E, 2, cfg["embed_dim"])
    assert n == [int(v) for v in g["exp_word_len"]]
    assert torch.equal(w[:, :, 0], w[:, :, 1]) and torch.equal(s[:, 0], s[:, 1])
    e3 = (w[:, :, 0].cpu()[:, ::stride] - torch.from_numpy(g["exp_word_feats"])).abs().max().item()
    e4 = (s[:, 0].cpu() - torch.from_numpy(g["exp_sentence_feats"])).abs().max().item()
    assert e3 < tol and e4 < tol, (e3, e4)
    return max(e1, e2, e3, e4)


def test_text_encoder_small_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "g16b_text_encoder_small.npz"))
    with cpu_ops():
        check_text_encoder(g, cases.TEXT_SMALL, "cpu", 2e-5)


def test_text_encoder_rn50x4_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "g16c_text_encoder_full.npz"))
    with cpu_ops():
        check_text_encoder(g, cases.TEXT_FULL, "cpu", 5e-5, stride=4)


def test_state_dict_layout_is_the_reference_one():
    enc = cases.build_text_encoder(cases.TEXT_SMALL)
    keys = set(enc.state_dict())
    want = {"positional_embedding", "text_projection", "token_embedding.weight", "ln_final.weight", "ln_final.bias"}
    for i in range(3):
        b = f"transformer.resblocks.{i}."
        want |= {b + k for k in ("attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj.weight", "attn.out_proj.bias",
                                 "ln_1.weight", "ln_1.bias", "ln_2.weight", "ln_2.bias", "mlp.c_fc.weight",
                                 "mlp.c_fc.bias", "mlp.c_proj.weight", "mlp.c_proj.bias")}
    assert keys == want
