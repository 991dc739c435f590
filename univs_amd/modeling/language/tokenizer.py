"""Byte-pair tokenizer of the CLIP text encoder and the prompt-template expansion of expressions / class names.

Counterpart of `univs/modeling/language/clip_prompt_utils.py` (same function names and return conventions):
    SimpleTokenizer.encode / decode        :61-143   CLIP's lower-cased byte-level BPE (49 152 merges-based vocabulary,
                                                      <|startoftext|> = 49406, <|endoftext|> = 49407)
    tokenize                               :150-165  [n, 77] int64, raises when a text does not fit
    get_prompt_templates                   :169-333  the 81 CLIP prompt-engineering templates ('{}.' + the 80 ImageNet ones)
    convert_example_to_features_bpe        :340-358  SOT + ids + EOT, truncated to the context and 0-padded
    pre_tokenize / pre_tokenize_expression :416-478  [#texts, 81, 77] int64
    clean_strings / clean_string_exp       :481-503

The merge table is OpenAI CLIP's public `bpe_simple_vocab_16e6.txt.gz`; it is data, not shipped in this repository: pass
its path, or set UNIVS_BPE_VOCAB, or put the file next to this module (where the reference keeps its copy).
Unicode repair: the reference runs `ftfy.fix_text` first; when ftfy is not installed that step is skipped (identity for
well-formed text, which is what the datasets' expressions are).
"""
import functools
import gzip
import html
import os
import re as _re

import regex
import torch

CONTEXT_LENGTH = 77
_N_MERGES = 49152 - 256 - 2
_WORD_END = "</w>"
_SOT, _EOT = "<|startoftext|>", "<|endoftext|>"
_SPLIT = regex.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
                       regex.IGNORECASE)


def default_bpe():
    for cand in (os.environ.get("UNIVS_BPE_VOCAB"),
                 os.path.join(os.path.dirname(os.path.abspath(__file__)), "bpe_simple_vocab_16e6.txt.gz")):
        if cand and os.path.isfile(cand):
            return cand
    raise FileNotFoundError("CLIP BPE merge table not found: set UNIVS_BPE_VOCAB to bpe_simple_vocab_16e6.txt.gz")


@functools.lru_cache()
def bytes_to_unicode():
    """byte value -> printable stand-in character: printable latin-1 bytes stand for themselves, the other 68 are moved
    to code points 256.. in byte order (the GPT-2 / CLIP convention)."""
    keep = set(range(ord("!"), ord("~") + 1)) | set(range(0xA1, 0xAC + 1)) | set(range(0xAE, 0xFF + 1))
    table, extra = {}, 0
    for b in sorted(keep):
        table[b] = chr(b)
    for b in range(256):
        if b not in keep:
            table[b] = chr(256 + extra)
            extra += 1
    # the vocabulary order below depends on insertion order: kept bytes first, then the moved ones
    return table


def _fix_text(text):
    try:
        import ftfy
        text = ftfy.fix_text(text)
    except ImportError:
        pass
    return html.unescape(html.unescape(text)).strip()


class SimpleTokenizer:
    def __init__(self, bpe_path: str = None):
        bpe_path = bpe_path or default_bpe()
        self.byte_encoder = bytes_to_unicode()
        self.byte_decoder = {c: b for b, c in self.byte_encoder.items()}
        with gzip.open(bpe_path) as f:
            lines = f.read().decode("utf-8").split("\n")
        merges = [tuple(line.split()) for line in lines[1:_N_MERGES + 1]]
        singles = list(self.byte_encoder.values())
        vocab = singles + [c + _WORD_END for c in singles] + ["".join(m) for m in merges] + [_SOT, _EOT]
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.decoder = {i: tok for tok, i in self.encoder.items()}
        self.bpe_ranks = {m: r for r, m in enumerate(merges)}
        self.cache = {_SOT: _SOT, _EOT: _EOT}

    def bpe(self, token: str) -> str:
        """Greedy merges of one pre-token (already mapped to stand-in characters): repeatedly fuse every occurrence of
        the adjacent pair with the lowest merge rank.  Returns the pieces joined by spaces."""
        hit = self.cache.get(token)
        if hit is not None:
            return hit
        parts = list(token[:-1]) + [token[-1] + _WORD_END]
        while len(parts) > 1:
            best, best_rank = None, None
            for pair in zip(parts[:-1], parts[1:]):
                r = self.bpe_ranks.get(pair)
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = pair, r
            if best is None:
                break
            fused, i = [], 0
            while i < len(parts):
                if i + 1 < len(parts) and parts[i] == best[0] and parts[i + 1] == best[1]:
                    fused.append(parts[i] + parts[i + 1])
                    i += 2
                else:
                    fused.append(parts[i])
                    i += 1
            parts = fused
        out = " ".join(parts)
        self.cache[token] = out
        return out

    def encode(self, text: str, return_link: bool = False):
        text = _re.sub(r"\s+", " ", _fix_text(text)).strip().lower()
        ids, links = [], []
        for word in _SPLIT.findall(text):
            mapped = "".join(self.byte_encoder[b] for b in word.encode("utf-8"))
            word_ids = [self.encoder[p] for p in self.bpe(mapped).split(" ")]
            ids.extend(word_ids)
            links.append([word, word_ids])
        return (ids, links) if return_link else ids

    def decode(self, tokens) -> str:
        chars = "".join(self.decoder[int(t)] for t in tokens)
        return bytearray(self.byte_decoder[c] for c in chars).decode("utf-8", errors="replace").replace(_WORD_END, " ")


@functools.lru_cache()
def _shared_tokenizer():
    return SimpleTokenizer()


def tokenize(texts, context_length: int = CONTEXT_LENGTH) -> torch.Tensor:
    if isinstance(texts, str):
        texts = [texts]
    tok = _shared_tokenizer()
    sot, eot = tok.encoder[_SOT], tok.encoder[_EOT]
    out = torch.zeros(len(texts), context_length, dtype=torch.long)
    for i, text in enumerate(texts):
        ids = [sot] + tok.encode(text) + [eot]
        if len(ids) > context_length:
            raise RuntimeError(f"Input {text} is too long for context length {context_length}")
        out[i, :len(ids)] = torch.tensor(ids)
    return out


_SUBJECTS = ("a photo of a {}.", "a bad photo of a {}.", "a photo of many {}.", "a sculpture of a {}.",
             "a photo of the hard to see {}.", "a low resolution photo of the {}.", "a rendering of a {}.",
             "graffiti of a {}.", "a bad photo of the {}.", "a cropped photo of the {}.", "a tattoo of a {}.",
             "the embroidered {}.", "a photo of a hard to see {}.", "a bright photo of a {}.", "a photo of a clean {}.",
             "a photo of a dirty {}.", "a dark photo of the {}.", "a drawing of a {}.", "a photo of my {}.",
             "the plastic {}.", "a photo of the cool {}.", "a close-up photo of a {}.",
             "a black and white photo of the {}.", "a painting of the {}.", "a painting of a {}.",
             "a pixelated photo of the {}.", "a sculpture of the {}.", "a bright photo of the {}.",
             "a cropped photo of a {}.", "a plastic {}.", "a photo of the dirty {}.", "a jpeg corrupted photo of a {}.",
             "a blurry photo of the {}.", "a photo of the {}.", "a good photo of the {}.", "a rendering of the {}.",
             "a {} in a video game.", "a photo of one {}.", "a doodle of a {}.", "a close-up photo of the {}.",
             "the origami {}.", "the {} in a video game.", "a sketch of a {}.", "a doodle of the {}.", "a origami {}.",
             "a low resolution photo of a {}.", "the toy {}.", "a rendition of the {}.", "a photo of the clean {}.",
             "a photo of a large {}.", "a rendition of a {}.", "a photo of a nice {}.", "a photo of a weird {}.",
             "a blurry photo of a {}.", "a cartoon {}.", "art of a {}.", "a sketch of the {}.", "a embroidered {}.",
             "a pixelated photo of a {}.", "itap of the {}.", "a jpeg corrupted photo of the {}.",
             "a good photo of a {}.", "a plushie {}.", "a photo of the nice {}.", "a photo of the small {}.",
             "a photo of the weird {}.", "the cartoon {}.", "art of the {}.", "a drawing of the {}.",
             "a photo of the large {}.", "a black and white photo of a {}.", "the plushie {}.", "a dark photo of a {}.",
             "itap of a {}.", "graffiti of the {}.", "a toy {}.", "itap of my {}.", "a photo of a cool {}.",
             "a photo of a small {}.", "a tattoo of the {}.")


def get_prompt_templates():
    """The bare text first (its per-token features become `exp_word_feats`), then CLIP's 80 ImageNet templates."""
    return ["{}."] + list(_SUBJECTS)


def prompt_engineering(classname: str, template: str = "") -> str:
    return template.replace("{}", classname.replace("/", "").replace(",", "").replace("+", " "))


def convert_example_to_features_bpe(text, tokenizer, sot_token, eot_token, context_length: int = CONTEXT_LENGTH):
    """-> list of `context_length` token ids: SOT + BPE ids + EOT, cut at the context length, 0-padded."""
    assert isinstance(text, str)
    ids = ([sot_token] + tokenizer.encode(text) + [eot_token])[:context_length]
    return ids + [0] * (context_length - len(ids))


def _expand(texts_per_entry, tokenizer=None):
    tok = tokenizer or _shared_tokenizer()
    sot, eot = tok.encoder[_SOT], tok.encoder[_EOT]
    rows = [[convert_example_to_features_bpe(t, tok, sot, eot) for t in texts] for texts in texts_per_entry]
    return torch.tensor(rows, dtype=torch.long)


def pre_tokenize(class_names, tokenizer=None) -> torch.Tensor:
    """class_names: list of names or of synonym lists -> [#classes, 81 * #synonyms, 77] (template-major order)."""
    templates = get_prompt_templates()
    per_class = []
    for entry in class_names:
        names = [entry] if isinstance(entry, str) else list(entry)
        per_class.append([prompt_engineering(n, template=pt) for pt in templates for n in names])
    return _expand(per_class, tokenizer)


def pre_tokenize_expression(expressions, tokenizer=None) -> torch.Tensor:
    """expressions: a sentence or a list of sentences -> [#expressions, 81, 77]."""
    if isinstance(expressions, str):
        expressions = [expressions]
    templates = get_prompt_templates()
    for e in expressions:
        assert isinstance(e, str)
    return _expand([[pt.replace("{}", e) for pt in templates] for e in expressions], tokenizer)


_DROP = set("0123456789()")


def clean_string_exp(expression: str) -> str:
    return _re.sub(r"([.,'!?\"()*#:;])", "", expression.lower()).replace("-", " ").replace("/", " ")


def clean_strings(strings):
    """underscores -> spaces, digits and parentheses dropped, then `clean_string_exp` (lists are cleaned in place)."""
    def one(s):
        return clean_string_exp("".join(ch for ch in " ".join(s.split("_")) if ch not in _DROP))
    if isinstance(strings, list):
        for i, s in enumerate(strings):
            strings[i] = one(s)
        return strings
    return one(strings)
