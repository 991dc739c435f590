"""CLIP text transformer that turns referring expressions / class names into the text prompts of the hot path
(`exp_word_feats`, `exp_sentence_feats`, the class-embedding table).

Counterpart of `univs/modeling/language/TextEncoder.py` (same class / function names and the same state-dict keys:
`token_embedding.weight`, `positional_embedding`, `transformer.resblocks.{i}.{ln_1,attn,ln_2,mlp.c_fc,mlp.c_proj}`,
`ln_final`, `text_projection`), so the reference's converted CLIP checkpoints load unchanged:
    ResidualAttentionBlock  :22-43    x += attn(ln_1(x)) [causal]; x += c_proj(QuickGELU(c_fc(ln_2(x))))
    CLIPLangEncoder         :57-141   encode_text(text, only_eot) -> sentence feature at the EOT token (and per-token
                                      features) projected by `text_projection`
    build_clip_language_encoder :144-184  widths / heads per CLIP visual depth flag (RN50 / RN101 / RN50x4)
Inference only, fp32.  The residual add is fused with the following LayerNorm (one HIP pass, `ops.layer_norm` with
`return_sum`), the causal mask + softmax is the in-place HIP masked softmax; GEMMs are hipBLASLt through ATen.
"""
import pickle

import torch
from torch import nn

from ...layers import MultiheadAttention, fp32_region, layer_norm


class QuickGELU(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(1.702 * x)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d_model: int, n_head: int):
        super().__init__()
        self.attn = MultiheadAttention(d_model, n_head)
        self.ln_1 = nn.LayerNorm(d_model)
        self.mlp = nn.Sequential()
        self.mlp.add_module("c_fc", nn.Linear(d_model, d_model * 4))
        self.mlp.add_module("gelu", QuickGELU())
        self.mlp.add_module("c_proj", nn.Linear(d_model * 4, d_model))
        self.ln_2 = nn.LayerNorm(d_model)

    def forward(self, x, normed, causal):
        """x: residual stream [L, N, E]; normed = ln_1(x), produced by the previous block's fused add + LayerNorm.
        Returns (x + attention branch, MLP branch); the caller fuses the last add with the next block's ln_1."""
        a = self.attn(normed, normed, normed, attn_mask=causal)[0]
        x, h = layer_norm(self.ln_2, a, residual=x, return_sum=True)        # x <- x + a ; h = ln_2(x)
        return x, self.mlp(h)


class Transformer(nn.Module):
    def __init__(self, width: int, layers: int, heads: int):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads) for _ in range(layers)])


class CLIPLangEncoder(nn.Module):
    def __init__(self, embed_dim: int, context_length: int, vocab_size: int, transformer_width: int,
                 transformer_heads: int, transformer_layers: int, out_features=None, freeze_at=None):
        super().__init__()
        self.context_length = context_length
        self.vocab_size = vocab_size
        self.transformer = Transformer(transformer_width, transformer_layers, transformer_heads)
        self.token_embedding = nn.Embedding(vocab_size, transformer_width)
        self.positional_embedding = nn.Parameter(torch.zeros(context_length, transformer_width))
        self.ln_final = nn.LayerNorm(transformer_width)
        self.text_projection = nn.Parameter(torch.zeros(transformer_width, embed_dim))
        # True = not attended (tokens only look backwards), TextEncoder.py:106-112
        self.register_buffer("causal_mask", torch.ones(context_length, context_length, dtype=torch.bool).triu_(1), False)

    @property
    def dtype(self):
        return self.text_projection.dtype

    @property
    def device(self):
        return self.token_embedding.weight.device

    @torch.no_grad()
    @fp32_region
    def encode_text(self, text: torch.Tensor, only_eot: bool = True):
        """text: int64 [N, context_length] token ids (0-padded; the end-of-text token has the highest id).
        -> x_eot [N, embed_dim]  (and x_word [N, context_length, embed_dim] first when only_eot=False)."""
        N, L = text.shape
        if L != self.context_length:
            raise ValueError(f"expected {self.context_length} tokens per text, got {L}")
        x = self.token_embedding(text.t())                      # [L, N, E] sequence-first, no transposes later
        blocks = self.transformer.resblocks
        # x <- emb + pos ; h = ln_1(x) of the first block
        x, h = layer_norm(blocks[0].ln_1, x, residual=self.positional_embedding[:, None, :].expand(L, N, -1).contiguous(),
                          return_sum=True)
        for i, blk in enumerate(blocks):
            x, m = blk(x, h, self.causal_mask)
            nxt = blocks[i + 1].ln_1 if i + 1 < len(blocks) else self.ln_final
            x, h = layer_norm(nxt, m, residual=x, return_sum=True)           # x <- x + mlp ; h = next norm
        # h = ln_final(x) [L, N, E]
        eot = text.argmax(dim=-1)
        x_eot = h[eot, torch.arange(N, device=h.device)] @ self.text_projection
        if only_eot:
            return x_eot
        x_word = (h.reshape(L * N, -1) @ self.text_projection).view(L, N, -1).transpose(0, 1)
        return x_word, x_eot


CLIP_TEXT_CONFIGS = {
    # RESNETS_DEPTH flag -> (embed_dim, width, heads)           TextEncoder.py:157-173
    50: (1024, 512, 8),
    101: (512, 512, 8),
    200: (640, 640, 10),      # RN50x4, the one UniVS uses (640-d text prompts)
}


def build_clip_language_encoder(cfg):
    embed_dim, width, heads = CLIP_TEXT_CONFIGS[cfg.MODEL.CLIP.RESNETS_DEPTH]
    model = CLIPLangEncoder(embed_dim, 77, 49408, width, heads, 12, ["res5"], cfg.MODEL.CLIP.BACKBONE_FREEZE_AT)
    load_checkpoint(model, cfg.MODEL.CLIP.WEIGHTS)
    return model.eval()


def load_checkpoint(model, path):
    """`.pkl` (pickled dict) or torch checkpoint with the weights under 'model' (TextEncoder.py:186-195)."""
    if not path:
        raise ValueError("MODEL.CLIP.WEIGHTS is empty")
    if path.endswith("pkl"):
        with open(path, "rb") as f:
            ckpt = pickle.load(f)
    else:
        ckpt = torch.load(path, map_location="cpu")
    sd = {k: torch.as_tensor(v) for k, v in ckpt["model"].items()}
    model.load_state_dict(sd, strict=True)
