"""Post/pre-norm attention + FFN layers with the reference's parameter names
(univs/modeling/transformer_decoder/transformer_layers.py:11-191); attention itself is
`layers.MultiheadAttention` (nn.MultiheadAttention-compatible state dict)."""
from typing import Optional

from torch import Tensor, nn

from ...layers import MLP, MultiheadAttention, get_activation_fn, layer_norm, linear, linear_act, linear_norm  # noqa: F401  (MLP re-exported)


def _with_pos(t, pos):
    return t if pos is None else t + pos


class SelfAttentionLayer(nn.Module):
    def __init__(self, d_model, nhead, dropout=0.0, activation="relu", normalize_before=False):
        super().__init__()
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.norm = nn.LayerNorm(d_model)
        self.activation = get_activation_fn(activation)
        self.normalize_before = normalize_before

    def forward(self, tgt, tgt_mask: Optional[Tensor] = None, tgt_key_padding_mask: Optional[Tensor] = None,
                query_pos: Optional[Tensor] = None, kv: Optional[Tensor] = None, kv_pos: Optional[Tensor] = None):
        """`kv` / `kv_pos` (frame-sharded decoding, univs_amd/distributed.py): `tgt` then holds only SOME of the tokens -- this
        rank's query rows -- and attends to the keys / values of ALL tokens in `kv`; `tgt_mask` is [rows, all].  Row r of the
        result equals row r of the full self-attention (…decoder_univs.py:408-414): softmax rows are independent."""
        assert tgt_key_padding_mask is None
        if kv is None:
            kv, kv_pos = tgt, query_pos
            full = True
        else:
            full = False
        if self.normalize_before:
            t2 = layer_norm(self.norm, tgt)
            kv2 = t2 if full else layer_norm(self.norm, kv)
            q = _with_pos(t2, query_pos)
            k = q if full else _with_pos(kv2, kv_pos)
            return tgt + self.self_attn(q, k, kv2, attn_mask=tgt_mask)[0]
        if full:
            # q = k = tgt + query_pos inside the projection kernel, norm(tgt + out_proj(.)) inside the output projection's
            return self.self_attn(tgt, tgt, tgt, attn_mask=tgt_mask, query_add=query_pos, residual=tgt, norm=self.norm)[0] \
                if query_pos is not None else self.self_attn(tgt, tgt, tgt, attn_mask=tgt_mask, residual=tgt, norm=self.norm)[0]
        q = _with_pos(tgt, query_pos)
        k = _with_pos(kv, kv_pos)
        return self.self_attn(q, k, kv, attn_mask=tgt_mask, residual=tgt, norm=self.norm)[0]


class CrossAttentionLayer(nn.Module):
    def __init__(self, d_model, nhead, dropout=0.0, activation="relu", normalize_before=False,
                 need_weights=False, average_attn_weights=False):
        super().__init__()
        self.multihead_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.norm = nn.LayerNorm(d_model)
        self.activation = get_activation_fn(activation)
        self.normalize_before = normalize_before
        self.need_weights = need_weights
        self.average_attn_weights = average_attn_weights

    def forward(self, tgt, memory, memory_mask: Optional[Tensor] = None,
                memory_key_padding_mask: Optional[Tensor] = None, pos: Optional[Tensor] = None,
                query_pos: Optional[Tensor] = None, key: Optional[Tensor] = None, kv=None):
        """`key`: optional precomputed `memory + pos` (it does not change across the decoder layers that
        attend to the same feature level, so the caller builds it once per level); `kv`: optional precomputed key / value
        PROJECTIONS of this layer (layers.MultiheadAttention.forward)."""
        assert memory_key_padding_mask is None
        src = layer_norm(self.norm, tgt) if self.normalize_before else tgt
        # nn.MultiheadAttention's default averages the returned weights over heads; the reference
        # never passes `average_attn_weights` through, so neither do we (transformer_layers.py:101-105)
        kk = key if (key is not None or kv is not None) else _with_pos(memory, pos)
        if self.normalize_before:
            out, w = self.multihead_attn(_with_pos(src, query_pos), kk, memory, attn_mask=memory_mask, need_weights=self.need_weights, kv=kv)
            tgt = tgt + out
        else:
            # `tgt + query_pos` inside the q projection, norm(tgt + out_proj(.)) inside the output projection (few rows: one launch each)
            tgt, w = self.multihead_attn(src, kk, memory, attn_mask=memory_mask, need_weights=self.need_weights, kv=kv,
                                         query_add=query_pos, residual=tgt, norm=self.norm)
        return (tgt, w) if self.need_weights else tgt


class FFNLayer(nn.Module):
    def __init__(self, d_model, dim_feedforward=2048, dropout=0.0, activation="relu", normalize_before=False):
        super().__init__()
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm = nn.LayerNorm(d_model)
        self.activation = get_activation_fn(activation)
        self.normalize_before = normalize_before

    def forward(self, tgt):
        if self.normalize_before:
            return tgt + self.linear2(linear_act(layer_norm(self.norm, tgt), self.linear1, self.activation))
        return linear_norm(linear_act(tgt, self.linear1, self.activation), self.linear2, tgt, self.norm)
