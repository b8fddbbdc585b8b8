"""Attention (lcasr/components/attention.py:448-551): fused qkv Linear (no bias by default) with the
reference's "(h d qkv)" column order, NeoX rotary on q,k, bidirectional softmax attention with optional
sliding window and key padding, output Linear.  FlashSelfAttention / SDPA are replaced by csrc/attention.hip."""
import torch.nn as nn

from .. import functional as Fn


def get_window_size(kwargs, direction=None):
    """attention.py:321-328."""
    if direction is None:
        return kwargs.get('attention_window_size', -1)
    if kwargs.get(f'attention_window_size_{direction}', None) is not None:
        return kwargs.get(f'attention_window_size_{direction}')
    return kwargs.get('attention_window_size', -1)


class Attention(nn.Module):
    def __init__(self, n_feats, head_dim, n_heads, dropout=0.0, **kwargs):
        super().__init__()
        if dropout != 0.0:
            raise NotImplementedError('attention dropout is 0 in every SConformerXL config')
        if kwargs.get('causal', False):
            raise NotImplementedError('causal attention is not on the hot path')
        self.layer_idx = kwargs.get('layer_idx', None)
        self.n_feats, self.head_dim, self.n_heads = n_feats, head_dim, n_heads
        self.dropout_p = dropout
        self.left_window, self.right_window = get_window_size(kwargs, 'left'), get_window_size(kwargs, 'right')
        self.causal = False
        self.return_attention_weights = kwargs.get('return_attention_weights', False)
        self.qkv_proj = nn.Linear(n_feats, 3 * n_heads * head_dim, bias=kwargs.get('qkv_bias', False))
        self.out_proj = nn.Linear(n_heads * head_dim, n_feats, bias=kwargs.get('bias', False))

    def forward_prenorm(self, x, norm, residual, lengths=None, rotary=None, **_):
        """x (B,N,d) f32.  lengths: int32 (B,) device tensor or None.  rotary: (cos, sin) compact tables or None."""
        B, N, _d = x.shape
        nw, nb = norm.norm_params()
        cos, sin = rotary if rotary is not None else (None, None)
        y = Fn.attn_block(x.reshape(B * N, -1), nw, nb, self.qkv_proj.weight, self.out_proj.weight, self.qkv_proj.bias,
                          self.out_proj.bias, cos, sin, lengths, B, N, self.n_heads, self.head_dim,
                          (self.left_window, self.right_window), norm.mode, norm.eps, residual)
        return y.view(B, N, -1)
