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
        """x (B,N,d) f32.  lengths: int32 (B,) device tensor or None.  rotary: (cos, sin) compact tables or None.
        norm None: x is already normalised (the module-level `forward`)."""
        B, N, _d = x.shape
        nw, nb = norm.norm_params() if norm is not None else (None, None)
        mode, eps = (norm.mode, norm.eps) if norm is not None else ('none', 0.0)
        cos, sin = rotary if rotary is not None else (None, None)
        y = Fn.attn_block(x.reshape(B * N, -1), nw, nb, self.qkv_proj.weight, self.out_proj.weight, self.qkv_proj.bias,
                          self.out_proj.bias, cos, sin, lengths, B, N, self.n_heads, self.head_dim,
                          (self.left_window, self.right_window), mode, eps, residual)
        return y.view(B, N, -1)

    def forward(self, x, attn_mask=None, length=None, pad_mask=None, flash_attn=True, rotary_emb_fn=None):
        """The reference's module-level call (attention.py:509-551) on an already normalised x (B,N,d).
        pad_mask (B,N) bool, True = padded position: must be the suffix mask sconformer_xl.py:207 builds from `length`;
        attn_mask carries the same information in the reference and is ignored here.  rotary_emb_fn: the reference's
        `apply_rotary` object (.cos/.sin of shape (1,n,1,D)) or a (cos, sin) pair of compact (n, D/2) tables."""
        import torch
        B, N, _d = x.shape
        lengths = None
        if pad_mask is not None:
            lengths = (~pad_mask).sum(-1).to(dtype=torch.int32).contiguous()
        elif length is not None and int(length.max()) != int(length.min()):
            lengths = length.to(device=x.device, dtype=torch.int32).contiguous()
        rotary = None
        if rotary_emb_fn is not None:
            if isinstance(rotary_emb_fn, (tuple, list)):
                rotary = tuple(rotary_emb_fn)
            else:
                if getattr(rotary_emb_fn, 'learned', False):
                    raise NotImplementedError('learned rotary frequencies are not on the hot path')
                h = self.head_dim // 2
                rotary = (rotary_emb_fn.cos[0, :N, 0, :h].float().contiguous(), rotary_emb_fn.sin[0, :N, 0, :h].float().contiguous())
        return self.forward_prenorm(x, None, False, lengths=lengths, rotary=rotary)
