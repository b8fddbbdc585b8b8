"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU fp32 restatement of the reference SConformerXL forward + CTC loss
(`/root/reference/lcasr/models/sconformer_xl.py` and `lcasr/components/*`),
written as explicit tensor math over a reference-layout ``state_dict`` so the
HIP path can be checked against it on the GPU box, where the reference itself
does not exist.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  The product package
(``long-context-asr_amd/``) never does; it fails loudly when the HIP library
is missing.

Parity pin: `oracle/make_golden.py` (run in the development container, where
`/root/reference` is importable) checks every function here against the
reference modules run on CPU and writes the fixtures in ``tests/golden/``;
``tests/test_oracle_golden.py`` re-checks the restatement against those
fixtures without the reference.  The reference itself holds no tests or
golden vectors, so the fixtures generated from the imported
reference are the pin.

Third-party arithmetic the reference calls on this path and that is used
here the same way: ``torch`` (conv2d/conv1d/layer_norm/softmax/ctc_loss).
Attention is written out as explicit softmax(QK^T/sqrt(D))V rather than via
``F.scaled_dot_product_attention`` (reference: attention.py:541) so that the
masking semantics are visible.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------
# configuration (mirrors SCConformerXL.__init__ kwargs, sconformer_xl.py:32-64)
# --------------------------------------------------------------------------
DEFAULTS = dict(
    vocab_size=128, feat_in=80, subsampling='dw_striding', subsampling_factor=8,
    subsampling_conv_channels=256, subsampling_act='silu', subsampling_norm_out=False,
    n_layers=6, d_model=768, n_heads=6, head_dim=128, expansion_factor=4,
    conv_kernel_size=9, conv_expansion_factor=1, decoder_norm=False, use_rotary=False,
    rotary_interpolation_factor=1.0, self_conditioning=True, default_norm='layer_norm',
    bias_in_ff=False, legasee_double_norm=True, rotary_base_freq=10000,
    checkpoint_every_n_layers=0, ff_checkpoint_lvl=0,
)


def make_config(**kw) -> dict:
    cfg = dict(DEFAULTS)
    cfg.update(kw)
    if cfg['subsampling_conv_channels'] == -1:  # sconformer_xl.py:104
        cfg['subsampling_conv_channels'] = cfg['d_model']
    return cfg


# --------------------------------------------------------------------------
# small pieces
# --------------------------------------------------------------------------
def calc_length(lengths: Tensor, repeat_num: int = 3) -> Tensor:
    """subsampling.py:557-567 with all_paddings=2, kernel=3, stride=2, floor."""
    lengths = lengths.to(torch.float)
    for _ in range(repeat_num):
        lengths = torch.floor((lengths + (2 - 3)) / 2 + 1.0)
    return lengths.to(torch.int)


def norm(x: Tensor, sd: Dict[str, Tensor], prefix: str, kind: str) -> Tensor:
    """LayerNorm (torch.nn.LayerNorm, eps 1e-5) or the local RMSNorm
    (normalisation.py:6-47: x / (||x||_2 * d^-1/2 + 1e-8) * scale)."""
    if kind == 'layer_norm':
        return F.layer_norm(x, (x.shape[-1],), sd[prefix + '.weight'], sd[prefix + '.bias'], 1e-5)
    d = x.shape[-1]
    rms = x.norm(2, dim=-1, keepdim=True) * d ** (-0.5)
    return sd[prefix + '.scale'] * (x / (rms + 1e-8))


def rotary_tables(n: int, head_dim: int, base: float, interp: float = 1.0):
    """rotary_emb.py:23,44-57 → cos/sin of shape (n, head_dim)."""
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2).float() / head_dim))
    t = torch.arange(n).float() / interp
    freqs = torch.einsum('i,j->ij', t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def rotate_half(x: Tensor) -> Tensor:
    """rotary_emb.py:61-65."""
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def apply_rotary(x: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    """x: (B,N,H,D); cos/sin (N,D).  rotary_emb.py:68-73."""
    return x * cos[None, :, None, :] + rotate_half(x) * sin[None, :, None, :]


def gelu_tanh(x: Tensor) -> Tensor:
    return F.gelu(x, approximate='tanh')


# --------------------------------------------------------------------------
# subsampler (subsampling.py:276-321, 384-428), dw_striding x8
# --------------------------------------------------------------------------
def subsample(x_bft: Tensor, lengths: Tensor, sd, cfg, cap: Optional[dict] = None):
    """x_bft: (B, feat, T) as handed to SCConformerXL.forward.  Returns (B,N,d), lengths."""
    p = 'subsampling.'
    x = x_bft.transpose(1, 2).unsqueeze(1)                     # (B,1,T,F)  sconformer_xl.py:185, subsampling.py:393
    x = F.silu(F.conv2d(x, sd[p + 'conv.0.weight'], sd[p + 'conv.0.bias'], stride=2, padding=1))
    if cap is not None: cap['sub.stage0'] = x
    for dw, pw in (('conv.2', 'conv.3'), ('conv.5', 'conv.6')):
        C = x.shape[1]
        x = F.conv2d(x, sd[p + dw + '.weight'], sd[p + dw + '.bias'], stride=2, padding=1, groups=C)
        x = F.silu(F.conv2d(x, sd[p + pw + '.weight'], sd[p + pw + '.bias']))
        if cap is not None: cap['sub.' + pw] = x
    b, c, t, f = x.shape
    x = x.transpose(1, 2).reshape(b, t, c * f)                 # subsampling.py:422-423
    x = F.linear(x, sd[p + 'out.weight'], sd.get(p + 'out.bias'))
    return x, calc_length(lengths)


# --------------------------------------------------------------------------
# conformer layer pieces
# --------------------------------------------------------------------------
def feed_forward(x: Tensor, sd, prefix: str, kind: str) -> Tensor:
    """Scale(0.5, PreNorm(FusedMLP)) — fused_dense.py:464-470, wrappers.py:5-28."""
    h = norm(x, sd, prefix + '.fn.norm', kind)
    h = F.linear(h, sd[prefix + '.fn.fn.fc1.weight'], sd.get(prefix + '.fn.fn.fc1.bias'))
    h = gelu_tanh(h)
    h = F.linear(h, sd[prefix + '.fn.fn.fc2.weight'], sd.get(prefix + '.fn.fn.fc2.bias'))
    return h * 0.5


def attention(x: Tensor, sd, prefix: str, kind: str, cfg, lengths: Optional[Tensor],
              rot, window=(-1, -1), cap: Optional[dict] = None) -> Tensor:
    """PreNorm(Attention) — attention.py:509-551 (CPU branch), masks per
    sconformer_xl.py:204-213 and SURVEY A.3.  `lengths` None ⇒ no masking."""
    B, N, _ = x.shape
    H, D = cfg['n_heads'], cfg['head_dim']
    h = norm(x, sd, prefix + '.norm', kind)
    pad = None
    if lengths is not None:
        pad = torch.arange(N)[None, :] >= lengths[:, None]      # (B,N) True = padded
        h = h.masked_fill(pad[..., None], 0.0)                  # attention.py:511
    qkv = F.linear(h, sd[prefix + '.fn.qkv_proj.weight'], sd.get(prefix + '.fn.qkv_proj.bias'))
    qkv = qkv.reshape(B, N, H, D, 3)                            # "b n (h d qkv)" attention.py:485
    q, k, v = qkv[..., 0], qkv[..., 1], qkv[..., 2]             # each (B,N,H,D)
    if rot is not None:
        q, k = apply_rotary(q, *rot), apply_rotary(k, *rot)
    if cap is not None:
        cap[prefix + '.q'], cap[prefix + '.k'], cap[prefix + '.v'] = q, k, v
    s = torch.einsum('bihd,bjhd->bhij', q, k) / math.sqrt(D)
    neg = -torch.finfo(s.dtype).max
    if pad is not None:
        # additive mask where EITHER query or key is padded (sconformer_xl.py:211-213)
        m = pad[:, None, :, None] | pad[:, None, None, :]
        s = s + m.to(s.dtype) * neg
    if window[0] >= 0 or window[1] >= 0:
        # flash-attn local window semantics (attention.py:330-410 / construct_local_mask):
        # key j visible to query i iff i - left <= j <= i + right (seqlen_q == seqlen_k)
        i = torch.arange(N)[:, None]; j = torch.arange(N)[None, :]
        left = window[0] if window[0] >= 0 else N
        right = window[1] if window[1] >= 0 else N
        s = s.masked_fill(((j < i - left) | (j > i + right))[None, None], float('-inf'))
    a = s.softmax(dim=-1)
    o = torch.einsum('bhij,bjhd->bihd', a, v).reshape(B, N, H * D)
    if pad is not None:
        o = o.masked_fill(pad[..., None], 0.0)                  # attention.py:546-547
    if cap is not None: cap[prefix + '.o'] = o
    return F.linear(o, sd[prefix + '.fn.out_proj.weight'], sd.get(prefix + '.fn.out_proj.bias'))


def brn_rmax_dmax(nbt: Tensor):
    """batchrenorm.py:40-50."""
    rmax = (2 / 35000 * nbt + 25 / 35).clamp(1.0, 3.0)
    dmax = (5 / 20000 * nbt - 25 / 20).clamp(0.0, 5.0)
    return rmax, dmax


def conv_module(x: Tensor, sd, prefix: str, kind: str, lengths: Optional[Tensor], training: bool,
                new_buffers: Optional[dict] = None, cap: Optional[dict] = None) -> Tensor:
    """PreNorm(ConformerConvolution) — convolution.py:103-124 with BatchRenorm1d
    (batchrenorm.py:52-92).  Works token-major (B,N,C); statistics are over all
    B*N positions (padded zeros included: the conv module passes no mask)."""
    B, N, d = x.shape
    p = prefix + '.fn.'
    h = norm(x, sd, prefix + '.norm', kind)
    h = F.linear(h, sd[p + 'pointwise_conv1.weight'].squeeze(-1), sd[p + 'pointwise_conv1.bias'])
    a, g = h[..., :d], h[..., d:]                               # GLU over channels (dim=1 in (B,C,N))
    h = a * torch.sigmoid(g)
    if lengths is not None:
        pad = torch.arange(N)[None, :] >= lengths[:, None]
        h = h.masked_fill(pad[..., None], 0.0)                  # convolution.py:109-110
    if cap is not None: cap[prefix + '.glu'] = h
    k = sd[p + 'depthwise_conv.weight'].shape[-1]
    h = F.conv1d(h.transpose(1, 2), sd[p + 'depthwise_conv.weight'], sd[p + 'depthwise_conv.bias'],
                 padding=(k - 1) // 2, groups=d).transpose(1, 2)          # (B,N,d)
    if cap is not None: cap[prefix + '.dw'] = h
    rm, rs = sd[p + 'batch_norm.running_mean'], sd[p + 'batch_norm.running_std']
    if training:
        flat = h.reshape(-1, d)
        mean = flat.mean(0)
        std = flat.std(0, unbiased=False) + 1e-3
        rmax, dmax = brn_rmax_dmax(sd[p + 'batch_norm.num_batches_tracked'])
        r = (std.detach() / rs).clamp(1 / rmax, rmax)
        dd = ((mean.detach() - rm) / rs).clamp(-dmax, dmax)
        h = (h - mean) / std * r + dd
        if new_buffers is not None:
            new_buffers[p + 'batch_norm.running_mean'] = rm + 0.01 * (mean.detach() - rm)
            new_buffers[p + 'batch_norm.running_std'] = rs + 0.01 * (std.detach() - rs)
            new_buffers[p + 'batch_norm.num_batches_tracked'] = sd[p + 'batch_norm.num_batches_tracked'] + 1
    else:
        h = (h - rm) / rs
    h = sd[p + 'batch_norm.weight'] * h + sd[p + 'batch_norm.bias']
    h = F.silu(h)
    if cap is not None: cap[prefix + '.brn_silu'] = h
    return F.linear(h, sd[p + 'pointwise_conv2.weight'].squeeze(-1), sd[p + 'pointwise_conv2.bias'])


def conformer_layer(x, sd, i, cfg, lengths, rot, training, new_buffers=None, cap=None, window=(-1, -1)):
    """ConformerLayer.forward — sconformer_xl.py:346-372 (dropouts are p=0)."""
    kind = cfg['default_norm']
    L = f'layers.{i}'
    x = feed_forward(x, sd, L + '.ff1', kind) + x
    if cap is not None: cap[L + '.after_ff1'] = x
    x = attention(x, sd, L + '.attend', kind, cfg, lengths, rot, window, cap) + x
    if cap is not None: cap[L + '.after_attn'] = x
    x = conv_module(x, sd, L + '.conv', kind, lengths, training, new_buffers, cap) + x
    if cap is not None: cap[L + '.after_conv'] = x
    x = feed_forward(x, sd, L + '.ff2', kind) + x
    x = norm(x, sd, L + '.norm_out', kind)
    if cap is not None: cap[L + '.out'] = x
    return x


def checkpointed_layer(x, sd, i, cfg, lengths, rot, new_buffers, window=(-1, -1)):
    """torch.utils.checkpoint around a layer in TRAIN mode — sconformer_xl.py:221-230.  The layer runs twice: once in the
    forward (no graph; its output is what the rest of the forward sees) and once more inside the backward (the recompute, whose
    graph carries the gradient).  BatchRenorm is stateful, so the recompute (a) reads the running statistics the first run
    already moved - its r / d clamps, which are constants of the graph, differ from the first run's - and (b) moves the
    running statistics and num_batches_tracked a SECOND time (measured on the reference: nbt = 2 after one step).
    ff_checkpoint_lvl (fused_dense.py:283-289) only changes what is kept for the backward, never a value."""
    nb1: dict = {}
    with torch.no_grad():
        y1 = conformer_layer(x.detach(), sd, i, cfg, lengths, rot, True, nb1, None, window)
    nb2: dict = {}
    y2 = conformer_layer(x, {**sd, **nb1}, i, cfg, lengths, rot, True, nb2, None, window)
    if new_buffers is not None:
        new_buffers.update(nb2)
    return y2 + (y1 - y2).detach()


def decoder_logits(x, sd, cfg):
    """ASRLinearSCDecoder.forward(logits=True) — decoder.py:22-26."""
    if cfg['decoder_norm']:
        x = norm(x, sd, 'decoder.norm', cfg['default_norm'])
    return F.linear(x, sd['decoder.ff.weight'], sd['decoder.ff.bias'])


# --------------------------------------------------------------------------
# whole model
# --------------------------------------------------------------------------
def forward(sd: Dict[str, Tensor], cfg: dict, audio_bft: Tensor, lengths: Optional[Tensor] = None,
            training: bool = True, return_logits: bool = False, new_buffers: Optional[dict] = None,
            cap: Optional[dict] = None, window=(-1, -1)):
    """SCConformerXL.forward — sconformer_xl.py:162-252.
    Returns {'final_posteriors': (B,N,V+1), 'length': (B,) int32}."""
    B, _, T = audio_bft.shape
    if lengths is None:
        lengths = torch.tensor([T] * B)
    x, length = subsample(audio_bft, lengths, sd, cfg, cap)
    if cap is not None: cap['sub.out'] = x
    N = x.shape[1]
    rot = None
    if cfg['use_rotary']:
        n_rot = int(length.max())                               # sconformer_xl.py:196-199
        cos, sin = rotary_tables(n_rot, cfg['head_dim'], cfg['rotary_base_freq'], cfg['rotary_interpolation_factor'])
        if n_rot < N:                                           # reference would fail to broadcast; keep strict
            raise ValueError('rotary table shorter than sequence (reference would error)')
        rot = (cos[:N], sin[:N])
    mlen = None if int(length.max()) == int(length.min()) else length   # sconformer_xl.py:204-205
    nb = new_buffers
    every = cfg.get('checkpoint_every_n_layers', 0)
    for i in range(cfg['n_layers']):
        if every > 0 and i % every == 0 and training and torch.is_grad_enabled() and x.requires_grad:
            x = checkpointed_layer(x, sd, i, cfg, mlen, rot, nb, window)
        else:
            x = conformer_layer(x, sd, i, cfg, mlen, rot, training, nb, cap, window)
        if i != cfg['n_layers'] - 1 and cfg['self_conditioning']:
            post = decoder_logits(x, sd, cfg).softmax(dim=-1)   # sconformer_xl.py:241-243
            x = x + F.linear(post, sd['decoder.reprojection.weight'], sd['decoder.reprojection.bias'])
            if cap is not None: cap[f'layers.{i}.after_sc'] = x
    if cfg['legasee_double_norm'] and cfg['decoder_norm']:
        x = norm(x, sd, 'decoder.norm', cfg['default_norm'])    # sconformer_xl.py:246
    logits = decoder_logits(x, sd, cfg)
    if cap is not None: cap['logits'] = logits
    out = logits if return_logits else F.log_softmax(logits, dim=-1)
    return {'final_posteriors': out, 'length': length}


def ctc_loss_sum(log_probs_bnc: Tensor, targets: Tensor, input_lengths: Tensor, target_lengths: Tensor) -> Tensor:
    """exp/train.py:104,249 — CTCLoss(blank=C-1, reduction='sum') on the (N,B,C) view."""
    C = log_probs_bnc.shape[-1]
    return F.ctc_loss(log_probs_bnc.transpose(0, 1), targets, input_lengths, target_lengths,
                      blank=C - 1, reduction='sum', zero_infinity=False)


def train_step_loss(sd, cfg, audio_bft, lengths, targets, target_lengths, new_buffers=None):
    """forward + CTC + the loss scaling of exp/train.py:275 (one chunk per backward)."""
    out = forward(sd, cfg, audio_bft, lengths, training=True, new_buffers=new_buffers)
    loss = ctc_loss_sum(out['final_posteriors'], targets, out['length'], target_lengths)
    B, _, T = audio_bft.shape
    scaled = loss / (T * B) * 100.0
    return loss, scaled, out
