"""SCConformerXL — drop-in for lcasr/models/sconformer_xl.py (same constructor kwargs, forward signature,
return dict, attribute names and state_dict keys), computing on the MI355X through the HIP blocks in
``lcasr_amd.functional``.

Construction order follows the reference (decoder, subsampler, then per layer conv / ff1 / ff2 / attention;
SURVEY.md A.2) so that ``torch.manual_seed(s)`` gives bit-identical initial weights.
"""
import torch
import torch.nn as nn
from torch.utils.checkpoint import checkpoint

from .. import functional as Fn
from ..components import convolution, decoder, fused_dense, subsampling, wrappers
from ..components.attention import Attention
from ..components.batchrenorm import BatchRenorm1d
from ..components.normalisation import LayerNorm, RMSNorm, get_norm_class
from ..components.rotary_emb import RotaryPositionalEmbedding
from .base import BaseModel

ConformerConvolution = convolution.ConformerConvolution
ConformerFeedForward = fused_dense.FusedMLP
ConvSubsampling = subsampling.ConvSubsampling
PreNorm, Scale = wrappers.PreNorm, wrappers.Scale
DEFAULT_NORM = RMSNorm


def _get_act(act: str):
    if act == 'silu':
        return nn.SiLU()
    raise NotImplementedError(f"subsampling_act='{act}' is not implemented on the HIP path (paper configs use 'silu')")


class SCConformerXL(BaseModel):
    def __init__(self, vocab_size=128, feat_in=80, subsampling='dw_striding', subsampling_factor=8,
                 subsampling_conv_channels=256, subsampling_act='silu', subsampling_norm_out=False, n_layers=6, d_model=768,
                 n_heads=6, head_dim=128, expansion_factor=4, dropout_ff=0.0, dropout_conv=0.0, dropout_attn=0.0,
                 checkpoint_every_n_layers=0, conv_kernel_size=9, conv_expansion_factor=1, decoder_norm=False,
                 use_rotary=False, rotary_interpolation_factor=1.0, learned_rotary=False, fourier_pos_enc=False,
                 self_conditioning=True, default_norm='layer_norm', sandwich_norm=False, bias_in_ff=False, transformer=False,
                 legasee_double_norm=True, **kwargs):
        super().__init__()
        if dropout_ff or dropout_conv or dropout_attn:
            raise NotImplementedError('dropout is 0.0 in every SConformerXL config; non-zero dropout is not implemented')
        if fourier_pos_enc:
            raise NotImplementedError('fourier_pos_enc is not on the hot path (SURVEY.md §2 #21)')
        if transformer:
            raise NotImplementedError('transformer=True (no conv module) is not implemented')
        self.feat_in, self.n_layers, self.d_model, self.n_heads, self.head_dim = feat_in, n_layers, d_model, n_heads, head_dim
        self.expansion_factor = expansion_factor                      # stored, never used (reference: dead key)
        self.conv_kernel_size, self.conv_expansion_factor = conv_kernel_size, conv_expansion_factor
        self.rotary_interpolation_factor, self.learned_rotary = rotary_interpolation_factor, learned_rotary
        self.self_conditioning, self.sandwich_norm, self.bias_in_ff = self_conditioning, sandwich_norm, bias_in_ff
        self.transformer, self.legasee_double_norm = transformer, legasee_double_norm
        self.checkpoint_subsampling = kwargs.get('checkpoint_subsampling', False)
        accepted_acts = ['silu', 'relu', 'gelu', 'none']
        assert subsampling_act in accepted_acts, f'subsampling_act must be one of {accepted_acts} (got {subsampling_act})'
        norm_cls = get_norm_class(default_norm)
        self.flash_attn = kwargs.get('flash_attn', True)
        self.checkpoint_every_n_layers = checkpoint_every_n_layers
        self.dropout_ff, self.dropout_conv, self.dropout_attn = dropout_ff, dropout_conv, dropout_attn
        self.subsampling_mode, self.subsampling_factor = subsampling, subsampling_factor
        self.subsampling_conv_channels = subsampling_conv_channels if subsampling_conv_channels != -1 else d_model
        self.whitelist_weight_decay_modules = (nn.LayerNorm, RMSNorm, LayerNorm, BatchRenorm1d, nn.GroupNorm)
        self.blacklist_weight_decay_modules = (nn.Linear, ConformerFeedForward, nn.Conv1d, nn.Conv2d, RotaryPositionalEmbedding)
        self.decoder_norm, self.use_rotary = decoder_norm, use_rotary

        self.rotary_pos_emb = None
        if use_rotary:
            self.rotary_pos_emb = RotaryPositionalEmbedding(dim=head_dim, base=kwargs.get('rotary_base_freq', 10000),
                                                            learned_freq=learned_rotary,
                                                            rotary_interpolation_factor=rotary_interpolation_factor)
        self.fourier_pos_enc = nn.Identity()
        self.decoder = decoder.ASRLinearSCDecoder(d_model=d_model, vocab_size=vocab_size, norm=decoder_norm, norm_fn=norm_cls, **kwargs)
        self.subsampling = ConvSubsampling(subsampling=subsampling, conv_channels=self.subsampling_conv_channels,
                                           activation=_get_act(subsampling_act), subsampling_factor=subsampling_factor,
                                           feat_in=feat_in, feat_out=d_model, norm_out=subsampling_norm_out, default_norm=norm_cls)
        self.layers = nn.ModuleList()
        for i in range(n_layers):
            self.layers.append(ConformerLayer(d_model=d_model, conv_kernel_size=conv_kernel_size, expansion_factor=expansion_factor,
                                              dropout_ff=dropout_ff, dropout_conv=dropout_conv, dropout_attn=dropout_attn,
                                              layer_idx=i, total_layers=n_layers, head_dim=head_dim, n_heads=n_heads,
                                              default_norm=norm_cls, sandwich_norm=sandwich_norm, bias_in_ff=bias_in_ff,
                                              transformer=transformer, conv_expansion_factor=conv_expansion_factor, **kwargs))

    def forward(self, audio_signal, length=None, cached_kvs=None, cached_kv_lengths=None, return_logits=False, ctc_targets=None):
        """audio_signal: (B, feat_in, T) f32/bf16 on the GPU; length: (B,) ints or None.
        Returns {'final_posteriors': (B, N, V+1) f32 log-probs (logits if return_logits), 'length': (B,) int32}.
        ctc_targets = (targets (B, S), target_lengths (B,)) (an extension of the reference signature, used by the training step):
        the head and the CTC loss run as one operator and the dict holds 'ctc_nll' (B,) - what
        CTCLoss(blank=vocab_size, reduction='none')(final_posteriors.transpose(0, 1), targets, length, target_lengths) would give -
        instead of the posteriors, which are then never written ('final_posteriors' is None)."""
        if cached_kvs is not None:
            raise NotImplementedError('cached_kvs is vestigial in the reference (SURVEY.md fact 8) and not supported')
        Fn.ops.require_gpu(audio_signal, 'audio_signal')
        Fn.refresh_weight_shadows()
        dec = self.decoder
        B, _, T = audio_signal.shape
        dev = audio_signal.device
        if length is None:
            length = torch.tensor([T] * B, device=dev)
        x, length = self.subsampling(torch.transpose(audio_signal, 1, 2), lengths=length)      # (B,N,d) f32
        N = x.size(1)
        len_host = length.tolist()                                    # one D2H sync (reference: length.max()==length.min())
        rotary = None
        if self.use_rotary:
            n_rot = max(len_host)
            if n_rot < N:
                raise RuntimeError('rotary table shorter than the padded sequence (the reference fails the same way)')
            cos, sin = self.rotary_pos_emb.tables(n_rot, dev)
            rotary = (cos[:N], sin[:N]) if n_rot > N else (cos, sin)
        lengths_dev = None if max(len_host) == min(len_host) else length.to(device=dev, dtype=torch.int32).contiguous()

        extra = 1 if self.legasee_double_norm else 0
        h = None
        dec.hold_padded()                                             # class padding for vocab_size + 1 not a multiple of 16 (no-op otherwise)
        for lth, layer in enumerate(self.layers):
            last = lth == len(self.layers) - 1
            sc = not last and self.self_conditioning
            # the decoder norm of the self-conditioning step - after the last layer: the head's norm(s) - is applied by the layer
            # together with its norm_out (one pass)
            post = dec.post_norm_spec(x.shape[-1], layer.norm_out.mode, extra if last else 0) if (sc or last) else None
            h = None
            if self.checkpoint_every_n_layers > 0 and lth % self.checkpoint_every_n_layers == 0:
                if post is None:
                    x = checkpoint(layer, x, lengths_dev, rotary, use_reentrant=False)
                else:                                     # both results are outputs of the checkpointed region
                    def run(x_, lengths_, rotary_, layer=layer, post=post):
                        spec = dict(post)
                        return layer(x_, lengths_, rotary_, post_norm=spec), spec['h']
                    x, h = checkpoint(run, x, lengths_dev, rotary, use_reentrant=False)
            else:
                x = layer(x, lengths_dev, rotary, post_norm=post)
                h = None if post is None else post.get('h')
            if sc:
                x = dec.self_condition(x, prenormed=h)
        if ctc_targets is not None:
            out = {'final_posteriors': None, 'length': length,
                   'ctc_nll': dec.ctc_nll(x, ctc_targets[0], length, ctc_targets[1], extra_norms=extra, prenormed=h)}
        else:
            out = {'final_posteriors': dec(x, logits=return_logits, extra_norms=extra, prenormed=h), 'length': length}
        dec.release_padded()
        if self.training and self.rotary_pos_emb is not None:
            self.rotary_pos_emb.reset_if_needed()
        return out


class ConformerLayer(nn.Module):
    def __init__(self, d_model, conv_kernel_size, dropout_ff, dropout_conv, dropout_attn, layer_idx, total_layers, head_dim,
                 n_heads, default_norm=DEFAULT_NORM, sandwich_norm=False, bias_in_ff=True, transformer=False,
                 conv_expansion_factor=1, **kwargs):
        super().__init__()
        self.d_model, self.conv_kernel_size, self.layer_idx, self.total_layers = d_model, conv_kernel_size, layer_idx, total_layers
        self.sandwich_norm, self.bias_in_ff, self.trasformer = sandwich_norm, bias_in_ff, transformer
        ckpt = kwargs.get('ff_checkpoint_lvl', 0)
        self.conv = PreNorm(d_model=d_model, fn=ConformerConvolution(d_model=d_model, kernel_size=conv_kernel_size,
                                                                     norm_type=kwargs.get('conv_norm', 'batch_renorm'),
                                                                     exp_factor=conv_expansion_factor), norm=default_norm)
        self.do_conv = nn.Dropout(dropout_conv)
        self.ff1 = Scale(0.5, PreNorm(d_model=d_model, fn=ConformerFeedForward(d_model, bias1=bias_in_ff, bias2=bias_in_ff, checkpoint_lvl=ckpt),
                                      norm=default_norm, sandwich_norm=sandwich_norm))
        self.ff2 = Scale(0.5, PreNorm(d_model=d_model, fn=ConformerFeedForward(d_model, bias1=bias_in_ff, bias2=bias_in_ff, checkpoint_lvl=ckpt),
                                      norm=default_norm, sandwich_norm=sandwich_norm))
        self.do_ff = nn.Dropout(dropout_ff)
        self.attend = PreNorm(d_model=d_model, fn=Attention(n_feats=d_model, head_dim=head_dim, n_heads=n_heads, dropout=dropout_attn,
                                                            bias=False, layer_idx=layer_idx, **kwargs), norm=default_norm)
        self.attn_norm_out = lambda x: x
        self.do_attn_out = nn.Dropout(min(dropout_ff, 0.1))
        self.norm_out = default_norm(d_model)

    def forward(self, x, lengths=None, rotary=None, post_norm=None):
        """sconformer_xl.py:346-372 with every branch fused with its residual add.
        x (B,N,d) f32; lengths int32 (B,) on device when the batch is ragged, else None; rotary = (cos, sin) tables.
        post_norm (optional, ASRLinearSCDecoder.post_norm_spec): the LayerNorm the consumer of this layer's output applies first;
        it is computed in the same pass as norm_out and left in post_norm['h'] (bf16)."""
        x = self.ff1(x, residual=True)
        x = self.attend(x, residual=True, lengths=lengths, rotary=rotary)
        x = self.conv(x, residual=True, lengths=lengths)
        x = self.ff2(x, residual=True)
        if post_norm is not None:
            nw, nb = self.norm_out.norm_params()
            shape = x.shape
            y, h = Fn.norm2(x.reshape(-1, shape[-1]), nw, nb, post_norm['w'], post_norm['b'], self.norm_out.eps, post_norm['eps'],
                            post_norm.get('twice', False))
            post_norm['h'] = h.view(shape)
            return y.view(shape)
        return self.norm_out(x)
