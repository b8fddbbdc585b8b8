"""Norm modules of the hot path (parameter containers + HIP forward).

LayerNorm  <-> torch.nn.LayerNorm / apex FusedLayerNorm (sconformer_xl.py:14-17): params `weight`, `bias`, eps 1e-5.
RMSNorm    <-> lcasr/components/normalisation.py:6-47: param `scale`, y = scale * x / (||x||_2 d^-1/2 + 1e-8).
"""
import torch
import torch.nn as nn

from .. import functional as Fn


class LayerNorm(nn.Module):
    mode = 'layer_norm'

    def __init__(self, d_model, eps=1e-5):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d_model))
        self.bias = nn.Parameter(torch.zeros(d_model))
        self.eps = eps

    def norm_params(self):
        return self.weight, self.bias

    def forward(self, x, out_dtype=None):
        return Fn.norm(x, self.weight, self.bias, self.mode, self.eps, out_dtype or x.dtype)


class RMSNorm(nn.Module):
    mode = 'rms_norm'

    def __init__(self, d_model, p=-1., eps=1e-8, bias=False):
        super().__init__()
        if bias or (0. <= p <= 1.):
            raise NotImplementedError('partial / biased RMSNorm is not on the SConformerXL hot path')
        self.eps, self.d, self.p, self.bias = eps, d_model, p, bias
        self.scale = nn.Parameter(torch.ones(d_model))

    def norm_params(self):
        return self.scale, None

    def forward(self, x, out_dtype=None):
        return Fn.norm(x, self.scale, None, self.mode, self.eps, out_dtype or x.dtype)


def get_norm_class(default_norm: str):
    if default_norm == 'rms_norm':
        return RMSNorm
    if default_norm == 'layer_norm':
        return LayerNorm
    raise ValueError(f'default_norm must be one of [rms_norm, layer_norm] (got {default_norm})')
