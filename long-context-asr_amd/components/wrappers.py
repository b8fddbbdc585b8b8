"""PreNorm / Scale (lcasr/components/wrappers.py:5-28): same module tree and state_dict keys.

Module-level calls (``layer.ff1(x)``, ``layer.attend(x, ...)``, ``layer.conv(x, ...)``) return the residual
BRANCH, like the reference.  ConformerLayer.forward passes ``residual=True``: the same modules then return
``x + branch`` from the fused branch+residual block (forward hooks on them still fire)."""
import torch.nn as nn

from .normalisation import RMSNorm


class PreNorm(nn.Module):
    def __init__(self, d_model, fn, norm=RMSNorm, sandwich_norm=False):
        super().__init__()
        if sandwich_norm:
            raise NotImplementedError('sandwich_norm is not used by any SConformerXL config (SURVEY.md §8)')
        self.norm = norm(d_model)
        self.fn = fn
        self.sandwich_norm = sandwich_norm

    def forward(self, x, residual=False, **kwargs):
        # the wrapped module fuses this PreNorm's norm into its first kernel (and, with residual=True, `x +` into its last)
        return self.fn.forward_prenorm(x, self.norm, residual=residual, **kwargs)


class Scale(nn.Module):
    def __init__(self, scale, fn):
        super().__init__()
        self.scale = scale
        self.fn = fn

    def forward(self, x, **kwargs):
        return self.fn(x, scale=self.scale, **kwargs)
