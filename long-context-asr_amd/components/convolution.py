"""ConformerConvolution (lcasr/components/convolution.py:41-124): pw Conv1d d->2d, GLU, depthwise Conv1d k,
BatchRenorm1d, SiLU, pw Conv1d d->d.  Conv1d modules are parameter containers (identical init/state_dict);
the math runs token-major in csrc/convmod.hip + GEMMs."""
import torch.nn as nn

from .. import functional as Fn
from .batchrenorm import BatchRenorm1d


class ConformerConvolution(nn.Module):
    def __init__(self, d_model, kernel_size, norm_type='batch_renorm', exp_factor=1, **kwargs):
        super().__init__()
        assert (kernel_size - 1) % 2 == 0
        if norm_type != 'batch_renorm':
            raise NotImplementedError("only conv_norm='batch_renorm' (the reference default) is implemented")
        if exp_factor != 1:
            raise NotImplementedError('conv_expansion_factor != 1 is not used by any SConformerXL config')
        self.d_model = d_model
        inner_dim = int(d_model * exp_factor)
        self.pointwise_conv1 = nn.Conv1d(d_model, inner_dim * 2, kernel_size=1, stride=1, padding=0, bias=True)
        self.depthwise_conv = nn.Conv1d(inner_dim, inner_dim, kernel_size=kernel_size, stride=1,
                                        padding=(kernel_size - 1) // 2, groups=inner_dim, bias=True)
        self.conv_kernel_size = kernel_size
        self.conv_padding = (kernel_size - 1) // 2
        self.batch_norm = BatchRenorm1d(inner_dim)
        self.use_fft_conv = False
        self.pointwise_conv2 = nn.Conv1d(inner_dim, d_model, kernel_size=1, stride=1, padding=0, bias=True)

    def forward(self, x, pad_mask=None, **kwargs):
        """The reference's module-level call (convolution.py:103-124) on an already normalised x (B,N,d); pad_mask (B,N) bool,
        True = padded (the suffix mask of sconformer_xl.py:207)."""
        import torch
        lengths = None if pad_mask is None else (~pad_mask).sum(-1).to(dtype=torch.int32).contiguous()
        return self.forward_prenorm(x, None, False, lengths=lengths)

    def forward_prenorm(self, x, norm, residual, lengths=None, **_):
        """norm None: x is already normalised (the module-level `forward`)."""
        B, N, _d = x.shape
        nw, nb = norm.norm_params() if norm is not None else (None, None)
        mode, eps = (norm.mode, norm.eps) if norm is not None else ('none', 0.0)
        bn = self.batch_norm
        y = Fn.conv_block(x.reshape(B * N, -1), nw, nb, self.pointwise_conv1.weight, self.pointwise_conv1.bias,
                          self.depthwise_conv.weight, self.depthwise_conv.bias, bn.weight, bn.bias, bn.running_mean,
                          bn.running_std, bn.num_batches_tracked, self.pointwise_conv2.weight, self.pointwise_conv2.bias,
                          lengths, B, N, self.training, mode, eps, residual)
        return y.view(B, N, -1)
