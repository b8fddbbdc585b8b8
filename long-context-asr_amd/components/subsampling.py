"""ConvSubsampling, 'dw_striding' mode (lcasr/components/subsampling.py:165-428, calc_length :557-567).

`conv` is the same nn.Sequential of Conv2d/SiLU modules as the reference (state_dict keys conv.{0,2,3,5,6});
they are parameter containers: the math runs channels-last in csrc/subsample.hip + GEMMs."""
import math

import torch
import torch.nn as nn

from .. import functional as Fn
from .normalisation import LayerNorm


def calc_length(lengths, all_paddings, kernel_size, stride, ceil_mode, repeat_num=1):
    """subsampling.py:557-567 (float floor arithmetic, then int)."""
    add_pad = all_paddings - kernel_size
    for _ in range(repeat_num):
        lengths = torch.div(lengths.to(dtype=torch.float) + add_pad, stride) + 1.0
        lengths = torch.ceil(lengths) if ceil_mode else torch.floor(lengths)
    return lengths.to(dtype=torch.int)


class ConvSubsampling(nn.Module):
    def __init__(self, subsampling, subsampling_factor, feat_in, feat_out, conv_channels, subsampling_conv_chunking_factor=1,
                 activation=None, is_causal=False, norm_out=False, default_norm=LayerNorm):
        super().__init__()
        if subsampling != 'dw_striding':
            raise NotImplementedError("only subsampling='dw_striding' is on the SConformerXL hot path (SURVEY.md §2 #2)")
        if subsampling_factor != 8:
            raise NotImplementedError('only subsampling_factor=8 (three stride-2 stages) is implemented')
        if is_causal:
            raise NotImplementedError('causal subsampling is not on the hot path')
        if subsampling_factor % 2 != 0:
            raise ValueError('Sampling factor should be a multiply of 2!')
        if not isinstance(activation, nn.SiLU):
            raise NotImplementedError("only subsampling_act='silu' is implemented")
        self._subsampling, self._conv_channels, self._feat_in, self._feat_out = subsampling, conv_channels, feat_in, feat_out
        self.has_norm_out = norm_out
        if norm_out:
            self.norm_out = default_norm(feat_out)
        self._sampling_num = int(math.log(subsampling_factor, 2))
        self.subsampling_factor = subsampling_factor
        self.is_causal = is_causal
        self._stride, self._kernel_size, self._ceil_mode = 2, 3, False
        self._left_padding = self._right_padding = 1
        C = conv_channels
        layers = [nn.Conv2d(1, C, 3, 2, 1), activation]
        for _ in range(self._sampling_num - 1):
            layers += [nn.Conv2d(C, C, 3, 2, 1, groups=C), nn.Conv2d(C, C, 1, 1, 0), activation]
        out_length = calc_length(torch.tensor(feat_in, dtype=torch.float), 2, 3, 2, False, self._sampling_num)
        self.out = nn.Linear(C * int(out_length), feat_out, bias=norm_out)
        self.conv = nn.Sequential(*layers)

    def forward(self, x, lengths):
        """x: (B, T, feat) as in the reference (the model transposes before the call); returns (B,N,d) f32, lengths."""
        lengths = calc_length(lengths, 2, 3, 2, False, self._sampling_num)
        c = self.conv
        y = Fn.subsample(x.transpose(1, 2), c[0].weight, c[0].bias, c[2].weight, c[2].bias, c[3].weight, c[3].bias,
                         c[5].weight, c[5].bias, c[6].weight, c[6].bias, self.out.weight, self.out.bias)
        if self.has_norm_out:
            y = self.norm_out(y)
        return y, lengths
