"""BatchRenorm1d parameter/buffer container (lcasr/components/batchrenorm.py:9-110).

The arithmetic (batch statistics over all B*N positions, r/d clamps, EMA of the running buffers,
num_batches_tracked += 1) runs inside the conv-module HIP kernels (csrc/convmod.hip)."""
import torch
import torch.nn as nn


class BatchRenorm1d(nn.Module):
    def __init__(self, num_features: int, eps: float = 1e-3, momentum: float = 0.01, affine: bool = True):
        super().__init__()
        if not affine or eps != 1e-3 or momentum != 0.01:
            raise NotImplementedError('only the reference defaults (affine, eps=1e-3, momentum=0.01) are implemented')
        self.register_buffer('running_mean', torch.zeros(num_features, dtype=torch.float))
        self.register_buffer('running_std', torch.ones(num_features, dtype=torch.float))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))
        self.weight = nn.Parameter(torch.ones(num_features, dtype=torch.float))
        self.bias = nn.Parameter(torch.zeros(num_features, dtype=torch.float))
        self.affine, self.eps, self.momentum, self.step = affine, eps, momentum, 0

    @property
    def rmax(self):
        return (2 / 35000 * self.num_batches_tracked + 25 / 35).clamp_(1.0, 3.0)

    @property
    def dmax(self):
        return (5 / 20000 * self.num_batches_tracked - 25 / 20).clamp_(0.0, 5.0)
