"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

numpy restatement of the dense, momentum != 0 branch of the reference MADGRAD
step (`/root/reference/lcasr/optim/madgrad.py:81-212`) plus the global-norm
clip that precedes it in `exp/train.py:46-61` (torch.nn.utils.clip_grad_norm_).
Pinned by tests/golden/madgrad.npz (generated from the imported reference by
oracle/make_golden.py).
"""
from __future__ import annotations

import math
import numpy as np


def clip_coef(grads, max_norm: float) -> float:
    """torch.nn.utils.clip_grad_norm_: coef = min(1, max_norm / (total_norm + 1e-6))."""
    total = math.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads))
    return min(1.0, max_norm / (total + 1e-6)), total


def madgrad_step(p, g, gss, s, x0, k: int, lr: float, momentum: float = 0.9, eps: float = 1e-6,
                 weight_decay: float = 0.0):
    """One step for one tensor; returns updated (p, gss, s).  madgrad.py:96-207."""
    if lr != 0.0:
        lr = lr + eps                                   # madgrad.py:100-101
    ck = 1 - momentum
    lamb = lr * math.pow(k + 1, 0.5)
    g = g.astype(np.float32)
    if weight_decay != 0:
        g = g + weight_decay * p
    gss = gss + lamb * g * g
    rms = np.cbrt(gss).astype(np.float32) + np.float32(eps)
    s = s + lamb * g
    z = x0 - s / rms
    p = p * (1 - ck) + ck * z
    return p.astype(np.float32), gss.astype(np.float32), s.astype(np.float32)
