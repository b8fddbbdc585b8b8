"""MADGRAD on the MI355X: one fused kernel over FLAT f32 parameter / gradient / state buffers
(replaces lcasr/optim/madgrad.py:19-212, dense momentum branch, and the clip + GradScaler logic of
exp/train.py:46-61).

`FlatParams` re-points every parameter's storage (and its .grad) at one contiguous buffer per parameter group, so
  * the optimiser step, the global-norm clip and zero_grad are single launches,
  * data-parallel gradient all-reduce works on contiguous slices (parallel.GradSync) without bucket copies.
"""
from __future__ import annotations

import math
from typing import Iterable, List, Optional

import torch

from .hip import ops


class FlatParams:
    """Flatten a list of parameters into one f32 buffer (+ a gradient buffer of the same shape)."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        # frozen parameters (requires_grad=False) stay where they are: the reference optimiser skips parameters without a
        # gradient (madgrad.py:113-114), so they must neither be decayed nor moved by the fused step
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        assert self.params, 'no trainable parameters'
        dev = self.params[0].device
        assert all(p.dtype == torch.float32 and p.device == dev for p in self.params), 'f32 parameters on one device expected'
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4                     # keep every tensor 16-B aligned
        self.numel = n
        self.data = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, self.offsets):
            self.data[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.data[o:o + p.numel()].view(p.shape)
            p.grad = self.grad[o:o + p.numel()].view(p.shape)
        from . import functional as _Fn
        _Fn.clear_weight_cache()                              # bf16 shadows registered against the old storage are stale

    def zero_grad(self):
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):           # re-attach if something replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view(p.shape)

    def gather_stray_grads(self):
        """If autograd replaced a .grad tensor (instead of accumulating in place) copy it back into the flat buffer."""
        for p, o in zip(self.params, self.offsets):
            if p.grad is not None and p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                self.grad[o:o + p.numel()].copy_(p.grad.reshape(-1))
                p.grad = self.grad[o:o + p.numel()].view(p.shape)


class MADGRAD(torch.optim.Optimizer):
    """Drop-in for lcasr.optim.madgrad.MADGRAD(params, lr, momentum, weight_decay, eps) with a fused HIP step.

    step(max_norm=..., grad_scale=...) additionally performs torch.nn.utils.clip_grad_norm_(params, max_norm) and the
    GradScaler inf/nan skip inside the same launch sequence (no host sync)."""

    def __init__(self, params, lr: float = 1e-2, momentum: float = 0.9, weight_decay: float = 0, eps: float = 1e-6,
                 decouple_decay: bool = False):
        if momentum <= 0 or momentum >= 1:
            raise ValueError(f'Momentum {momentum} must be in (0,1) (the momentum == 0 branch is not implemented)')
        if lr < 0 or weight_decay < 0 or eps < 0:
            raise ValueError('lr, weight_decay and eps must be non-negative')
        if decouple_decay:
            raise NotImplementedError('decouple_decay is EXPERIMENTAL in the reference and unused by its configs')
        super().__init__(params, dict(lr=lr, eps=eps, momentum=momentum, weight_decay=weight_decay, decouple_decay=False))
        self.flat: List[FlatParams] = []
        for group in self.param_groups:
            fp = FlatParams(group['params'])
            self.flat.append(fp)
            group['_gss'] = torch.zeros_like(fp.data)
            group['_s'] = torch.zeros_like(fp.data)
            group['_x0'] = torch.zeros_like(fp.data)            # created by the kernel at the first applied step (k == 0: x0 := p),
                                                                 # like the reference's lazy state (madgrad.py:121-125)
        dev = self.flat[0].data.device
        self._k = torch.zeros((), dtype=torch.int64, device=dev)   # steps APPLIED so far, on the device (skipped steps do not count)
        self._sumsq = torch.zeros((), dtype=torch.float64, device=dev)
        self.last_sumsq: Optional[torch.Tensor] = None

    @property
    def k(self) -> int:
        """Steps applied so far (host value; syncs)."""
        return int(self._k)

    @k.setter
    def k(self, v: int) -> None:
        self._k.fill_(int(v))

    def zero_grad(self, set_to_none: bool = False):
        for fp in self.flat:
            fp.zero_grad()

    @torch.no_grad()
    def step(self, closure=None, max_norm: float = 0.0, grad_scale: float = 1.0):
        loss = closure() if closure is not None else None
        self._sumsq.zero_()
        for fp in self.flat:
            fp.gather_stray_grads()
            ops.sumsq_(fp.grad, self._sumsq)
        for group, fp in zip(self.param_groups, self.flat):
            ops.madgrad_step_(fp.data, fp.grad, group['_gss'], group['_s'], group['_x0'], None, self._sumsq, max_norm,
                              grad_scale, group['lr'], group['momentum'], group['eps'], group['weight_decay'], self._k)
        ops.madgrad_advance_(self._k, self._sumsq, grad_scale)
        self.last_sumsq = self._sumsq
        from . import functional as _Fn
        _Fn.bump_weight_epoch()                                # the kernel wrote the parameters without bumping their version counters
        return loss

    # ---- checkpoint compatibility with lcasr.optim.madgrad.MADGRAD (optim/madgrad.py:95-130) --------------------------
    # The reference keeps, per parameter, state {'grad_sum_sq', 's', 'x0'} (created lazily at the first step) and the step
    # counter as a 1-element long tensor under the non-parameter key 'k'.  Here that state lives in flat buffers; state_dict()
    # / load_state_dict() translate, so checkpoints written by either implementation resume in the other.
    _PRIVATE = ('params', '_gss', '_s', '_x0')

    def state_dict(self):
        state, groups, idx = {}, [], 0
        k = self.k
        for group, fp in zip(self.param_groups, self.flat):
            ids = []
            where = {id(p): o for p, o in zip(fp.params, fp.offsets)}
            for p in group['params']:                            # indices count every parameter of the group, like torch's
                o = where.get(id(p))
                if k > 0 and o is not None:                      # frozen parameters have no state (madgrad.py:113-114)
                    n = p.numel()
                    state[idx] = {name: group[key][o:o + n].view(p.shape).clone()
                                  for name, key in (('grad_sum_sq', '_gss'), ('s', '_s'), ('x0', '_x0'))}
                ids.append(idx); idx += 1
            g = {k_: v for k_, v in group.items() if k_ not in self._PRIVATE}
            g['params'] = ids
            groups.append(g)
        if k > 0:
            state['k'] = torch.tensor([k], dtype=torch.long)
        return {'state': state, 'param_groups': groups}

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        groups, state = state_dict['param_groups'], state_dict['state']
        if len(groups) != len(self.param_groups):
            raise ValueError('loaded state dict has a different number of parameter groups')
        idx = 0
        for saved, group, fp in zip(groups, self.param_groups, self.flat):
            if len(saved['params']) != len(group['params']):
                raise ValueError("loaded state dict contains a parameter group that doesn't match the size of optimizer's group")
            for k, v in saved.items():
                if k not in self._PRIVATE:
                    group[k] = v
            where = {id(p): o for p, o in zip(fp.params, fp.offsets)}
            for p in group['params']:
                st = state.get(idx, state.get(str(idx)))
                o = where.get(id(p))
                idx += 1
                if o is None:
                    continue
                n = p.numel()
                if st is not None:
                    for name, key in (('grad_sum_sq', '_gss'), ('s', '_s'), ('x0', '_x0')):
                        group[key][o:o + n].copy_(st[name].reshape(-1).to(group[key]))
                else:                                            # the reference would create it at the next step
                    group['_gss'][o:o + n].zero_(); group['_s'][o:o + n].zero_(); group['_x0'][o:o + n].copy_(fp.data[o:o + n])
        k = state.get('k', None)
        self.k = int(k.reshape(-1)[0]) if k is not None else 0

    def grad_norm(self) -> float:
        """Host value of the last step's global gradient norm (syncs)."""
        return math.sqrt(float(self._sumsq))
