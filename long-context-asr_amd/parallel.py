"""Single-node data parallelism for the training driver: one process per GPU, RCCL (torch.distributed backend
"nccl") all-reduce of gradients over xGMI, overlapped with backward.

The reference has no distributed code at all; this is the new capability the north star asks for.
Design for MI355X: gradients already live in ONE flat f32 buffer (optim.FlatParams), so a bucket is a contiguous
slice — no bucket copies.  Buckets are cut in parameter-registration order; backward produces gradients in roughly
the reverse order, so the LAST slices complete first and their all-reduce overlaps the rest of the backward.
xGMI is point-to-point (7 links x ~153 GB/s): ring all-reduce is per-link bound, so buckets are large (default
64 MiB) — few, large collectives.  The loss normaliser stays the GLOBAL constant T*B_global (exp/train.py:275), so
SUM (not mean) reproduces the single-GPU gradient; BatchRenorm statistics stay rank-local (standard DDP behaviour).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


def broadcast_module_state(module: torch.nn.Module, src: int = 0) -> None:
    """Make parameters and buffers (BatchRenorm running stats) identical on every rank."""
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)


class GradSync:
    """Bucketed, backward-overlapped gradient all-reduce over a flat gradient buffer."""

    def __init__(self, params: List[torch.nn.Parameter], flat_grad: torch.Tensor, offsets: List[int],
                 bucket_bytes: int = 64 << 20, group: Optional[dist.ProcessGroup] = None):
        self.flat_grad, self.group = flat_grad, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.buckets = []                                   # (start, end, n_params)
        self.param_bucket = {}
        cap = max(1, bucket_bytes // 4)
        start, count = 0, 0
        ends = [o + p.numel() for p, o in zip(params, offsets)]
        for i, (p, o) in enumerate(zip(params, offsets)):
            self.param_bucket[i] = len(self.buckets)
            count += 1
            last = i == len(params) - 1
            if ends[i] - start >= cap or last:
                self.buckets.append((start, flat_grad.numel() if last else ends[i], count))
                start, count = ends[i], 0
        self.pending = [0] * len(self.buckets)
        self.handles = []
        self._hooks = []
        self._index = {id(p): i for i, p in enumerate(params)}
        self._seen, self._expect = {}, None                 # direct-write counts of this backward / of the first one
        if self.world > 1:
            for i, p in enumerate(params):
                if p.requires_grad:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
        self.reset()

    def on_grad_ready(self, p: torch.nn.Parameter) -> None:
        """A backward kernel has accumulated this parameter's gradient straight into the flat buffer
        (functional.set_direct_grad): autograd's post-accumulate hook will not fire for it, so count it here."""
        i = self._index.get(id(p))
        if i is None or self.world == 1:
            return
        # A parameter used by several blocks (the decoder inside every self-conditioning layer) is written several times
        # per backward.  The first backward only counts the writes; later ones release the parameter on its last write.
        self._seen[i] = self._seen.get(i, 0) + 1
        if self._expect is not None and self._seen[i] == self._expect.get(i, 0):
            self._ready(i)

    def reset(self):
        self.pending = [n for (_, _, n) in self.buckets]
        self.handles = []
        self._seen = {}

    def _ready(self, i):
        b = self.param_bucket[i]
        self.pending[b] -= 1
        if self.pending[b] == 0:
            s, e, _ = self.buckets[b]
            self.handles.append(dist.all_reduce(self.flat_grad[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _make_hook(self, i):
        def hook(_p):
            self._ready(i)
        return hook

    def finish(self):
        """Wait for the in-flight all-reduces; reduce any bucket whose hooks did not all fire (unused parameters)."""
        if self.world > 1:
            if self._expect is None:
                self._expect = dict(self._seen)
            elif any(n > self._expect.get(i, 0) for i, n in self._seen.items()):
                raise RuntimeError('GradSync: a parameter received more direct gradient writes than in the first backward; '
                                   'its bucket may have been reduced early (the graph must not change between steps)')
            for b, n in enumerate(self.pending):
                if n > 0:
                    s, e, _ = self.buckets[b]
                    self.handles.append(dist.all_reduce(self.flat_grad[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            for h in self.handles:
                h.wait()
        self.reset()

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
