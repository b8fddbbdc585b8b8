"""Training-step counterpart of the hot loop in exp/train.py:212-293 (one chunk == one optimiser step, the paper
schedule `backprop_every: 1`), on synthetic mel batches, with optional single-node data parallelism.

    out  = model(audio, length)                                           # train.py:236
    loss = CTCLoss(blank=V, 'sum')(out.transpose(0,1), txt, out_len, txt_len)   # train.py:104,249
    (loss / (chunk_size * batch_size) * 100).backward()                  # train.py:275  (GLOBAL batch under DDP)
    clip_grad_norm_(0.8); MADGRAD.step(); zero_grad()                     # train.py:46-61
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from .losses import CTCLoss
from . import functional as Fn
from .optim import MADGRAD
from .parallel import GradSync


def synthetic_batch(B: int, T: int, vocab_size: int, seed: int = 0, device='cuda', dtype=torch.float32):
    """mel ~ N(0,1) (B,80,T), lengths = T, targets uniform in [0,V), S = N/4 tokens per sample."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 80, T, generator=g).to(dtype)
    N = ((T - 1) // 2 + 1 - 1) // 2 + 1
    N = (N - 1) // 2 + 1
    S = max(N // 4, 1)
    targets = torch.randint(0, vocab_size, (B, S), generator=g)
    return (x.to(device), torch.full((B,), T, dtype=torch.long, device=device), targets.to(device),
            torch.full((B,), S, dtype=torch.long, device=device))


class Trainer:
    def __init__(self, model, lr: float = 3e-3, clip_value: float = 0.8, global_batch: Optional[int] = None,
                 bucket_bytes: int = 64 << 20, fused_loss: bool = True):
        """fused_loss: run the decoder head and the CTC loss as one operator (model(..., ctc_targets=...)); False keeps the
        reference's two calls (posteriors, then CTCLoss) - same loss and gradients, 25 GB more HBM traffic per step at B = 128."""
        self.model = model
        self.fused_loss = fused_loss and os.environ.get('SCONF_FUSED_LOSS', '1') != '0'       # env switch: A/B runs
        self.opt = MADGRAD(model.parameters(), lr=lr)
        self.ctc = CTCLoss(blank=model.decoder.num_classes - 1, reduction='sum')
        self.clip_value = clip_value
        self.global_batch = global_batch
        fp = self.opt.flat[0]
        self.sync = GradSync(fp.params, fp.grad, fp.offsets, bucket_bytes)

    def step(self, audio, lengths, targets, target_lengths, norm_frames: Optional[int] = None, norm_batch: Optional[int] = None):
        """One optimiser step; returns the (device) summed CTC loss of this rank's shard.
        The loss is scaled by 100 / (norm_frames * norm_batch): the reference divides by the CONSTANT chunk_size * batch_size
        (exp/train.py:275), so ragged last chunks and shrunken batches weigh less; defaults: this batch's width / batch."""
        B, _, T = audio.shape
        if self.fused_loss:                              # head + log_softmax + CTC as one operator: the posteriors are never written
            loss = self.model(audio, length=lengths, ctc_targets=(targets, target_lengths))['ctc_nll'].sum()
        else:
            out = self.model(audio, length=lengths)
            loss = self.ctc(out['final_posteriors'].transpose(0, 1), targets, out['length'], target_lengths)
        T = norm_frames or T
        gb = norm_batch or self.global_batch or B * self.sync.world
        # parameter gradients are accumulated by the backward kernels straight into the flat gradient buffer (no per-
        # parameter add / zero-fill launches); valid because this is a plain .backward() into pre-attached .grad views
        Fn.set_direct_grad(True); Fn.set_grad_ready_hook(self.sync.on_grad_ready)
        try:
            (loss / (T * gb) * 100).backward()
        finally:
            Fn.set_direct_grad(False); Fn.set_grad_ready_hook(None)
        self.sync.finish()
        self.opt.step(max_norm=self.clip_value)
        self.opt.zero_grad()
        return loss.detach()

    def step_without_data(self):
        """This rank has nothing to train in a step the other ranks do take (its recordings have run out): contribute zero
        gradients to the same all-reduces and apply the same optimiser step, so that every rank issues the same collectives and
        the PARAMETERS stay identical.  BatchRenorm buffers do not move on this rank (no forward ran): `train_recording` re-broadcasts
        rank 0's buffers after a batch of recordings with unequal participation."""
        self.sync.finish()
        self.opt.step(max_norm=self.clip_value)
        self.opt.zero_grad()
        return None

    def train_recording(self, audio, audio_lengths, chunk_size: int, chunk_overlap: int, targets_for_chunk):
        """One batch of long recordings, chunk by chunk (exp/train.py:174-293 with backwards_every = backprop_every = 1, the
        paper configs): audio (B, F, T_total), audio_lengths (B,) frames.  Recordings that have run out drop out of the batch
        (`selection_mask`), the last chunk of each is ragged (per-sample lengths -> the varlen attention / masking path).
        targets_for_chunk(ix, chunk) -> (targets (B', S), target_lengths (B',)) for the rows alive in chunk ix (the text side
        of the reference's chunker is data plumbing; the benchmark feeds synthetic targets).  Returns the per-chunk losses."""
        from .utils.dataloading import chunk_spectogram, plan_chunks
        import torch.distributed as dist
        plan = plan_chunks(chunk_spectogram(audio, chunk_size, chunk_overlap), audio_lengths, chunk_overlap)
        world = self.sync.world
        nominal = audio.shape[0] * world
        # what this rank can train on, chunk by chunk (audio stays a view until its step)
        works = []
        for ix, c in enumerate(plan):
            work = None
            keep = c['audio_lengths'] > 0                       # an exactly exhausted recording: the reference feeds a zero-length
            if not bool(keep.all()):                            # row and lands in its NaN-skip branch; here the row is dropped
                c = {k: (v[keep] if torch.is_tensor(v) and v.shape[:1] == keep.shape else v) for k, v in c.items()}
            if c['audio'].shape[0] > 0:
                tg, tl = targets_for_chunk(ix, c)
                if int(tl.max()) > 0:                           # train.py:186-187: nothing to align in this chunk -> skipped
                    work = (c['audio'], c['audio_lengths'], tg, tl)
            works.append(work)
        n_chunks = len(works)
        ranks_with_work = None
        if world > 1:
            # The reference has no data parallelism; here every optimiser step is a set of collectives, so all ranks must take
            # the same number of steps although their recordings differ in length: agree ONCE per recording batch on the longest
            # plan (MAX) and on how many ranks have something to train in each chunk (SUM of the whole flag vector) - two host
            # syncs per batch of recordings instead of two per chunk.  A rank without work in a chunk another rank trains on
            # runs `step_without_data`.
            t = torch.tensor([n_chunks], dtype=torch.int64, device=audio.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.sync.group)
            n_chunks = int(t)
            flags = torch.zeros(n_chunks, dtype=torch.int64, device=audio.device)
            if works:
                flags[:len(works)] = torch.tensor([int(w is not None) for w in works], dtype=torch.int64, device=audio.device)
            dist.all_reduce(flags, op=dist.ReduceOp.SUM, group=self.sync.group)
            ranks_with_work = flags.tolist()
        losses = []
        for ix in range(n_chunks):
            work = works[ix] if ix < len(works) else None
            if world > 1:
                if ranks_with_work[ix] == 0:
                    continue                                    # no rank has work: everyone skips together
                if work is None:
                    self.step_without_data()
                    continue
            elif work is None:
                continue
            losses.append(self.step(work[0].contiguous(), *work[1:], norm_frames=chunk_size, norm_batch=nominal))
        if world > 1 and any(0 < n < world for n in ranks_with_work):
            # Ranks that sat a step out did not run its forward, so their BatchRenorm running statistics (and step counters, on
            # which the rmax / dmax ramps depend) have fallen behind.  Rank 0's buffers are authoritative - what DistributedDataParallel
            # does on every forward (broadcast_buffers) - and are re-broadcast whenever participation was unequal, so the replicas
            # (and whichever rank writes the checkpoint) stay identical.
            from .parallel import broadcast_module_buffers
            broadcast_module_buffers(self.model, 0, self.sync.group)
        return losses

