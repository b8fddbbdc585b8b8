"""Sliding-window inference — mirror of lcasr/eval/utils.py:45-111 (`fetch_logits`), forward-only reuse of the HIP path.

`fetch_logits` keeps the reference's signature and window arithmetic (same quirks: buffer sized with `spec_n // 4`,
`overlap_ds = int(overlap / (u_len / ds_len))`, the loop stops one window after the first short one).  What changes is
the execution: the reference runs one window per forward and averages on the CPU (`# TODO: write batched version`,
utils.py:45); here all equally long windows go through the model in batches of `max_batch` and the overlap-average
(exp, accumulate, count, log) is two HIP kernels on the GPU (csrc/infer.hip).  `batched=False` reproduces the
reference's one-window-at-a-time order; both produce the same averages."""
from __future__ import annotations

from typing import List, Tuple

import torch

from .. import functional as Fn          # Fn.ops: the HIP op layer (tests swap it for the CPU kernel references)


def window_plan(spec_n: int, seq_len: int, overlap: int) -> List[Tuple[int, int]]:
    """(start, length) of every window the reference loop processes (utils.py:74-85)."""
    plan, last_ulen, kill_next = [], None, False
    for i in range(0, spec_n, seq_len - overlap):
        u_len = min(seq_len, spec_n - i)
        if kill_next:
            break
        if last_ulen is not None and u_len < last_ulen:
            kill_next = True
        last_ulen = u_len
        plan.append((i, u_len))
    return plan


@torch.no_grad()
def fetch_logits(args, model, spec: torch.Tensor, seq_len: int, overlap: int, tokenizer, use_tqdm=True, batched: bool = True,
                 max_batch: int = 16, return_numpy: bool = True):
    """Overlap-averaged log-probs (N, vocab+1) of a whole recording spec (1, F, T), windows of seq_len frames.

    args / tokenizer are used exactly as in the reference (config defaults for -1, vocab size).  Returns a numpy array like
    the reference unless return_numpy=False (then the GPU tensor)."""
    if spec.dim() != 3 or spec.shape[0] != 1:
        raise ValueError(f'spec must be (1, features, time), got {tuple(spec.shape)}')
    spec_n = spec.shape[-1]
    downsampling_factor = model.subsampling.subsampling_factor
    seq_len = seq_len if seq_len != -1 else args.config['audio_chunking']['size']
    if seq_len > spec_n:
        seq_len = spec_n
        overlap = 0
    else:
        overlap = overlap if overlap != -1 else args.config['audio_chunking']['overlap']
    assert overlap / downsampling_factor == overlap // downsampling_factor, 'Overlap must be a multiple of the downsampling factor'

    dev = next(model.parameters()).device
    C = tokenizer.vocab_size() + 1
    n_rows = spec_n // 4 + seq_len
    acc = torch.zeros(n_rows, C, dtype=torch.float32, device=dev)
    count = torch.zeros(n_rows, dtype=torch.float32, device=dev)
    spec = spec.to(dev)

    plan = window_plan(spec_n, seq_len, overlap)
    full = [p for p in plan if p[1] == seq_len]
    ragged = [p for p in plan if p[1] != seq_len]                # at most one: the last, shorter window
    logit_position, first = 0, True

    def place(lp, u_len):                                        # lp: (W, ds_len, C) of W consecutive windows
        nonlocal logit_position, first
        W, ds_len, _ = lp.shape
        overlap_ds = int(overlap / (u_len / ds_len))
        if not first:
            logit_position -= overlap_ds
        first = False
        stride = ds_len - overlap_ds
        Fn.ops.overlap_add_exp_(lp, acc, count, logit_position, stride if W > 1 else ds_len)
        logit_position += (W - 1) * stride + ds_len

    step = max_batch if batched else 1
    for k in range(0, len(full), step):
        grp = full[k:k + step]
        chunk = torch.stack([spec[0, :, s:s + seq_len] for s, _ in grp])
        out = model(chunk)['final_posteriors'].float().contiguous()
        place(out, seq_len)
    for s, u_len in ragged:
        out = model(spec[:, :, s:s + u_len].contiguous())['final_posteriors'].float().contiguous()
        place(out, u_len)

    logits = Fn.ops.overlap_finalize(acc, count, logit_position)   # rows with count != 0 are exactly the first logit_position
    return logits.cpu().numpy() if return_numpy else logits
