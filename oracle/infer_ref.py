"""ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of the sliding-window inference path:

  fetch_logits   lcasr/eval/utils.py:45-111   overlap-average of exp(log-probs) over windows, then log
  greedy_decode  lcasr/decoding/greedy.py:9-23  argmax -> unique_consecutive -> drop blank

on top of the oracle forward (oracle/sconformer_ref.py).  Pinned by tests/golden/infer_tiny.npz, which
oracle/make_golden.py generates by running the reference's own fetch_logits / GreedyCTCDecoder on the reference model."""
from __future__ import annotations

import numpy as np
import torch

from . import sconformer_ref as O


def fetch_logits(sd, cfg, spec: torch.Tensor, seq_len: int, overlap: int, vocab_size: int) -> np.ndarray:
    """utils.py:45-111 with args.config defaults already resolved; spec (1,F,T)."""
    spec_n = spec.shape[-1]
    if seq_len > spec_n:                                        # utils.py:51-53
        seq_len, overlap = spec_n, 0
    assert overlap % 8 == 0                                     # utils.py:59 (downsampling factor 8)
    C = vocab_size + 1
    all_logits = torch.zeros(1, spec_n // 4 + seq_len, C)       # utils.py:64-65 (the // 4 is the reference's)
    logit_count = torch.zeros(1, spec_n // 4 + seq_len, C)
    pos, last_ulen, kill_next = 0, None, False
    for i in range(0, spec_n, seq_len - overlap):               # utils.py:73-104
        chunk = spec[:, :, i:i + seq_len]
        u_len = chunk.shape[-1]
        if kill_next:
            break
        if last_ulen is not None and u_len < last_ulen:
            kill_next = True
        last_ulen = u_len
        with torch.no_grad():
            lp = O.forward(sd, cfg, chunk, None, training=False)['final_posteriors']
        p = torch.exp(lp)
        ds_len = p.shape[-2]
        overlap_ds = int(overlap / (u_len / ds_len))
        if i != 0:
            pos -= overlap_ds
        logit_count[:, pos:pos + ds_len] += 1
        all_logits[:, pos:pos + ds_len] += p
        pos += ds_len
    keep = logit_count.sum(-1) != 0                             # utils.py:107-111
    out = all_logits[keep].reshape(1, -1, C) / logit_count[keep].reshape(1, -1, C)
    return torch.log(out).squeeze(0).numpy()


def greedy_decode(emission: np.ndarray, blank: int):
    """greedy.py:19-22 (decode=False path): list of token ids."""
    idx = np.argmax(emission, axis=-1)
    out, prev = [], None
    for i in idx.tolist():
        if i != prev:
            out.append(i)
        prev = i
    return [i for i in out if i != blank]
