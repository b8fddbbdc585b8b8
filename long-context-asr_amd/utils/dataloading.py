"""Chunking of long recordings for the training driver — mirror of `chunk_spectogram` (lcasr/utils/dataloading.py:14-25)
and of the per-recording chunk bookkeeping that exp/train.py does inline (train.py:174-201): which samples of the batch are
still alive at chunk ix (`selection_mask`), and how many frames of the chunk are real for each of them (`audio_lengths`).

The corpus loaders / tokeniser / word-timestamp text chunking of the reference file are data plumbing and out of scope
(SURVEY §2 row 11); the benchmark and the tests feed synthetic mel through exactly this chunk plan."""
from __future__ import annotations

from typing import Dict, List

import torch


def chunk_spectogram(spec: torch.Tensor, chunk_size: int, chunk_overlap: int) -> List[torch.Tensor]:
    """spec (batch, features, time) -> views of chunk_size frames every chunk_size - chunk_overlap frames (last one ragged)."""
    assert len(spec.shape) == 3, "Audio must be 3D i.e. (batch, features, time)"
    assert chunk_size > chunk_overlap, "chunk_size must be greater than chunk_overlap"
    hop = chunk_size - chunk_overlap
    return [spec[:, :, i:i + chunk_size] for i in range(0, spec.shape[2], hop)]


def plan_chunks(audio_chunks: List[torch.Tensor], audio_lengths: torch.Tensor, chunk_overlap: int) -> List[Dict]:
    """The bookkeeping of train.py:174-201 (without the text side): for every chunk, the rows of the batch that still have
    audio (`selection_mask`, a recording drops out once the frames consumed so far exceed its length), the chunk restricted to
    them (`audio`), their valid frame counts inside the chunk (`audio_lengths`) and the frames consumed before it
    (`cur_culm_lengths`)."""
    consumed = torch.zeros_like(audio_lengths)
    plan = []
    for ix, el in enumerate(audio_chunks):
        alive = ~(consumed > audio_lengths)
        cur, cur_consumed = el[alive], consumed[alive]
        width = cur.shape[-1]
        pad = (cur_consumed + width - audio_lengths[alive] - chunk_overlap).clamp(0)       # frames past the recording's end
        plan.append({'audio': cur, 'audio_lengths': width - pad, 'selection_mask': alive, 'cur_culm_lengths': cur_consumed})
        consumed[alive] += width - (chunk_overlap if ix != 0 else 0)
    return plan
