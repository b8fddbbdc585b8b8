"""Greedy CTC decoding — mirror of lcasr/decoding/greedy.py:4-23 (same class, constructor and call signature).

The row argmax runs on the GPU (csrc/infer.hip, sconf_argmax_rows); the collapse of repeats and the blank removal work
on the (num_seq,) index vector, as in the reference."""
from __future__ import annotations

import torch

from .. import functional as Fn          # Fn.ops: the HIP op layer (tests swap it for the CPU kernel references)


class GreedyCTCDecoder(torch.nn.Module):
    def __init__(self, tokenizer=None, blank_id=0):
        super().__init__()
        self.tokenizer = tokenizer
        self.blank = blank_id

    def forward(self, emission: torch.Tensor, decode=True):
        """emission: (num_seq, num_label) log-probs or logits on the GPU (numpy / CPU input is moved there).
        Returns the transcript (tokenizer given and decode=True) or the list of token ids."""
        decode = decode and self.tokenizer is not None
        if not torch.is_tensor(emission):
            emission = torch.as_tensor(emission)
        if emission.dim() != 2:
            raise ValueError(f'emission must be (num_seq, num_label), got {tuple(emission.shape)}')
        if not emission.is_cuda and torch.cuda.is_available():
            emission = emission.cuda()                                      # (no GPU: the op layer refuses CPU tensors loudly)
        indices = Fn.ops.argmax_rows(emission.float().contiguous())         # [num_seq,]
        indices = torch.unique_consecutive(indices, dim=-1).tolist()
        indices = [i for i in indices if i != self.blank]
        return self.tokenizer.decode(indices) if decode else indices
