"""CTC loss seam: ``ctc_loss_fn(log_probs(T,B,C), targets, input_lengths, target_lengths)`` with
torch.nn.CTCLoss(blank, reduction) semantics (exp/train.py:104,249), computed by csrc/ctc.hip."""
import torch

from . import functional as Fn


class CTCLoss(torch.nn.Module):
    def __init__(self, blank: int = 0, reduction: str = 'mean', zero_infinity: bool = False):
        super().__init__()
        if zero_infinity:
            raise NotImplementedError('zero_infinity=True is not used by the reference training loop')
        if reduction not in ('none', 'sum', 'mean'):
            raise ValueError(f'bad reduction {reduction}')
        self.blank, self.reduction = blank, reduction

    def forward(self, log_probs, targets, input_lengths, target_lengths):
        """log_probs: (T, B, C) f32 log-probabilities — typically ``out['final_posteriors'].transpose(0, 1)``,
        which is a zero-copy view of the batch-major tensor the kernels consume."""
        Fn.ops.require_gpu(log_probs, 'log_probs')
        nll = Fn.ctc_nll(log_probs.transpose(0, 1), targets, input_lengths, target_lengths, self.blank)
        if self.reduction == 'none':
            return nll
        if self.reduction == 'sum':
            return nll.sum()
        tl = target_lengths.to(nll.device).clamp_min(1).to(nll.dtype)
        return (nll / tl).mean()
