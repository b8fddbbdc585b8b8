"""ASRLinearSCDecoder (lcasr/components/decoder.py:6-32): ff Linear(d -> V+1), reprojection Linear(V+1 -> d), norm."""
import torch.nn as nn

from .. import functional as Fn
from .normalisation import RMSNorm


class ASRLinearSCDecoder(nn.Module):
    def __init__(self, d_model, vocab_size, norm=False, norm_fn=RMSNorm, **kwargs):
        super().__init__()
        self.num_classes = vocab_size + 1                      # + blank
        if self.num_classes % 16 != 0:
            # the NT GEMM writes 16 output columns per lane and the softmax / CTC kernels move 4 classes per access
            raise ValueError(f'vocab_size + 1 (blank) must be a multiple of 16 on the HIP path, got {self.num_classes}: use e.g. '
                             f'vocab_size={(self.num_classes + 15) // 16 * 16 - 1} (the paper configs use 4095) and leave the extra ids unused')
        self.ff = nn.Linear(d_model, self.num_classes)
        self.reprojection = nn.Linear(self.num_classes, d_model)
        self.norm = norm_fn(d_model) if norm else nn.Identity()
        self.has_norm = bool(norm)

    def _np(self):
        return self.norm.norm_params() if self.has_norm else (None, None)

    def forward(self, x, logits=False, extra_norms=0, prenormed=None):
        """log_softmax(ff(norm(x)))  (decoder.py:22-26); extra_norms=1 applies the norm twice (legacy double norm).
        prenormed: the norm(s) of x in bf16 if the producer of x has already applied them (post_norm_spec(..., extra_norms))."""
        shape = x.shape
        nw, nb = self._np()
        n_norms = (1 + extra_norms) if self.has_norm else 0
        mode, eps = (self.norm.mode, self.norm.eps) if self.has_norm else ('layer_norm', 1e-5)
        y = Fn.decoder_head(x.reshape(-1, shape[-1]), nw, nb, self.ff.weight, self.ff.bias, n_norms, mode, eps, logits,
                            prenormed=None if prenormed is None else prenormed.reshape(-1, shape[-1]))
        return y.view(*shape[:-1], self.num_classes)

    def ctc_nll(self, x, targets, input_lengths, target_lengths, extra_norms=0, prenormed=None):
        """CTCLoss(blank = vocab_size, reduction='none')(log_softmax(ff(norm(x)))) as ONE operator (Fn.HeadCTCFn): (B,) negative
        log-likelihoods; the (B,N,V+1) log-probabilities and their gradient are never materialised."""
        shape = x.shape
        nw, nb = self._np()
        n_norms = (1 + extra_norms) if self.has_norm else 0
        mode, eps = (self.norm.mode, self.norm.eps) if self.has_norm else ('layer_norm', 1e-5)
        return Fn.decoder_head_ctc(x.reshape(-1, shape[-1]), nw, nb, self.ff.weight, self.ff.bias, shape[0], targets, input_lengths,
                                   target_lengths, self.num_classes - 1, n_norms, mode, eps,
                                   prenormed=None if prenormed is None else prenormed.reshape(-1, shape[-1]))

    def post_norm_spec(self, d_model, producer_mode, extra_norms=0):
        """What a layer needs to apply this decoder's norm together with its own `norm_out` (Fn.norm2): a dict the layer fills with
        'h' = norm(layer output) in bf16 (extra_norms=1: the norm applied twice, the head's legacy double norm), or None when the
        pair cannot be fused (no norm, not LayerNorm, rows wider than 768)."""
        if not self.has_norm or not Fn.norm2_enabled(d_model, producer_mode, self.norm.mode):
            return None
        nw, nb = self._np()
        return {'w': nw, 'b': nb, 'eps': self.norm.eps, 'twice': bool(extra_norms)}

    def self_condition(self, x, prenormed=None):
        """x + reprojection(softmax(ff(norm(x))))  (sconformer_xl.py:241-243).  prenormed: norm(x) in bf16 if the producer of x
        has already applied this decoder's norm (post_norm_spec)."""
        shape = x.shape
        nw, nb = self._np()
        mode, eps = (self.norm.mode, self.norm.eps) if self.has_norm else ('layer_norm', 1e-5)
        y = Fn.selfcond_block(x.reshape(-1, shape[-1]), nw, nb, self.ff.weight, self.ff.bias, self.reprojection.weight,
                              self.reprojection.bias, self.has_norm, mode, eps,
                              prenormed=None if prenormed is None else prenormed.reshape(-1, shape[-1]))
        return y.view(shape)
