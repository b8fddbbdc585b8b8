"""ASRLinearSCDecoder (lcasr/components/decoder.py:6-32): ff Linear(d -> V+1), reprojection Linear(V+1 -> d), norm."""
import torch.nn as nn

from .. import functional as Fn
from .normalisation import RMSNorm


class ASRLinearSCDecoder(nn.Module):
    def __init__(self, d_model, vocab_size, norm=False, norm_fn=RMSNorm, **kwargs):
        super().__init__()
        self.num_classes = vocab_size + 1                      # + blank
        # The NT GEMM writes 16 output columns per lane and the softmax / CTC kernels move 4 classes per access: a class count that
        # is not a multiple of 16 (the reference's default vocab_size=128 -> 129, sconformer_xl.py:34) is PADDED for the kernels -
        # zero weight rows / reprojection columns and a -1e30 bias, so the extra classes have probability exactly 0 everywhere
        # (softmax, CTC normaliser, argmax) and receive no gradient - and sliced away again before anything is returned.
        # Parameters, state_dict and outputs keep the reference's shapes.  The paper configs (4096 classes) take no padding.
        self.padded_classes = (self.num_classes + 15) // 16 * 16
        self.ff = nn.Linear(d_model, self.num_classes)
        self.reprojection = nn.Linear(self.num_classes, d_model)
        self.norm = norm_fn(d_model) if norm else nn.Identity()
        self.has_norm = bool(norm)
        self._held = None                                      # padded stand-ins shared by the uses inside one model forward

    PAD_LOGIT = -1e30

    def _weights(self):
        """(ff.weight, ff.bias, reprojection.weight) as the kernels see them: the parameters themselves, or class-padded
        (differentiable) copies."""
        pad = self.padded_classes - self.num_classes
        if pad == 0:
            return self.ff.weight, self.ff.bias, self.reprojection.weight
        if self._held is not None:
            return self._held
        import torch.nn.functional as F
        return (F.pad(self.ff.weight, (0, 0, 0, pad)), F.pad(self.ff.bias, (0, pad), value=self.PAD_LOGIT),
                F.pad(self.reprojection.weight, (0, pad)))

    def hold_padded(self):
        """Model forward: build the padded stand-ins once for all uses of the decoder (5 self-conditioning steps + the head)."""
        if self.padded_classes != self.num_classes and self._held is None:
            self._held = self._weights()

    def release_padded(self):
        if self._held is not None:
            Fn.forget_weight_shadows(*self._held)
            self._held = None

    def _done(self, ws):
        if self._held is None and self.padded_classes != self.num_classes:
            Fn.forget_weight_shadows(*ws)                      # module-level call: the stand-ins die with it

    def _np(self):
        return self.norm.norm_params() if self.has_norm else (None, None)

    def forward(self, x, logits=False, extra_norms=0, prenormed=None):
        """log_softmax(ff(norm(x)))  (decoder.py:22-26); extra_norms=1 applies the norm twice (legacy double norm).
        prenormed: the norm(s) of x in bf16 if the producer of x has already applied them (post_norm_spec(..., extra_norms))."""
        shape = x.shape
        nw, nb = self._np()
        n_norms = (1 + extra_norms) if self.has_norm else 0
        mode, eps = (self.norm.mode, self.norm.eps) if self.has_norm else ('layer_norm', 1e-5)
        ws = self._weights()
        y = Fn.decoder_head(x.reshape(-1, shape[-1]), nw, nb, ws[0], ws[1], n_norms, mode, eps, logits,
                            prenormed=None if prenormed is None else prenormed.reshape(-1, shape[-1]))
        self._done(ws)
        y = y.view(*shape[:-1], self.padded_classes)
        return y if self.padded_classes == self.num_classes else y[..., :self.num_classes].contiguous()

    def ctc_nll(self, x, targets, input_lengths, target_lengths, extra_norms=0, prenormed=None):
        """CTCLoss(blank = vocab_size, reduction='none')(log_softmax(ff(norm(x)))) as ONE operator (Fn.HeadCTCFn): (B,) negative
        log-likelihoods; the (B,N,V+1) log-probabilities and their gradient are never materialised."""
        shape = x.shape
        nw, nb = self._np()
        n_norms = (1 + extra_norms) if self.has_norm else 0
        mode, eps = (self.norm.mode, self.norm.eps) if self.has_norm else ('layer_norm', 1e-5)
        ws = self._weights()
        nll = Fn.decoder_head_ctc(x.reshape(-1, shape[-1]), nw, nb, ws[0], ws[1], shape[0], targets, input_lengths,
                                  target_lengths, self.num_classes - 1, n_norms, mode, eps,
                                  prenormed=None if prenormed is None else prenormed.reshape(-1, shape[-1]),
                                  num_labels=self.num_classes)
        self._done(ws)
        return nll

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
        ws = self._weights()
        y = Fn.selfcond_block(x.reshape(-1, shape[-1]), nw, nb, ws[0], ws[1], ws[2],
                              self.reprojection.bias, self.has_norm, mode, eps,
                              prenormed=None if prenormed is None else prenormed.reshape(-1, shape[-1]))
        self._done(ws)
        return y.view(shape)
