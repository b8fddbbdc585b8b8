"""Autograd blocks of the SConformerXL hot path, composed from the HIP ops (``lcasr_amd.hip.ops``).

Each block is ONE ``torch.autograd.Function`` covering a residual branch of the reference
(`/root/reference/lcasr/models/sconformer_xl.py:346-372`) so that
  * the residual stream stays f32 (what bf16-autocast gives the reference: LayerNorm outputs f32 and
    ``bf16 + f32 -> f32``), every GEMM operand is bf16 and accumulates in f32;
  * bias / GELU / GLU / residual / 0.5-scale live in GEMM epilogues or in the neighbouring HBM-bound kernel;
  * the backward fuses the residual gradient into the norm backward (dx = dy + norm'(..)) and never transposes an
    ACTIVATION in HBM: wgrad is a TN GEMM (hardware transposed LDS reads), dgrad is an NT GEMM against a transposed bf16
    shadow of the (few-MB) weight, cast once per forward.

Parameters stay f32 masters in the reference's ``state_dict`` layout; a bf16 copy is cast once per forward
(``wcast``) and reused by the backward.
"""
from __future__ import annotations

import os

from typing import Optional

import torch
from torch.autograd import Function

from .hip import ops

BF16, F32 = torch.bfloat16, torch.float32

# ---- bf16 weight shadows ---------------------------------------------------------------------------------------------
# The GEMMs read bf16 copies of the f32 master weights: a row-major copy (forward, NT) and a transposed copy (dgrad as
# an NT GEMM).  The copies live in persistent buffers; `refresh_weight_shadows()` (called at the start of every model
# forward) re-derives ALL of them from the masters in one kernel launch.  Module-level calls (layer.ff1(x), decoder(x), ...)
# do not pass through the model forward, so every shadow also remembers the state of its master when it was cast - the
# tensor's version counter plus an epoch that the fused optimiser bumps (its kernel writes the parameters behind autograd's
# back) - and `wcast` / `wcast_t` re-cast a shadow whose master has moved on.  What the check cannot see is a write through
# `.data` (no version bump): whoever does that calls `bump_weight_epoch()` (parallel.broadcast_module_state does) or
# `clear_weight_cache()`; a model forward re-casts everything regardless.
class _Shadow:
    __slots__ = ('w', 'rows', 'cols', 'n', 't', 'vn', 'vt', 'regroup')

    def __init__(self, w, regroup=False):
        self.w = w                                   # keeps the master alive and lets refresh re-read its data_ptr
        self.rows = w.shape[0]; self.cols = w.numel() // max(w.shape[0], 1)
        self.n = None; self.t = None
        self.vn = self.vt = None                     # (version, epoch) of the master at the last cast of n / t
        self.regroup = regroup                       # rows "(h d qkv)" -> [q | k | v] (the fused qkv projection, attention.py:485)


_shadows: dict = {}                                  # (data_ptr, shape) -> _Shadow
_table = {'key': None, 'dev': None, 'tiles': 0}
_epoch = [0]


def bump_weight_epoch() -> None:
    """Parameters were rewritten in place by something autograd's version counters do not see (the fused MADGRAD kernel)."""
    _epoch[0] += 1


def _state(e: _Shadow):
    return (e.w._version, _epoch[0])


def forget_weight_shadows(*ws: torch.Tensor) -> None:
    """Drop the shadows of temporaries that stood in for parameters during one forward (the class-padded decoder weights of a
    vocabulary that is not a multiple of 16): they would otherwise be kept alive, and re-cast, for ever."""
    for w in ws:
        for rg in (False, True):
            _shadows.pop((w.data_ptr(), tuple(w.shape), rg), None)


def clear_weight_cache() -> None:
    """Forget every shadow (tests; after re-pointing parameter storage)."""
    _shadows.clear()
    _table.update(key=None, dev=None, tiles=0)


def refresh_weight_shadows() -> None:
    """Re-cast all registered shadows from their f32 masters: one launch (sconf_cast_shadows)."""
    _twins.clear()                                   # gradients parked by a backward nobody consumed
    if not _shadows:
        return
    ents = list(_shadows.values())
    key = tuple((e.w.data_ptr(), 0 if e.n is None else e.n.data_ptr(), 0 if e.t is None else e.t.data_ptr()) for e in ents)
    if key != _table['key']:
        rows, tile0 = [], 0
        for e, k in zip(ents, key):
            rows.append([k[0], k[1], k[2], -e.rows if e.regroup else e.rows, e.cols, tile0])
            tile0 += ((e.rows + 31) // 32) * ((e.cols + 31) // 32)
        rows.append([0, 0, 0, 0, 0, tile0])
        _table.update(key=key, dev=torch.tensor(rows, dtype=torch.int64, device=ents[0].w.device), tiles=tile0)
    ops.cast_shadows(_table['dev'], len(ents), _table['tiles'])
    for e in ents:
        st = _state(e)
        if e.n is not None: e.vn = st
        if e.t is not None: e.vt = st


def _shadow(w: torch.Tensor, regroup: bool = False) -> _Shadow:
    key = (w.data_ptr(), tuple(w.shape), regroup)
    e = _shadows.get(key)
    if e is None:
        if w.dtype != F32:
            raise TypeError('master weights are float32')
        if regroup and w.shape[0] % 3:
            raise ValueError('a regrouped shadow needs 3 * H * D rows')
        e = _Shadow(w.detach(), regroup)
        _shadows[key] = e
    return e


def _regrouped(w2: torch.Tensor) -> torch.Tensor:
    """rows (j, which) -> (which, j): "(h d qkv)" -> [q | k | v] (what sconf_cast_shadows does for a regroup entry)."""
    R, C = w2.shape
    return w2.view(R // 3, 3, C).permute(1, 0, 2).reshape(R, C)


def wcast(w: torch.Tensor, regroup: bool = False) -> torch.Tensor:
    """bf16 copy of an f32 master weight, viewed 2-D (out_features, in_features*k).  regroup: the fused qkv projection's rows
    "(h d qkv)" come out as [q | k | v], so its GEMM writes three contiguous (h, d) blocks per token."""
    e = _shadow(w, regroup)
    st = _state(e)
    if e.n is None or e.vn != st:                    # first use / the master changed since the last cast (module-level use after an update)
        src = w.detach().reshape(e.rows, e.cols)
        fresh = ops.cast(_regrouped(src).contiguous() if regroup else src, BF16)
        if e.n is None: e.n = fresh                  # refreshed in bulk from the next forward on
        else: e.n.copy_(fresh)
    e.vn = st
    return e.n


def wcast_t(w: torch.Tensor, regroup: bool = False) -> torch.Tensor:
    """Transposed bf16 copy (in_features*k, out_features) of an f32 master weight: dgrad dx = dy W becomes the NT GEMM
    dy (W^T)^T whose B operand is K-contiguous (wide epilogue, no transposed LDS reads).  Weights are a few MB."""
    e = _shadow(w, regroup)
    st = _state(e)
    if e.t is None or e.vt != st:
        src = w.detach().reshape(e.rows, e.cols)
        fresh = ops.cast_transpose((_regrouped(src) if regroup else src).contiguous())
        if e.t is None: e.t = fresh
        else: e.t.copy_(fresh)
    e.vt = st
    return e.t


# ---- parameter gradients ---------------------------------------------------------------------------------------------
# Default: the backward functions RETURN parameter gradients and autograd accumulates them (p.grad += g: one add kernel
# per parameter plus the zero-fills of the temporaries).  With `set_direct_grad(True)` (the training driver does this once
# its flat gradient buffer is attached to every p.grad) the kernels accumulate straight into p.grad and the functions
# return None for those inputs.  That is only valid under loss.backward() semantics (never torch.autograd.grad), which is
# why it is opt-in.  `set_grad_ready_hook(fn)` gets fn(param) after each direct write: the data-parallel bucket logic
# (parallel.GradSync) listens there, since autograd's post-accumulate hooks do not fire for gradients it never sees.
_direct = {'on': False, 'hook': None}


def set_direct_grad(on: bool) -> None:
    _direct['on'] = bool(on)


def set_grad_ready_hook(fn) -> None:
    _direct['hook'] = fn


class _G:
    """Destination of one parameter's gradient inside a backward: p.grad itself (direct mode) or a zeroed temporary."""
    __slots__ = ('p', 'direct', 't')

    def __init__(self, p: Optional[torch.Tensor], shape=None):
        self.p = p
        if p is None:
            self.direct, self.t = False, None
            return
        g = p.grad if p.is_leaf else None                # (a non-leaf stands in for a parameter: the class-padded decoder weights)
        self.direct = bool(_direct['on'] and g is not None and g.dtype == F32 and g.is_contiguous() and g.shape == p.shape)
        shape = tuple(p.shape) if shape is None else tuple(shape)
        self.t = g.view(shape) if self.direct else torch.zeros(shape, dtype=F32, device=p.device)

    def out(self):
        """What the backward returns for this input."""
        if self.p is None:
            return None
        if self.direct:
            if _direct['hook'] is not None:
                _direct['hook'](self.p)
            return None
        return self.t.view(self.p.shape)


# ---- bf16 twins of residual-stream gradients -------------------------------------------------------------------------------
# A block's backward ends in norm_bwd, which writes dx (f32, the gradient of the residual stream).  The block BEFORE it
# receives that tensor as dy and starts with dy16 = bf16(dy) and, if its output projection has a bias, colsum(dy16): two
# more passes over 4 M d bytes.  norm_bwd can produce both in the pass it already makes (ops.norm_bwd(twin=True)); the
# producer parks them here and the consumer takes them IF dy is that very memory, unmodified (same storage + version counter;
# the strong reference keeps the storage alive and keeps autograd from accumulating into it in place).  Anything else falls back to cast / colsum.
_twins: dict = {}


def _park_twin(dx: torch.Tensor, dx16: torch.Tensor, colsum: torch.Tensor) -> None:
    _twins[dx.data_ptr()] = (dx, dx._version, dx16, colsum)       # the entry keeps dx's storage alive: the address cannot be reused


def _take_twin(dy: torch.Tensor):
    """(dy16, colsum or None) for an f32 residual-stream gradient: the parked twin if dy is the parked dx (or a view of all of
    it - autograd reshapes between blocks; views share the version counter), else a fresh cast."""
    e = _twins.pop(dy.data_ptr(), None)
    if e is not None and dy.dtype == F32 and dy.is_contiguous() and dy.numel() == e[0].numel() and dy._version == e[1] \
            and dy.untyped_storage().data_ptr() == e[0].untyped_storage().data_ptr():
        return e[2].view(dy.shape), e[3]
    return ops.cast(dy, BF16), None


def _pre(x, nw, nb, mode, eps):
    """The PreNorm in front of a block -> (h bf16, mean, rstd).  mode 'none': the module-level forward() of the wrapped module
    (attention.py:509, fused_dense.py:489, convolution.py:103), whose input is already normalised: a bf16 cast."""
    if mode == 'none':
        return ops.cast(x, BF16), None, None
    return ops.norm_fwd(x, nw, nb, mode, eps, BF16)


def _norm_bwd_res(dh, x, nw, mean, rstd, mode, eps, dres, dnw, dnb, twin=True):
    """norm backward of a residual branch: dx = dres + norm'(x) dh (f32), with the bf16 twin parked for the receiving block
    (twin=False when the receiver is not a GEMM block: the self-conditioning block hands its dx to norm_out's backward, which
    reads the f32 tensor)."""
    if mode == 'none':
        if dres is not None:
            raise RuntimeError('a block without its PreNorm has no fused residual')
        return ops.cast(dh, x.dtype)
    if dres is None or not twin:
        return ops.norm_bwd(dh, x, nw, mean, rstd, mode, eps, dres, F32, dnw, dnb)
    dx, dx16, cs = ops.norm_bwd(dh, x, nw, mean, rstd, mode, eps, dres, F32, dnw, dnb, twin=True)
    _park_twin(dx, dx16, cs)
    return dx


def _wgrad(dy16: torch.Tensor, x16: torch.Tensor, w: torch.Tensor, alpha: float = 1.0):
    """dW[N',K'] (+)= alpha * dy^T x  (TN GEMM, split-K over the token dimension when the tile count is small)."""
    Mtok, Nout = dy16.shape
    Kin = x16.shape[1]
    sk = ops.pick_split_k(Nout, Kin, Mtok)
    g = _G(w, (Nout, Kin))
    if g.direct:
        ops.gemm(dy16, x16, 'tn', alpha=alpha, out_dtype=F32, split_k=sk, accum=g.t)
        return g.out()
    return ops.gemm(dy16, x16, 'tn', alpha=alpha, out_dtype=F32, split_k=sk).reshape(w.shape)


def _bgrad(dy16: torch.Tensor, bias: Optional[torch.Tensor], alpha: float = 1.0, colsum: Optional[torch.Tensor] = None):
    """bias gradient = alpha * column sums of dy16 (taken from `colsum` when the producer of dy already made them)."""
    if bias is None:
        return None
    g = _G(bias)
    if colsum is not None:
        g.t.add_(colsum.view(g.t.shape), alpha=alpha)
    else:
        ops.colsum_(dy16, g.t, alpha)
    return g.out()


# =================================================================================================
# standalone norm  (norm_out, decoder norms) — sconformer_xl.py:371, decoder.py:23
# =================================================================================================
class NormFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, mode: str, eps: float, out_dtype):
        x = x.contiguous()
        y, mean, rstd = ops.norm_fwd(x, weight, bias, mode, eps, out_dtype)
        ctx.save_for_backward(x, weight, mean, rstd)
        ctx.mode, ctx.eps, ctx.P = mode, eps, (weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, mean, rstd = ctx.saved_tensors
        dw, db = _G(ctx.P[0]), _G(ctx.P[1])
        if x.dtype == F32:                      # norm_out: the layer's last block (ff2) receives dx as its output gradient
            dx, dx16, cs = ops.norm_bwd(dy.contiguous(), x, weight, mean, rstd, ctx.mode, ctx.eps, None, F32, dw.t, db.t, twin=True)
            _park_twin(dx, dx16, cs)
        else:
            dx = ops.norm_bwd(dy.contiguous(), x, weight, mean, rstd, ctx.mode, ctx.eps, None, x.dtype, dw.t, db.t)
        return dx, dw.out(), db.out(), None, None, None


def norm(x, weight, bias, mode='layer_norm', eps=1e-5, out_dtype=F32):
    return NormFn.apply(x, weight, bias, mode, eps, out_dtype)


# =================================================================================================
# norm_out followed by the decoder norm of the self-conditioning step — sconformer_xl.py:371 then 241-243, decoder.py:23
# =================================================================================================
class Norm2Fn(Function):
    """(y1, h2) = (LN(x; w1, b1) f32, LN(y1; w2, b2) bf16) in one pass; the backward takes both gradients at once, so the gradient
    of y1 the second norm contributes never exists in memory (16 instead of 28 bytes per element; forward 10 instead of 14).
    twice: h2 = LN(LN(y1; w2, b2); w2, b2) - norm_out of the LAST layer followed by the head's legacy double norm (36 -> 12 and 22 -> 10)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, eps1: float, eps2: float, twice: bool):
        ctx.set_materialize_grads(False)                  # an unused output's gradient arrives as None, not as an (M, d) zero tensor:
        x = x.contiguous()                                # the last layer's y1 (the head takes h2 only) is 0.8 GB at the bench shape
        y1, h2, st = ops.norm2_fwd(x, w1, b1, w2, b2, eps1, eps2, twice)
        ctx.save_for_backward(x, w1, b1, w2, b2, *st)
        ctx.P, ctx.eps1 = (w1, b1, w2, b2), eps1
        return y1, h2

    @staticmethod
    def backward(ctx, dy1, dh2):
        sv = ctx.saved_tensors                            # ONE access: inside a checkpointed region a second unpack raises
        x, w1, b1, w2, b2 = sv[:5]
        st = tuple(sv[5:])
        pw1, pb1, pw2, pb2 = ctx.P
        if dy1 is None and dh2 is None:
            return (None,) * 8
        dw1, db1 = _G(pw1), _G(pb1)
        if dh2 is None:                                   # h2 was not used: this is norm_out alone
            dx, dx16, cs = ops.norm_bwd(dy1.contiguous(), x, w1, st[0], st[1], 'layer_norm', ctx.eps1, None, F32, dw1.t, db1.t, twin=True)
            _park_twin(dx, dx16, cs)
            return dx, dw1.out(), db1.out(), None, None, None, None, None
        dw2, db2 = _G(pw2), _G(pb2)
        dx, dx16, cs = ops.norm2_bwd(dh2.contiguous(), x, w1, b1, w2, b2, st, None if dy1 is None else dy1.contiguous(),
                                     dw1.t, db1.t, dw2.t, db2.t, twin=True)
        _park_twin(dx, dx16, cs)                          # the layer's last block (ff2) receives dx as its output gradient
        return dx, dw1.out(), db1.out(), dw2.out(), db2.out(), None, None, None


def norm2_enabled(d: int, *modes) -> bool:
    """The fused pair exists for LayerNorm rows of at most 768 elements (SCONF_NORM2=0 switches it off: A/B, tests)."""
    return d % 4 == 0 and d <= 768 and all(m == 'layer_norm' for m in modes) and os.environ.get('SCONF_NORM2', '1') != '0'


def norm2(x, w1, b1, w2, b2, eps1=1e-5, eps2=1e-5, twice=False):
    return Norm2Fn.apply(x, w1, b1, w2, b2, eps1, eps2, twice)


# =================================================================================================
# x + scale * FusedMLP(norm(x))  — Scale(0.5, PreNorm(FusedMLP)); fused_dense.py:425-498, wrappers.py:5-28
# =================================================================================================
class FFBlockFn(Function):
    @staticmethod
    def forward(ctx, x, nw, nb, w1, w2, b1, b2, scale: float, mode: str, eps: float, ckpt: int, residual: bool):
        x = x.contiguous()
        h, mean, rstd = _pre(x, nw, nb, mode, eps)
        w1h, w2h = wcast(w1), wcast(w2)
        a, u = ops.gemm(h, w1h, 'nt', bias=b1, act='gelu_dsave', save_pre=True)   # u := gelu'(pre-activation), bf16
        y = ops.gemm(a, w2h, 'nt', bias=b2, resid=x if residual else None, alpha=scale, out_dtype=F32)
        w1t, w2t = wcast_t(w1), wcast_t(w2)
        if ckpt >= 1:                       # checkpoint_lvl 1/2 (fused_dense.py:283-289): recompute in backward
            ctx.save_for_backward(x, nw, nb, mean, rstd, w1h, w2h, b1, b2, w1t, w2t)
        else:
            ctx.save_for_backward(x, nw, nb, mean, rstd, w1h, w2h, b1, b2, w1t, w2t, h, u, a)
        ctx.cfg = (scale, mode, eps, ckpt, residual)
        ctx.P = (nw, nb, w1, w2, b1, b2)
        return y

    @staticmethod
    def backward(ctx, dy):
        scale, mode, eps, ckpt, residual = ctx.cfg
        pnw, pnb, pw1, pw2, pb1, pb2 = ctx.P
        t = ctx.saved_tensors
        x, nw, nb, mean, rstd, w1h, w2h, b1, b2, w1t, w2t = t[:11]
        if ckpt >= 1:
            h, _, _ = _pre(x, nw, nb, mode, eps)
            a, u = ops.gemm(h, w1h, 'nt', bias=b1, act='gelu_dsave', save_pre=True)
        else:
            h, u, a = t[11:]
        dy = dy.contiguous()
        dy16, dycs = _take_twin(dy)
        du = ops.gemm(dy16, w2t, 'nt', aux=u, act='mulaux', alpha=scale)           # (M,4d): dy W2 * gelu'(pre)
        dw2 = _wgrad(dy16, a, pw2, alpha=scale)
        db2 = _bgrad(dy16, pb2, alpha=scale, colsum=dycs)
        dw1 = _wgrad(du, h, pw1)
        db1 = _bgrad(du, pb1)
        dh = ops.gemm(du, w1t, 'nt')
        dnw, dnb = _G(pnw), _G(pnb)
        dx = _norm_bwd_res(dh, x, nw, mean, rstd, mode, eps, dy if residual else None, dnw.t, dnb.t)
        return dx, dnw.out(), dnb.out(), dw1, dw2, db1, db2, None, None, None, None, None


def ff_block(x, nw, nb, w1, w2, b1, b2, scale=0.5, mode='layer_norm', eps=1e-5, ckpt=0, residual=True):
    """residual=True: x + scale*MLP(norm(x)) (ConformerLayer fast path); False: the branch alone (module-level call)."""
    return FFBlockFn.apply(x, nw, nb, w1, w2, b1, b2, scale, mode, eps, ckpt, residual)


# =================================================================================================
# x + out_proj(Attention(rotary(qkv(norm(x)))))  — PreNorm(Attention); attention.py:509-551
# =================================================================================================
class AttnBlockFn(Function):
    """The qkv projection reads a REGROUPED bf16 shadow of its weight (rows [q | k | v] instead of the reference's "(h d qkv)"
    interleave), so its GEMM writes (M, 3, H, D) and q, k, v are strided views of that one buffer: no de-interleave pass.
    Rotary is applied in place to the q and k blocks (forward) and, transposed, inside the attention backward's dQ / dK
    epilogues, which write straight into one (M, 3, H, D) gradient buffer = the operand of the qkv dgrad / wgrad GEMMs.  Only the
    (few MB) weight gradient is brought back to the reference's row order."""

    @staticmethod
    def forward(ctx, x, nw, nb, wqkv, wout, bqkv, bout, cos, sin, lengths, B: int, N: int, H: int, D: int, window,
                mode: str, eps: float, residual: bool):
        x = x.contiguous()
        h, mean, rstd = _pre(x, nw, nb, mode, eps)
        if lengths is not None:
            if h is x: h = h.clone()
            ops.mask_rows_(h, lengths, B, N)                                          # attention.py:511
        wqh, woh = wcast(wqkv, regroup=True), wcast(wout)
        bq = None if bqkv is None else _regrouped(bqkv.detach().view(-1, 1)).view(-1).contiguous()
        if cos is not None:                                                           # rotation in the GEMM epilogue (head_dim 128), else in place
            qkv = ops.gemm_qkv_rotary(h, wqh, bq, cos, sin, N, H, D)                  # (M, 3*H*D) = (B, N, 3, H, D)
        else:
            qkv = ops.gemm(h, wqh, 'nt', bias=bq)
        q5 = qkv.view(B, N, 3, H, D)
        o, lse = ops.attn_fwd(q5[:, :, 0], q5[:, :, 1], q5[:, :, 2], lengths, window)  # padded query rows come back zero
        y = ops.gemm(o.view(B * N, H * D), woh, 'nt', bias=bout, resid=x if residual else None, out_dtype=F32)
        ctx.save_for_backward(x, nw, nb, mean, rstd, wcast_t(wqkv, regroup=True), wcast_t(wout), bqkv, bout, cos, sin, lengths, h, qkv, o, lse)
        ctx.cfg = (B, N, H, D, window, mode, eps, residual)
        ctx.P = (nw, nb, wqkv, wout, bqkv, bout)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, N, H, D, window, mode, eps, residual = ctx.cfg
        pnw, pnb, pwq, pwo, pbq, pbo = ctx.P
        x, nw, nb, mean, rstd, wqt, wot, bqkv, bout, cos, sin, lengths, h, qkv, o, lse = ctx.saved_tensors
        dy = dy.contiguous()
        dy16, dycs = _take_twin(dy)
        o2 = o.view(B * N, H * D)
        do = ops.gemm(dy16, wot, 'nt')                                                # (M, H*D)
        dwo = _wgrad(dy16, o2, pwo)
        dbo = _bgrad(dy16, pbo, colsum=dycs)
        q5 = qkv.view(B, N, 3, H, D)
        dqkv = torch.empty_like(qkv)
        d5 = dqkv.view(B, N, 3, H, D)
        ops.attn_bwd(q5[:, :, 0], q5[:, :, 1], q5[:, :, 2], o, do.view(B, N, H, D), lse, lengths, window,
                     rot=None if cos is None else (cos, sin), out=(d5[:, :, 0], d5[:, :, 1], d5[:, :, 2]))
        # weight / bias gradients come out in the regrouped row order: bring the (3HD, d) result back to "(h d qkv)"
        Mtok, HD3 = dqkv.shape
        sk = ops.pick_split_k(HD3, h.shape[1], Mtok)
        gq = ops.gemm(dqkv, h, 'tn', out_dtype=F32, split_k=sk)                       # rows [q | k | v]
        gq = gq.view(3, HD3 // 3, -1).permute(1, 0, 2)                                # -> (h d) x qkv x in_features (a view)
        gw = _G(pwq, (HD3 // 3, 3, h.shape[1]))
        if gw.direct: gw.t.add_(gq)
        dwq = gw.out() if gw.direct else gq.reshape(pwq.shape)
        dbq = None
        if pbq is not None:
            gb = _G(pbq, (HD3 // 3, 3))
            cs = torch.zeros(HD3, dtype=F32, device=dqkv.device)
            ops.colsum_(dqkv, cs)
            if gb.direct: gb.t.add_(cs.view(3, HD3 // 3).t())
            dbq = gb.out() if gb.direct else cs.view(3, HD3 // 3).t().reshape(pbq.shape)
        dh = ops.gemm(dqkv, wqt, 'nt')
        if lengths is not None:
            ops.mask_rows_(dh, lengths, B, N)
        dnw, dnb = _G(pnw), _G(pnb)
        dx = _norm_bwd_res(dh, x, nw, mean, rstd, mode, eps, dy if residual else None, dnw.t, dnb.t)
        return (dx, dnw.out(), dnb.out(), dwq, dwo, dbq, dbo) + (None,) * 11


def attn_block(x, nw, nb, wqkv, wout, bqkv, bout, cos, sin, lengths, B, N, H, D, window=(-1, -1), mode='layer_norm', eps=1e-5,
               residual=True):
    return AttnBlockFn.apply(x, nw, nb, wqkv, wout, bqkv, bout, cos, sin, lengths, B, N, H, D, tuple(window), mode, eps, residual)


# =================================================================================================
# x + ConformerConvolution(norm(x))  — PreNorm(ConformerConvolution) with BatchRenorm1d; convolution.py:103-124
# =================================================================================================
BRN_EPS, BRN_MOMENTUM = 1e-3, 0.01                                                    # batchrenorm.py:13-14


class ConvBlockFn(Function):
    @staticmethod
    def forward(ctx, x, nw, nb, wpw1, bpw1, wdw, bdw, brn_w, brn_b, running_mean, running_std, nbt, wpw2, bpw2, lengths,
                B: int, N: int, training: bool, mode: str, eps: float, residual: bool):
        x = x.contiguous()
        d = x.shape[-1]
        h, mean, rstd = _pre(x, nw, nb, mode, eps)
        w1h, w2h = wcast(wpw1), wcast(wpw2)
        g = ops.gemm(h, w1h, 'nt', bias=bpw1)                                         # (M, 2d)
        wdw2 = wdw.detach().reshape(d, -1).contiguous()
        hc, stats = ops.glu_dwconv_fwd(g, lengths, wdw2, bdw, B, N)
        coef = ops.brn_finalize(stats, B * N, running_mean, running_std, nbt, brn_w, brn_b, training, BRN_EPS, BRN_MOMENTUM)
        y2 = ops.affine_silu_fwd(hc, coef)
        y = ops.gemm(y2, w2h, 'nt', bias=bpw2, resid=x if residual else None, out_dtype=F32)
        ctx.save_for_backward(x, nw, nb, mean, rstd, wcast_t(wpw1), wcast_t(wpw2), bpw1, bpw2, wdw2, brn_w, lengths, h, g, hc, coef, y2)
        ctx.cfg = (B, N, training, mode, eps, residual)
        ctx.P = (nw, nb, wpw1, bpw1, wdw, bdw, brn_w, brn_b, wpw2, bpw2)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, N, training, mode, eps, residual = ctx.cfg
        pnw, pnb, pw1, pb1, pwdw, pbdw, pbrnw, pbrnb, pw2, pb2 = ctx.P
        x, nw, nb, mean, rstd, w1t, w2t, bpw1, bpw2, wdw2, brn_w, lengths, h, g, hc, coef, y2 = ctx.saved_tensors
        dy = dy.contiguous()
        dy16, dycs = _take_twin(dy)
        dy2 = ops.gemm(dy16, w2t, 'nt')                                               # (M, d)
        dw2 = _wgrad(dy16, y2, pw2)
        db2 = _bgrad(dy16, pb2, colsum=dycs)
        ddw, dbdw, dbrnw, dbrnb = _G(pwdw, wdw2.shape), _G(pbdw), _G(pbrnw), _G(pbrnb)
        dg, dgcs = ops.convmod_bwd(dy2, hc, g, lengths, wdw2, brn_w, coef, B, N, training, BRN_EPS, ddw.t, dbdw.t, dbrnw.t, dbrnb.t,
                                   colsum=pb1 is not None)
        dw1 = _wgrad(dg, h, pw1)
        db1 = _bgrad(dg, pb1, colsum=dgcs)
        dh = ops.gemm(dg, w1t, 'nt')
        dnw, dnb = _G(pnw), _G(pnb)
        dx = _norm_bwd_res(dh, x, nw, mean, rstd, mode, eps, dy if residual else None, dnw.t, dnb.t)
        return (dx, dnw.out(), dnb.out(), dw1, db1, ddw.out(), dbdw.out(), dbrnw.out(), dbrnb.out(), None, None, None, dw2, db2) + (None,) * 7


def conv_block(x, nw, nb, wpw1, bpw1, wdw, bdw, brn_w, brn_b, running_mean, running_std, nbt, wpw2, bpw2, lengths, B, N,
               training, mode='layer_norm', eps=1e-5, residual=True):
    return ConvBlockFn.apply(x, nw, nb, wpw1, bpw1, wdw, bdw, brn_w, brn_b, running_mean, running_std, nbt, wpw2, bpw2,
                             lengths, B, N, training, mode, eps, residual)


# =================================================================================================
# self-conditioning: x + reprojection(softmax(ff(norm(x))))  — sconformer_xl.py:241-243, decoder.py:6-32
# =================================================================================================
def sc_delta_enabled(M: int, V: int, d: int) -> bool:
    """Self-conditioning backward with the softmax backward inside the reprojection's dgrad GEMM (SCONF_SC_DELTA=0: the separate
    GEMM + softmax_bwd passes): needs a shape the 256-row NT GEMM kernel takes (the paper configs; not the tiny test models)."""
    return os.environ.get('SCONF_SC_DELTA', '1') != '0' and ops.gemm_softmax_bwd_eligible(M, V, d)


class SelfCondFn(Function):
    @staticmethod
    def forward(ctx, x, hn_pre, nw, nb, wff, bff, wre, bre, has_norm: bool, mode: str, eps: float):
        """hn_pre (bf16, optional): norm(x) already computed by the producer of x (Norm2Fn) - its gradient is returned to it."""
        x = x.contiguous()
        if hn_pre is not None:
            hn, mean, rstd = hn_pre.contiguous(), None, None
        elif has_norm:
            hn, mean, rstd = ops.norm_fwd(x, nw, nb, mode, eps, BF16)
        else:
            hn, mean, rstd = ops.cast(x, BF16), None, None
        wfh, wrh = wcast(wff), wcast(wre)
        logits = ops.gemm(hn, wfh, 'nt', bias=bff)                                    # (M, V+1) bf16
        p = ops.softmax_fwd(logits, False, BF16)
        # Backward without the (M, V+1) gradient of the probabilities (sc_delta): the softmax backward needs delta = sum_v p dp with
        # dp = dy Wr; that is sum_c dy[c] r[c] with r = p Wr^T, the product of THIS GEMM - kept in bf16 (acc + bias, 2 B per element of
        # (M, d)) so that the backward's dgrad GEMM can apply p * (dp - delta) in its epilogue and never write dp.
        delta_path = sc_delta_enabled(p.shape[0], p.shape[1], x.shape[1])
        if delta_path:
            y, r16 = ops.gemm(p, wrh, 'nt', bias=bre, resid=x, out_dtype=F32, save_pre=True)
        else:
            y, r16 = ops.gemm(p, wrh, 'nt', bias=bre, resid=x, out_dtype=F32), None
        ctx.save_for_backward(x, nw, nb, mean, rstd, wcast_t(wff), wcast_t(wre), bff, bre, hn, p, r16)
        ctx.cfg = (has_norm, mode, eps, hn_pre is not None)
        ctx.P = (nw, nb, wff, bff, wre, bre)
        return y

    @staticmethod
    def backward(ctx, dy):
        has_norm, mode, eps, pre = ctx.cfg
        pnw, pnb, pwf, pbf, pwr, pbr = ctx.P
        x, nw, nb, mean, rstd, wft, wrt, bff, bre, hn, p, r16 = ctx.saved_tensors
        dy = dy.contiguous()
        dy16, dycs = _take_twin(dy)
        dwr = _wgrad(dy16, p, pwr)
        dbr = _bgrad(dy16, pbr, colsum=dycs)
        gbf = _G(pbf)                                                                  # bias gradient = column sums of dl, same pass
        if r16 is not None:
            delta = ops.rowdot(dy16, r16, bre)                                         # (M,) = sum_v p * (dy Wr)
            dl = ops.gemm_softmax_bwd(dy16, wrt, p, delta, colsum_into=gbf.t)          # p * (dy Wr - delta), bf16
        else:
            dp = ops.gemm(dy16, wrt, 'nt')                                             # (M, V+1)
            dl = ops.softmax_bwd(p, dp, False, BF16, colsum_into=gbf.t)
        dwf = _wgrad(dl, hn, pwf)
        dbf = gbf.out()
        dhn = ops.gemm(dl, wft, 'nt')
        if pre:                                                                        # the producer's backward applies both norms
            return dy, dhn, None, None, dwf, dbf, dwr, dbr, None, None, None
        if has_norm:
            dnw, dnb = _G(pnw), _G(pnb)
            dx = _norm_bwd_res(dhn, x, nw, mean, rstd, mode, eps, dy, dnw.t, dnb.t, twin=False)
            return dx, None, dnw.out(), dnb.out(), dwf, dbf, dwr, dbr, None, None, None
        dx = dy + ops.cast(dhn, F32)
        return dx, None, None, None, dwf, dbf, dwr, dbr, None, None, None


def selfcond_block(x, nw, nb, wff, bff, wre, bre, has_norm=True, mode='layer_norm', eps=1e-5, prenormed=None):
    return SelfCondFn.apply(x, prenormed, nw, nb, wff, bff, wre, bre, has_norm, mode, eps)


# =================================================================================================
# decoder head: [norm] -> norm -> Linear -> log_softmax   — sconformer_xl.py:246-247, decoder.py:22-26
# =================================================================================================
class HeadFn(Function):
    @staticmethod
    def forward(ctx, x, hn_pre, nw, nb, wff, bff, n_norms: int, mode: str, eps: float, return_logits: bool):
        """hn_pre (bf16, optional): the decoder norm(s) of x already applied by the producer of x (Norm2Fn) - x is then unused here
        and the gradient of hn_pre goes back to the producer."""
        saved_norm = []
        if hn_pre is not None:
            hn, n_norms = hn_pre.contiguous(), -1
        else:
            x = x.contiguous()
            cur = x
            for i in range(n_norms):                                                   # legacy double norm: applied twice
                out_dt = BF16 if i == n_norms - 1 else F32
                y, mean, rstd = ops.norm_fwd(cur, nw, nb, mode, eps, out_dt)
                saved_norm += [cur, mean, rstd]
                cur = y
            hn = cur if n_norms > 0 else ops.cast(x, BF16)
        wfh = wcast(wff)
        logits = ops.gemm(hn, wfh, 'nt', bias=bff, out_dtype=F32)
        out = logits if return_logits else ops.softmax_fwd(logits, True, F32)
        ctx.save_for_backward(nw, nb, wcast_t(wff), bff, hn, out, *saved_norm)
        ctx.cfg = (n_norms, mode, eps, return_logits)
        ctx.P = (nw, nb, wff, bff)
        return out

    @staticmethod
    def backward(ctx, dout):
        n_norms, mode, eps, return_logits = ctx.cfg
        pnw, pnb, pwf, pbf = ctx.P
        sv = ctx.saved_tensors
        nw, nb, wft, bff, hn, out = sv[:6]
        sn = sv[6:]
        dout = dout.contiguous()
        if return_logits:
            dl = ops.cast(dout, BF16)
            dbf = _bgrad(dl, pbf)
        else:
            gbf = _G(pbf)
            dl = ops.softmax_bwd(out, dout, True, BF16, colsum_into=gbf.t)
            dbf = gbf.out()
        dwf = _wgrad(dl, hn, pwf)
        g = ops.gemm(dl, wft, 'nt')                                                    # (M,d) bf16
        if n_norms < 0:                                                                # pre-normalised input: its producer applies the norms' backward
            return None, g, None, None, dwf, dbf, None, None, None, None
        dnw, dnb = _G(pnw if n_norms > 0 else None), _G(pnb if n_norms > 0 else None)
        for i in reversed(range(n_norms)):
            xin, mean, rstd = sn[3 * i:3 * i + 3]
            g = ops.norm_bwd(g, xin, nw, mean, rstd, mode, eps, None, F32, dnw.t, dnb.t)
        if n_norms == 0:
            g = ops.cast(g, F32)
        return g, None, dnw.out(), dnb.out(), dwf, dbf, None, None, None, None


def decoder_head(x, nw, nb, wff, bff, n_norms=1, mode='layer_norm', eps=1e-5, return_logits=False, prenormed=None):
    return HeadFn.apply(x, prenormed, nw, nb, wff, bff, n_norms, mode, eps, return_logits)


# =================================================================================================
# decoder head + CTC loss as one operator: [norm] -> norm -> Linear -> (log_softmax + CTCLoss)
#   sconformer_xl.py:246-247, decoder.py:22-26, exp/train.py:104,249
# =================================================================================================
class HeadCTCFn(Function):
    """Per-sample CTC negative log-likelihoods straight from the encoder output.  The log-probabilities are never written: the
    forward folds log_softmax into the CTC emission gather (row log-sum-exp kept), the backward produces d nll / d logits in bf16
    in one pass (CTC gradient through log_softmax, with the decoder bias gradient as its column sums).  Against the separate
    operators that is 25 GB less HBM traffic per step at the benchmark size (log_softmax forward, emission gather, CTC gradient
    and log_softmax backward: 41 GB -> 15 GB around the (B,N,4096) f32 tensors)."""

    @staticmethod
    def forward(ctx, x, hn_pre, nw, nb, wff, bff, n_norms: int, mode: str, eps: float, B: int, targets, input_lengths, target_lengths, blank: int):
        saved_norm = []
        if hn_pre is not None:                                                         # as in HeadFn
            hn, n_norms = hn_pre.contiguous(), -1
        else:
            x = x.contiguous()
            cur = x
            for i in range(n_norms):                                                   # legacy double norm: applied twice
                out_dt = BF16 if i == n_norms - 1 else F32
                y, mean, rstd = ops.norm_fwd(cur, nw, nb, mode, eps, out_dt)
                saved_norm += [cur, mean, rstd]
                cur = y
            hn = cur if n_norms > 0 else ops.cast(x, BF16)
        logits = ops.gemm(hn, wcast(wff), 'nt', bias=bff, out_dtype=F32)               # (B N, V+1) f32
        lg3 = logits.view(B, -1, logits.shape[-1])
        nll, ws = ops.ctc_fwd_logits(lg3, targets, input_lengths, target_lengths, blank)
        ctx.save_for_backward(nw, nb, wcast_t(wff), bff, hn, logits, nll, targets, input_lengths, target_lengths,
                              *[w for w in ws if w is not None], *saved_norm)
        ctx.cfg = (n_norms, mode, eps, B, blank, sum(w is not None for w in ws))
        ctx.P = (nw, nb, wff, bff)
        return nll

    @staticmethod
    def backward(ctx, dnll):
        n_norms, mode, eps, B, blank, nws = ctx.cfg
        pnw, pnb, pwf, pbf = ctx.P
        sv = ctx.saved_tensors
        nw, nb, wft, bff, hn, logits, nll, targets, input_lengths, target_lengths = sv[:10]
        ws = tuple(sv[10:10 + nws]) if nws else (None,) * 5
        sn = sv[10 + nws:]
        gbf = _G(pbf)
        dl = ops.ctc_bwd_logits(logits.view(B, -1, logits.shape[-1]), ws, nll, targets, input_lengths, target_lengths,
                                dnll.contiguous().to(F32), blank, colsum_into=gbf.t if pbf is not None else None)
        dl = dl.view(logits.shape)
        dbf = gbf.out()
        dwf = _wgrad(dl, hn, pwf)
        g = ops.gemm(dl, wft, 'nt')                                                    # (M,d) bf16
        if n_norms < 0:
            return (None, g, None, None, dwf, dbf) + (None,) * 8
        dnw, dnb = _G(pnw if n_norms > 0 else None), _G(pnb if n_norms > 0 else None)
        for i in reversed(range(n_norms)):
            xin, mean, rstd = sn[3 * i:3 * i + 3]
            g = ops.norm_bwd(g, xin, nw, mean, rstd, mode, eps, None, F32, dnw.t, dnb.t)
        if n_norms == 0:
            g = ops.cast(g, F32)
        return (g, None, dnw.out(), dnb.out(), dwf, dbf) + (None,) * 8


def decoder_head_ctc(x, nw, nb, wff, bff, B, targets, input_lengths, target_lengths, blank, n_norms=1, mode='layer_norm', eps=1e-5, prenormed=None,
                     num_labels=None):
    """(B,) CTC negative log-likelihoods of the head applied to x (B N, d); integer tensors as for ctc_nll.
    num_labels: the number of real classes when wff carries padding rows (labels must stay below it)."""
    dev = x.device
    _check_ctc_host_args(targets, input_lengths, target_lengths, x.shape[0] // max(B, 1), wff.shape[0] if num_labels is None else num_labels)
    tg = targets.to(device=dev, dtype=torch.int32).contiguous()
    il = input_lengths.to(device=dev, dtype=torch.int32).contiguous()
    tl = target_lengths.to(device=dev, dtype=torch.int32).contiguous()
    return HeadCTCFn.apply(x, prenormed, nw, nb, wff, bff, n_norms, mode, eps, B, tg, il, tl, blank)


# =================================================================================================
# ConvSubsampling 'dw_striding' x8  — subsampling.py:276-321, 384-428
# =================================================================================================
class SubsampleFn(Function):
    @staticmethod
    def forward(ctx, audio, w0, b0, wd1, bd1, wp1, bp1, wd2, bd2, wp2, bp2, wout, bout):
        audio = audio.contiguous()
        B = audio.shape[0]
        C = w0.shape[0]
        w0f, wd1f, wd2f = (t.detach().reshape(C, 9).contiguous() for t in (w0, wd1, wd2))
        wp1h, wp2h, woh = wcast(wp1), wcast(wp2), wcast(wout)
        d1 = ops.sub_stage01_fwd(audio, w0f, b0, wd1f, bd1)                            # (B,T4,F4,C); stage 0 never hits HBM
        pre1 = ops.gemm(d1.view(-1, C), wp1h, 'nt', bias=bp1).view(d1.shape)
        d2 = ops.sub_dwconv_fwd(pre1, wd2f, bd2)                                       # (B,N,F8,C)
        pre2 = ops.gemm(d2.view(-1, C), wp2h, 'nt', bias=bp2).view(d2.shape)
        N, F8 = d2.shape[1], d2.shape[2]
        s = ops.sub_silu_transpose(pre2.view(B * N, F8, C))                            # (B*N, C*F8)
        x = ops.gemm(s, woh, 'nt', bias=bout, out_dtype=F32)
        ctx.save_for_backward(audio, w0f, wd1f, wd2f, wcast_t(wp1), wcast_t(wp2), wcast_t(wout), bp1, bp2, bout, b0, d1, pre1, d2, pre2, s)
        ctx.P = (w0, b0, wd1, bd1, wp1, bp1, wd2, bd2, wp2, bp2, wout, bout)
        return x.view(B, N, -1)

    @staticmethod
    def backward(ctx, dx):
        pw0, pb0, pwd1, pbd1, pwp1, pbp1, pwd2, pbd2, pwp2, pbp2, pwo, pbo = ctx.P
        audio, w0f, wd1f, wd2f, wp1t, wp2t, wot, bp1, bp2, bout, b0, d1, pre1, d2, pre2, s = ctx.saved_tensors
        B, N, F8, C = d2.shape
        dev = dx.device
        dx16 = ops.cast(dx.contiguous().view(B * N, -1), BF16)
        ds = ops.gemm(dx16, wot, 'nt')                                                 # (B*N, C*F8)
        dwo = _wgrad(dx16, s, pwo)
        dbo = _bgrad(dx16, pbo)
        dpre2 = ops.sub_silu_transpose(pre2.view(B * N, F8, C), ds).view(-1, C)
        dwp2 = _wgrad(dpre2, d2.view(-1, C), pwp2)
        dbp2 = _bgrad(dpre2, pbp2)
        dd2 = ops.gemm(dpre2, wp2t, 'nt').view(d2.shape)
        dwd2, dbd2 = _G(pwd2, (C, 9)), _G(pbd2)
        gbp1 = _G(pbp1)                                                                # its bias gradient = column sums of dpre1: same pass
        dpre1 = ops.sub_dwconv_bwd(dd2, wd2f, pre1, dwd2.t, dbd2.t, colsum_into=gbp1.t).view(-1, C)
        dwp1 = _wgrad(dpre1, d1.view(-1, C), pwp1)
        dbp1 = gbp1.out()
        dd1 = ops.gemm(dpre1, wp1t, 'nt').view(d1.shape)
        dwd1, dbd1, dw0, db0 = _G(pwd1, (C, 9)), _G(pbd1), _G(pw0, (C, 9)), _G(pb0)
        ops.sub_stage01_bwd_(dd1, audio, w0f, b0, wd1f, dw0.t, db0.t, dwd1.t, dbd1.t)  # conv0 recomputed; no (B,T/2,F/2,C) grads
        return (None, dw0.out(), db0.out(), dwd1.out(), dbd1.out(), dwp1, dbp1, dwd2.out(), dbd2.out(), dwp2, dbp2, dwo, dbo)


def subsample(audio, w0, b0, wd1, bd1, wp1, bp1, wd2, bd2, wp2, bp2, wout, bout):
    return SubsampleFn.apply(audio, w0, b0, wd1, bd1, wp1, bp1, wd2, bd2, wp2, bp2, wout, bout)


# =================================================================================================
# CTC  — torch.nn.CTCLoss(blank, reduction) seam, exp/train.py:104,249
# =================================================================================================
class CTCFn(Function):
    @staticmethod
    def forward(ctx, log_probs_bnc, targets, input_lengths, target_lengths, blank: int):
        lp = log_probs_bnc.contiguous()
        nll, ws = ops.ctc_fwd(lp, targets, input_lengths, target_lengths, blank)
        ctx.save_for_backward(lp, nll, targets, input_lengths, target_lengths, *[w for w in ws if w is not None])
        ctx.blank = blank
        return nll

    @staticmethod
    def backward(ctx, dnll):
        sv = ctx.saved_tensors
        lp, nll, targets, input_lengths, target_lengths = sv[:5]
        ws = tuple(sv[5:]) if len(sv) > 5 else (None,) * 4
        g = ops.ctc_bwd(lp, ws, nll, targets, input_lengths, target_lengths, dnll.contiguous().to(F32), ctx.blank)
        return g, None, None, None, None


def _check_ctc_host_args(targets, input_lengths, target_lengths, N: int, C: int) -> None:
    if not input_lengths.is_cuda and input_lengths.numel() and int(input_lengths.max()) > N:
        raise ValueError(f'CTC: input_lengths must not exceed the {N} time steps of log_probs')
    if not target_lengths.is_cuda and target_lengths.numel() and (int(target_lengths.max()) > targets.shape[1] or int(target_lengths.min()) < 0):
        raise ValueError('CTC: target_lengths out of range for the targets tensor')
    if not targets.is_cuda and targets.numel() and (int(targets.min()) < 0 or int(targets.max()) >= C):
        raise ValueError(f'CTC: target labels must be in [0, {C})')


def ctc_nll(log_probs_bnc, targets, input_lengths, target_lengths, blank: int) -> torch.Tensor:
    """Per-sample negative log-likelihoods (B,) from batch-major (B,N,C) f32 log-probs.
    Arguments torch.nn.CTCLoss rejects (input_length > N, target_length > targets.shape[1], a label outside [0, C)) raise
    here when the tensors are on the host (checking costs nothing); for device tensors a check would stall the launch queue
    for a whole forward, so the kernels poison the sample instead (nll and its gradient rows NaN - the optimiser skips the step)."""
    dev = log_probs_bnc.device
    _B, _N, _C = log_probs_bnc.shape
    _check_ctc_host_args(targets, input_lengths, target_lengths, _N, _C)
    if _C % 4:                                            # the kernels move 4 classes per access: pad with impossible classes (p = 0)
        log_probs_bnc = torch.nn.functional.pad(log_probs_bnc, (0, 4 - _C % 4), value=-1e30)
    tg = targets.to(device=dev, dtype=torch.int32).contiguous()
    il = input_lengths.to(device=dev, dtype=torch.int32).contiguous()
    tl = target_lengths.to(device=dev, dtype=torch.int32).contiguous()
    return CTCFn.apply(log_probs_bnc, tg, il, tl, blank)
