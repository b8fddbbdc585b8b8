"""Tensor-level wrappers over the sconf C ABI (include/sconf.h).

Every function takes/returns torch tensors resident on the GPU, allocates its outputs, and enqueues the
HIP kernels on torch's CURRENT stream.  torch is used only for device memory and streams.  There is no
fallback path: a missing library or a failing call raises RuntimeError.

`tests/kernel_refs.py` holds a plain-PyTorch fp32 reference with the same signature for every function
here; the GPU parity tests compare the two.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib

F32, BF16 = 0, 1
ACT = {'none': 0, 'gelu': 1, 'silu': 2, 'dgelu': 3, 'dsilu': 4, 'gelu_dsave': 5, 'mulaux': 6}
LAYOUT = {'nt': 0, 'nn': 1, 'tn': 2}
NORM_MODE = {'layer_norm': 0, 'rms_norm': 1, 'rms_norm_apex': 2}


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f'unsupported dtype {t.dtype}')


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu(t: torch.Tensor, what: str = 'input') -> None:
    """The product path is HIP-only: refuse CPU tensors loudly (there is no CPU fallback)."""
    if not t.is_cuda:
        raise RuntimeError(f'{what} must live on the GPU: lcasr_amd runs on the MI355X HIP path only (no CPU fallback)')
    _lib.load()


def _chk(t: torch.Tensor, name: str, dtype=None):
    if not t.is_cuda:
        raise RuntimeError(f'{name} must be a GPU tensor (the HIP path has no CPU fallback)')
    if not t.is_contiguous():
        raise RuntimeError(f'{name} must be contiguous')
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f'{name} must be {dtype}, got {t.dtype}')
    return t


# ------------------------------------------------------------------------------------------------
# GEMM
# ------------------------------------------------------------------------------------------------
def gemm(a: torch.Tensor, b: torch.Tensor, layout: str = 'nt', bias: Optional[torch.Tensor] = None,
         resid: Optional[torch.Tensor] = None, aux: Optional[torch.Tensor] = None, act: str = 'none',
         alpha: float = 1.0, out_dtype: torch.dtype = torch.bfloat16, save_pre: bool = False, split_k: int = 1,
         accum: Optional[torch.Tensor] = None):
    """C[M,N] = resid + alpha * act(A·B + bias)   (bf16 operands, f32 accumulate).

    layout 'nt': a (M,K), b (N,K) — y = x W^T;  'nn': a (M,K), b (K,N) — dx = dy W;  'tn': a (K,M), b (K,N) — dW = dy^T x.
    act 'dgelu'/'dsilu' multiplies by the activation derivative evaluated at aux (M,N) bf16.
    save_pre returns (C, pre) with pre = A·B + bias in bf16 (act 'gelu_dsave': pre = gelu'(A·B + bias), the factor the
    backward multiplies by with act 'mulaux').  split_k > 1 needs out_dtype float32.
    accum (f32 (M,N)): the product is ADDED into it in place (accum += alpha * A·B) and returned - weight gradients going
    straight into the optimiser's flat gradient buffer.
    """
    _chk(a, 'a', torch.bfloat16); _chk(b, 'b', torch.bfloat16)
    if layout == 'nt':
        M, K = a.shape; N, K2 = b.shape
    elif layout == 'nn':
        M, K = a.shape; K2, N = b.shape
    else:
        K, M = a.shape; K2, N = b.shape
    if K != K2:
        raise ValueError(f'gemm {layout}: inner dims differ: {tuple(a.shape)} x {tuple(b.shape)}')
    if accum is not None:
        _chk(accum, 'accum', torch.float32)
        if accum.numel() != M * N or bias is not None or resid is not None or act != 'none' or save_pre:
            raise ValueError('gemm accum: needs an (M,N) f32 tensor and the plain epilogue')
        out_dtype = torch.float32
    out_f32 = out_dtype == torch.float32
    splits = _lib.load().sconf_gemm_num_splits(K, int(split_k)) if split_k > 1 else 1
    one_slab = False
    if accum is not None and splits == 1 and layout == 'tn':
        # in-place accumulation is a residual epilogue, which only the 128x128 kernel has for K-strided operands: when the 256-row TN
        # kernel would take the plain problem, write one slab and add it with the split-K reduce (config 5's weight gradients: 13 per step)
        lib = _lib.load()
        one_slab = (lib.sconf_gemm_variant(2, M, N, K, a.stride(0), b.stride(0), 1, 0, 0, 0) == 3 and
                    lib.sconf_gemm_variant(2, M, N, K, a.stride(0), b.stride(0), 1, 0, 1, 0) != 3)
    if splits > 1 or one_slab:                                         # deterministic split-K: partial slabs + fixed-order reduce
        c = torch.empty(splits, M, N, dtype=torch.float32, device=a.device)
    elif accum is not None:
        c = resid = accum                                              # out = accum + alpha * A·B, element-wise in place
    else:
        c = torch.empty(M, N, dtype=out_dtype, device=a.device)
    pre = torch.empty(M, N, dtype=torch.bfloat16, device=a.device) if save_pre else None
    if bias is not None: _chk(bias, 'bias', torch.float32)
    if resid is not None:
        _chk(resid, 'resid', torch.float32)
        if tuple(resid.shape) != (M, N): raise ValueError('resid shape mismatch')
    if aux is not None:
        _chk(aux, 'aux', torch.bfloat16)
        if tuple(aux.shape) != (M, N): raise ValueError('aux shape mismatch')
    _lib.call('sconf_gemm_bf16', LAYOUT[layout], _p(a), _p(b), _p(c), M, N, K, a.stride(0), b.stride(0), N,
              _p(bias), _p(resid), N, _p(aux), N, _p(pre), N, float(alpha), ACT[act], int(out_f32), int(split_k), _stream())
    if splits > 1 or one_slab:
        out = accum if accum is not None else torch.empty(M, N, dtype=torch.float32, device=a.device)
        _lib.call('sconf_splitk_reduce', _p(c), _p(out), splits, M * N, int(accum is not None), _stream())
        c = out
    return (c, pre) if save_pre else c


def gemm_qkv_rotary(x: torch.Tensor, w_regrouped: torch.Tensor, bias: Optional[torch.Tensor], cos: torch.Tensor, sin: torch.Tensor,
                    N: int, H: int, D: int) -> torch.Tensor:
    """qkv projection + NeoX rotary in one launch: x (M, K) bf16, w_regrouped (3 H D, K) bf16 with rows [q | k | v] (bias regrouped
    alike, f32) -> (M, 3 H D) bf16 = (B, N, 3, H, D) with q, k rotated for position (row % N); cos / sin (N, D/2) f32."""
    _chk(x, 'x', torch.bfloat16); _chk(w_regrouped, 'w', torch.bfloat16); _chk(cos, 'cos', torch.float32); _chk(sin, 'sin', torch.float32)
    M, K = x.shape
    if tuple(w_regrouped.shape) != (3 * H * D, K): raise ValueError('gemm_qkv_rotary: weight shape')
    if tuple(cos.shape) != (N, D // 2) or tuple(sin.shape) != (N, D // 2): raise ValueError('gemm_qkv_rotary: rotary tables must be (N, D/2)')
    if bias is not None: _chk(bias, 'bias', torch.float32)
    out = torch.empty(M, 3 * H * D, dtype=torch.bfloat16, device=x.device)
    _lib.call('sconf_gemm_qkv_rotary', _p(x), _p(w_regrouped), _p(out), M, K, H, D, x.stride(0), w_regrouped.stride(0), _p(bias),
              _p(cos), _p(sin), N, _stream())
    return out


def pick_split_k(M: int, N: int, K: int, n_cus: int = 256) -> int:
    """Split-K factor for weight-gradient GEMMs (few output tiles, very long K).

    The GEMM runs at most 2 persistent workgroups per CU (`slots`) that walk the tiles x s work items, each item being
    ceil(nkt / s) K-steps plus an epilogue of f32 atomics worth ~3 K-steps.  Pick the s that minimises the critical path."""
    nkt = (K + 63) // 64
    if M % 256 == 0 and N % 256 == 0 and K % 64 == 0:
        # 256x256-tile kernel (gemm256.hip): one workgroup per CU, so aim at one full round of (tile, split) items; with
        # very few tiles the per-split slabs get too many and the 128x128 kernel below does better
        tiles256 = (M // 256) * (N // 256)
        if tiles256 <= n_cus and nkt >= 32:
            sk = max(1, min(n_cus // tiles256, nkt // 32))               # at least 32 K-tiles per work item
            if tiles256 * sk * 4 >= 3 * n_cus:                           # the kernel's own eligibility rule: >= 3/4 of a round
                return sk
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    slots = 2 * n_cus
    if tiles >= slots or nkt < 32:
        return 1
    best, best_cost = 1, None
    for s_ in range(1, min(128, nkt // 8) + 1):
        items = tiles * s_                                   # (tile, split) work items, walked by <= slots workgroups
        cost = -(-items // min(items, slots)) * (-(-nkt // s_) + (3 if s_ > 1 else 1))
        if best_cost is None or cost < best_cost:
            best, best_cost = s_, cost
    return best


# ------------------------------------------------------------------------------------------------
# norms
# ------------------------------------------------------------------------------------------------
def norm_fwd(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], mode: str, eps: float,
             out_dtype: torch.dtype) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Row norm over the last dim.  Returns (y, mean, rstd) with f32 per-row statistics."""
    _chk(x, 'x'); _chk(weight, 'weight', torch.float32)
    d = x.shape[-1]; M = x.numel() // d
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    mean = torch.empty(M, dtype=torch.float32, device=x.device)
    rstd = torch.empty(M, dtype=torch.float32, device=x.device)
    _lib.call('sconf_norm_fwd', NORM_MODE[mode], _p(x), _dt(x), _p(weight), _p(bias), _p(y), _dt(y), _p(mean), _p(rstd),
              M, d, float(eps), _stream())
    return y, mean, rstd


def norm_bwd(dy: torch.Tensor, x: torch.Tensor, weight: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor, mode: str,
             eps: float, dres: Optional[torch.Tensor], dx_dtype: torch.dtype, dweight: torch.Tensor,
             dbias: Optional[torch.Tensor], twin: bool = False):
    """dx = dres + norm'(x)·dy;  dweight / dbias are accumulated in place (f32).
    twin=True (f32 dx only) also returns (dx16, colsum): a bf16 copy of dx and its column sums (d,) f32, produced in the
    same pass for the block that receives dx as its output gradient."""
    _chk(dy, 'dy'); _chk(x, 'x'); _chk(dweight, 'dweight', torch.float32)
    d = x.shape[-1]; M = x.numel() // d
    if dres is not None: _chk(dres, 'dres', torch.float32)
    dx = torch.empty(x.shape, dtype=dx_dtype, device=x.device)
    nws = int(_lib.load().sconf_norm_bwd_workspace(M, d))               # per-workgroup column sums (no atomics, fixed order)
    ws = torch.empty(nws, dtype=torch.float32, device=x.device)
    twin = twin and dx_dtype == torch.float32
    dx16 = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if twin else None
    cs = torch.empty(d, dtype=torch.float32, device=x.device) if twin else None
    _lib.call('sconf_norm_bwd', NORM_MODE[mode], _p(dy), _dt(dy), _p(x), _dt(x), _p(weight), _p(mean), _p(rstd), _p(dres),
              _p(dx), _dt(dx), _p(dweight), _p(dbias), _p(ws), nws, _p(dx16), _p(cs), M, d, float(eps), _stream())
    return (dx, dx16, cs) if twin else dx


def norm2_fwd(x: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor, eps1: float, eps2: float,
              twice: bool = False):
    """y1 = LayerNorm(x; w1, b1) (f32) and h2 = LayerNorm(y1; w2, b2) (bf16) in one pass (d <= 768); twice: the second norm is
    applied two times (the head's legacy double norm).  Returns (y1, h2, stats): stats = (mean1, rstd1, mean2, rstd2[, mean3, rstd3])."""
    _chk(x, 'x', torch.float32)
    for n, t in (('w1', w1), ('b1', b1), ('w2', w2), ('b2', b2)): _chk(t, n, torch.float32)
    d = x.shape[-1]; M = x.numel() // d
    y1 = torch.empty_like(x)
    h2 = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    ns = 6 if twice else 4
    st = torch.empty(ns, M, dtype=torch.float32, device=x.device)
    _lib.call('sconf_norm2_fwd', _p(x), _p(w1), _p(b1), _p(w2), _p(b2), _p(y1), _p(h2), _p(st[0]), _p(st[1]), _p(st[2]), _p(st[3]),
              _p(st[4]) if twice else None, _p(st[5]) if twice else None, int(twice), M, d, float(eps1), float(eps2), _stream())
    return y1, h2, tuple(st[i] for i in range(ns))


def norm2_bwd(dh2: torch.Tensor, x: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor, stats,
              dres: Optional[torch.Tensor], dw1: torch.Tensor, db1: torch.Tensor, dw2: torch.Tensor, db2: torch.Tensor, twin: bool = False):
    """Backward of norm2_fwd (six statistics: the twice form): dx = LN1'(dres + LN2'(dh2)); the four parameter gradients are
    accumulated in place (f32).  twin=True also returns (dx16, colsum) as norm_bwd does."""
    _chk(dh2, 'dh2', torch.bfloat16); _chk(x, 'x', torch.float32)
    for n, t in (('dw1', dw1), ('db1', db1), ('dw2', dw2), ('db2', db2)): _chk(t, n, torch.float32)
    if dres is not None: _chk(dres, 'dres', torch.float32)
    d = x.shape[-1]; M = x.numel() // d
    twice = len(stats) == 6
    dx = torch.empty_like(x)
    nws = int(_lib.load().sconf_norm2_bwd_workspace(M, d))
    ws = torch.empty(nws, dtype=torch.float32, device=x.device)
    dx16 = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if twin else None
    cs = torch.empty(d, dtype=torch.float32, device=x.device) if twin else None
    _lib.call('sconf_norm2_bwd', _p(dh2), _p(x), _p(w1), _p(b1), _p(w2), _p(b2), _p(stats[0]), _p(stats[1]), _p(stats[2]), _p(stats[3]),
              _p(stats[4]) if twice else None, _p(stats[5]) if twice else None, int(twice), _p(dres), _p(dx),
              _p(dw1), _p(db1), _p(dw2), _p(db2), _p(ws), nws, _p(dx16), _p(cs), M, d, _stream())
    return (dx, dx16, cs) if twin else dx


# ------------------------------------------------------------------------------------------------
# elementwise / rows
# ------------------------------------------------------------------------------------------------
def cast(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    _chk(x, 'x')
    if x.dtype == dtype:
        return x
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    _lib.call('sconf_cast', _p(x), _dt(x), _p(y), _dt(y), x.numel(), _stream())
    return y


def cast_transpose(w: torch.Tensor) -> torch.Tensor:
    """f32 (R,C) -> bf16 (C,R)."""
    _chk(w, 'w', torch.float32)
    R, Cc = w.shape
    out = torch.empty(Cc, R, dtype=torch.bfloat16, device=w.device)
    _lib.call('sconf_cast_transpose', _p(w), _p(out), R, Cc, _stream())
    return out


def cast_shadows(table: torch.Tensor, n_entries: int, total_tiles: int) -> None:
    """Refresh every bf16 weight shadow listed in `table` (device int64 (n_entries + 1, 6), see sconf.h) in one launch."""
    _chk(table, 'table', torch.int64)
    _lib.call('sconf_cast_shadows', _p(table), int(n_entries), int(total_tiles), _stream())


def rotary_qkv_fwd(qkv: torch.Tensor, cos: Optional[torch.Tensor], sin: Optional[torch.Tensor], B: int, N: int, H: int, D: int):
    """qkv (B*N, H*D*3) bf16 in the reference's "(h d qkv)" column order -> q,k,v (B,N,H,D) bf16, rotary on q,k.
    cos/sin: (N, D/2) f32 or None (no rotary)."""
    _chk(qkv, 'qkv', torch.bfloat16)
    q = torch.empty(B, N, H, D, dtype=torch.bfloat16, device=qkv.device)
    k = torch.empty_like(q); v = torch.empty_like(q)
    use = cos is not None
    if use: _chk(cos, 'cos', torch.float32); _chk(sin, 'sin', torch.float32)
    _lib.call('sconf_rotary_qkv', 0, _p(qkv), _p(cos), _p(sin), _p(q), _p(k), _p(v), B, N, H, D, int(use), _stream())
    return q, k, v


def rotary_qkv_bwd(dq: torch.Tensor, dk: torch.Tensor, dv: torch.Tensor, cos, sin, B: int, N: int, H: int, D: int) -> torch.Tensor:
    """(dq,dk,dv) (B,N,H,D) bf16 -> dqkv (B*N, H*D*3) bf16 (transpose of rotary_qkv_fwd)."""
    for t, n in ((dq, 'dq'), (dk, 'dk'), (dv, 'dv')): _chk(t, n, torch.bfloat16)
    dqkv = torch.empty(B * N, H * D * 3, dtype=torch.bfloat16, device=dq.device)
    use = cos is not None
    _lib.call('sconf_rotary_qkv', 1, _p(dqkv), _p(cos), _p(sin), _p(dq), _p(dk), _p(dv), B, N, H, D, int(use), _stream())
    return dqkv


def rotary_inplace_(qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, B: int, N: int, H: int, D: int) -> torch.Tensor:
    """NeoX rotary on the q and k blocks of a regrouped qkv activation (B*N, 3*H*D) = (B,N,3,H,D) bf16, in place."""
    _chk(qkv, 'qkv', torch.bfloat16); _chk(cos, 'cos', torch.float32); _chk(sin, 'sin', torch.float32)
    if qkv.numel() != B * N * 3 * H * D or tuple(cos.shape) != (N, D // 2): raise ValueError('rotary_inplace_: shape mismatch')
    _lib.call('sconf_rotary_inplace', _p(qkv), _p(cos), _p(sin), B, N, H, D, _stream())
    return qkv


def softmax_fwd(x: torch.Tensor, log: bool, out_dtype: torch.dtype) -> torch.Tensor:
    _chk(x, 'x')
    Cn = x.shape[-1]; M = x.numel() // Cn
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    _lib.call('sconf_softmax_fwd', int(log), _p(x), _dt(x), _p(y), _dt(y), M, Cn, _stream())
    return y


def softmax_bwd(y: torch.Tensor, dy: torch.Tensor, log: bool, out_dtype: torch.dtype,
                colsum_into: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Backward of softmax / log_softmax over the last dim.  colsum_into (f32, C elements): += column sums of the returned dx
    (the bias gradient of the Linear that produced the logits), computed in the same pass."""
    _chk(y, 'y'); _chk(dy, 'dy')
    Cn = y.shape[-1]; M = y.numel() // Cn
    dx = torch.empty(y.shape, dtype=out_dtype, device=y.device)
    ws = None
    if colsum_into is not None:
        _chk(colsum_into, 'colsum_into', torch.float32)
        if colsum_into.numel() != Cn: raise ValueError('colsum_into must have one element per class')
        ws = torch.empty(int(_lib.load().sconf_softmax_bwd_workspace(M, Cn)), dtype=torch.float32, device=y.device)
    _lib.call('sconf_softmax_bwd', int(log), _p(y), _dt(y), _p(dy), _dt(dy), _p(dx), _dt(dx), _p(colsum_into), _p(ws), M, Cn, _stream())
    return dx


def rowdot(a: torch.Tensor, b: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[m] = sum_c a[m, c] * (b[m, c] - bias[c])  (a, b bf16 (M, d); bias f32 (d) or None) -> (M,) f32."""
    _chk(a, 'a', torch.bfloat16); _chk(b, 'b', torch.bfloat16)
    if a.shape != b.shape or a.dim() != 2: raise ValueError('rowdot: a and b must be (M, d) of one shape')
    if bias is not None: _chk(bias, 'bias', torch.float32)
    M, d = a.shape
    out = torch.empty(M, dtype=torch.float32, device=a.device)
    _lib.call('sconf_rowdot', _p(a), _p(b), _p(bias), _p(out), M, d, a.stride(0), b.stride(0), _stream())
    return out


def gemm_softmax_bwd_eligible(M: int, V: int, K: int) -> bool:
    """Whether gemm_softmax_bwd has a kernel for (M, K) x (V, K)^T (the 256-row NT kernels; same decision code as the launch)."""
    return int(_lib.load().sconf_gemm_variant(0, M, V, K, K, K, 1, 7, 0, 0)) in (1, 2)


def gemm_softmax_bwd(dy: torch.Tensor, wt: torch.Tensor, probs: torch.Tensor, delta: torch.Tensor,
                     colsum_into: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dl = probs * (dy @ wt.T - delta[:, None]) in bf16 without writing dy @ wt.T (softmax backward in the GEMM epilogue);
    colsum_into (f32, V) += column sums of dl.  dy (M, K), wt (V, K), probs (M, V) bf16; delta (M,) f32 = sum_v probs * (dy @ wt.T)."""
    _chk(dy, 'dy', torch.bfloat16); _chk(wt, 'wt', torch.bfloat16); _chk(probs, 'probs', torch.bfloat16); _chk(delta, 'delta', torch.float32)
    M, K = dy.shape; V = wt.shape[0]
    if tuple(probs.shape) != (M, V) or wt.shape[1] != K or delta.numel() != M: raise ValueError('gemm_softmax_bwd: shapes')
    if not gemm_softmax_bwd_eligible(M, V, K): raise ValueError(f'gemm_softmax_bwd: no kernel for {M} x {V} x {K} (check gemm_softmax_bwd_eligible)')
    dl = torch.empty(M, V, dtype=torch.bfloat16, device=dy.device)
    slab = torch.empty(2 * (M // 256), V, dtype=torch.float32, device=dy.device)
    _lib.call('sconf_gemm_softmax_bwd', _p(dy), _p(wt), _p(probs), _p(delta), _p(dl), _p(slab), M, V, K, dy.stride(0), wt.stride(0),
              probs.stride(0), _stream())
    if colsum_into is not None:
        _chk(colsum_into, 'colsum_into', torch.float32)
        colsum_(slab, colsum_into)
    return dl


def colsum_(x: torch.Tensor, out: torch.Tensor, alpha: float = 1.0) -> torch.Tensor:
    """out[n] += alpha * sum_m x[m, n]  (in place, f32)."""
    _chk(x, 'x'); _chk(out, 'out', torch.float32)
    N = x.shape[-1]; M = x.numel() // N
    _lib.call('sconf_colsum', _p(x), _dt(x), _p(out), M, N, N, float(alpha), _stream())
    return out


def mask_rows_(x: torch.Tensor, lengths: torch.Tensor, B: int, N: int) -> torch.Tensor:
    """Zero rows n >= lengths[b] of x viewed as (B,N,d), in place."""
    _chk(x, 'x'); _chk(lengths, 'lengths', torch.int32)
    d = x.numel() // (B * N)
    _lib.call('sconf_mask_rows', _p(x), _dt(x), _p(lengths), B, N, d, _stream())
    return x


# ------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------
def _strides3(t: torch.Tensor):
    if t.stride(3) != 1:
        raise RuntimeError('attention operands need a contiguous head_dim')
    return (C.c_int64 * 3)(t.stride(0), t.stride(1), t.stride(2))


def attn_fwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, lengths: Optional[torch.Tensor], window=(-1, -1),
             scale: Optional[float] = None):
    """q,k,v (B,N,H,D) bf16 (may be strided views) -> o (B,N,H,D) bf16, lse (B,H,N) f32."""
    B, N, H, D = q.shape
    o = torch.empty(B, N, H, D, dtype=torch.bfloat16, device=q.device)
    lse = torch.empty(B, H, N, dtype=torch.float32, device=q.device)
    if lengths is not None: _chk(lengths, 'lengths', torch.int32)
    sc = float(scale) if scale is not None else D ** -0.5
    _lib.call('sconf_attn_fwd', _p(q), _p(k), _p(v), _p(o), _p(lse), _p(lengths), B, N, H, D, _strides3(q), _strides3(k),
              _strides3(v), _strides3(o), int(window[0]), int(window[1]), sc, _stream())
    return o, lse


def attn_bwd(q, k, v, o, dout, lse, lengths, window=(-1, -1), scale: Optional[float] = None, rot=None, out=None):
    """-> (dq, dk, dv) bf16 (B,N,H,D).  out: three (B,N,H,D) bf16 views to write into (e.g. the blocks of one (B,N,3,H,D) buffer).
    rot = (cos, sin) f32 (N, D/2): q and k are rotary-rotated; dq and dk come back as gradients of the unrotated q, k."""
    B, N, H, D = q.shape
    if out is None:
        dq = torch.empty(B, N, H, D, dtype=torch.bfloat16, device=q.device)
        dk = torch.empty_like(dq); dv = torch.empty_like(dq)
    else:
        dq, dk, dv = out
    delta = torch.empty(2, B, H, N, dtype=torch.float32, device=q.device)     # scratch: the dQ kernel leaves [-delta | -lse log2(e)] for the dK/dV kernel
    sc = float(scale) if scale is not None else D ** -0.5
    cos, sin = rot if rot is not None else (None, None)
    if cos is not None:
        _chk(cos, 'cos', torch.float32); _chk(sin, 'sin', torch.float32)
        if tuple(cos.shape) != (N, D // 2): raise ValueError('rotary tables must be (N, D/2)')
    _lib.call('sconf_attn_bwd', _p(q), _p(k), _p(v), _p(o), _p(dout), _p(lse), _p(delta), _p(dq), _p(dk), _p(dv), _p(lengths),
              B, N, H, D, _strides3(q), _strides3(k), _strides3(v), _strides3(o), _strides3(dout), _strides3(dq), _strides3(dk),
              _strides3(dv), int(window[0]), int(window[1]), sc, _p(cos), _p(sin), _stream())
    return dq, dk, dv


# ------------------------------------------------------------------------------------------------
# conformer conv module
# ------------------------------------------------------------------------------------------------
def glu_dwconv_fwd(g: torch.Tensor, lengths: Optional[torch.Tensor], w: torch.Tensor, bias: torch.Tensor, B: int, N: int):
    """g (B*N, 2d) bf16 -> h (B*N, d) bf16 = dwconv_k(mask(GLU(g))) + bias; stats f64 (2,d) = [sum h, sum h^2]."""
    _chk(g, 'g', torch.bfloat16); _chk(w, 'w', torch.float32); _chk(bias, 'bias', torch.float32)
    d = g.shape[-1] // 2
    ks = w.numel() // d
    h = torch.empty(B * N, d, dtype=torch.bfloat16, device=g.device)
    stats = torch.zeros(2, d, dtype=torch.float64, device=g.device)
    _lib.call('sconf_glu_dwconv_fwd', _p(g), _p(lengths), _p(w), _p(bias), _p(h), _p(stats), B, N, d, ks, _stream())
    return h, stats


def brn_finalize(stats: torch.Tensor, count: int, running_mean, running_std, num_batches_tracked, weight, bias,
                 training: bool, eps: float = 1e-3, momentum: float = 0.01) -> torch.Tensor:
    """-> coef f32 (6,d): mean, s, r, d, A, Bc.  Training: updates the running buffers in place."""
    d = weight.numel()
    coef = torch.empty(6, d, dtype=torch.float32, device=weight.device)
    _chk(num_batches_tracked, 'num_batches_tracked', torch.int64)
    _lib.call('sconf_brn_finalize', _p(stats), int(count), _p(running_mean), _p(running_std), _p(num_batches_tracked), _p(weight),
              _p(bias), _p(coef), d, int(training), float(eps), float(momentum), _stream())
    return coef


def affine_silu_fwd(h: torch.Tensor, coef: torch.Tensor) -> torch.Tensor:
    _chk(h, 'h', torch.bfloat16)
    d = h.shape[-1]
    y = torch.empty_like(h)
    _lib.call('sconf_affine_silu_fwd', _p(h), _p(coef), _p(y), h.numel() // d, d, _stream())
    return y


def convmod_bwd(dy, h, g, lengths, w, brn_weight, coef, B: int, N: int, training: bool, eps: float,
                dw, dbias, dbrn_weight, dbrn_bias, colsum: bool = False):
    """Backward of [GLU -> mask -> dwconv -> BatchRenorm -> SiLU]; returns dg (B*N, 2d) bf16; accumulates the 4 grads.
    colsum=True also returns the column sums of dg (2d,) f32 from the same pass (the bias gradient of the producing 1x1 conv)."""
    _chk(dy, 'dy', torch.bfloat16); _chk(h, 'h', torch.bfloat16); _chk(g, 'g', torch.bfloat16)
    d = h.shape[-1]; ks = w.numel() // d
    dg = torch.empty(B * N, 2 * d, dtype=torch.bfloat16, device=dy.device)
    red = torch.zeros(2, d, dtype=torch.float64, device=dy.device)
    bcoef = torch.empty(3, d, dtype=torch.float32, device=dy.device)
    dwt = torch.zeros(ks, d, dtype=torch.float32, device=dy.device)       # kernel accumulates the tap gradients as [k][d]
    cs = torch.zeros(2 * d, dtype=torch.float32, device=dy.device) if colsum else None
    _lib.call('sconf_convmod_bwd', _p(dy), _p(h), _p(g), _p(lengths), _p(w), _p(brn_weight), _p(coef), _p(red), _p(bcoef), _p(dg),
              _p(dwt), _p(dbias), _p(dbrn_weight), _p(dbrn_bias), _p(cs), B, N, d, ks, int(training), float(eps), _stream())
    dw.add_(dwt.t())
    return (dg, cs) if colsum else dg


# ------------------------------------------------------------------------------------------------
# subsampler
# ------------------------------------------------------------------------------------------------
def _half(n: int) -> int:
    return (n - 1) // 2 + 1


def sub_conv0_fwd(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """x (B,F,T) f32/bf16 -> pre0 (B,T2,F2,C) bf16 (pre-activation)."""
    _chk(x, 'x'); _chk(w, 'w', torch.float32)
    B, F, T = x.shape; Cc = w.shape[0]
    y = torch.empty(B, _half(T), _half(F), Cc, dtype=torch.bfloat16, device=x.device)
    _lib.call('sconf_sub_conv0_fwd', _p(x), _dt(x), _p(w), _p(bias), _p(y), B, F, T, Cc, _stream())
    return y


def sub_dwconv_fwd(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """x (B,Ti,Fi,C) bf16 pre-activation -> dwconv3x3s2(SiLU(x)) + bias, (B,To,Fo,C) bf16."""
    _chk(x, 'x', torch.bfloat16)
    B, Ti, Fi, Cc = x.shape
    y = torch.empty(B, _half(Ti), _half(Fi), Cc, dtype=torch.bfloat16, device=x.device)
    _lib.call('sconf_sub_dwconv_fwd', _p(x), _p(w), _p(bias), _p(y), B, Ti, Fi, Cc, _stream())
    return y


def sub_dwconv_bwd(dout: torch.Tensor, w: torch.Tensor, pre_in: torch.Tensor, dw: torch.Tensor, dbias: torch.Tensor,
                   colsum_into: Optional[torch.Tensor] = None) -> torch.Tensor:
    """colsum_into (f32, C elements): += column sums of the returned dpre (the bias gradient of the 1x1 conv that produced pre_in)."""
    _chk(dout, 'dout', torch.bfloat16); _chk(pre_in, 'pre_in', torch.bfloat16)
    B, Ti, Fi, Cc = pre_in.shape
    dpre = torch.empty_like(pre_in)
    if colsum_into is not None:
        _chk(colsum_into, 'colsum_into', torch.float32)
        if colsum_into.numel() != Cc: raise ValueError('colsum_into must have one element per channel')
    _lib.call('sconf_sub_dwconv_bwd', _p(dout), _p(w), _p(pre_in), _p(dpre), _p(dw), _p(dbias), _p(colsum_into), B, Ti, Fi, Cc, _stream())
    return dpre


def sub_conv0_bwd_(dpre0: torch.Tensor, x: torch.Tensor, dw: torch.Tensor, dbias: torch.Tensor) -> None:
    _chk(dpre0, 'dpre0', torch.bfloat16); _chk(x, 'x')
    B, F, T = x.shape; Cc = dpre0.shape[-1]
    _lib.call('sconf_sub_conv0_bwd', _p(dpre0), _p(x), _dt(x), _p(dw), _p(dbias), B, F, T, Cc, _stream())


def sub_stage01_fwd(x: torch.Tensor, w0: torch.Tensor, b0: torch.Tensor, wd: torch.Tensor, bd: torch.Tensor) -> torch.Tensor:
    """Fused conv0 + SiLU + first depthwise conv: x (B,F,T) -> d1 (B,T4,F4,C) bf16 (no stage-0 tensor in HBM)."""
    _chk(x, 'x'); _chk(w0, 'w0', torch.float32); _chk(wd, 'wd', torch.float32)
    B, F, T = x.shape; Cc = w0.shape[0]
    d1 = torch.empty(B, _half(_half(T)), _half(_half(F)), Cc, dtype=torch.bfloat16, device=x.device)
    _lib.call('sconf_sub_stage01_fwd', _p(x), _dt(x), _p(w0), _p(b0), _p(wd), _p(bd), _p(d1), B, F, T, Cc, _stream())
    return d1


def sub_stage01_bwd_(dd1: torch.Tensor, x: torch.Tensor, w0, b0, wd, dw0, db0, dwd, dbd) -> None:
    """Parameter gradients of the fused stage (accumulated in place) from dd1 (B,T4,F4,C) bf16."""
    _chk(dd1, 'dd1', torch.bfloat16); _chk(x, 'x')
    B, F, T = x.shape; Cc = dd1.shape[-1]
    _lib.call('sconf_sub_stage01_bwd', _p(dd1), _p(x), _dt(x), _p(w0), _p(b0), _p(wd), _p(dw0), _p(db0), _p(dwd), _p(dbd),
              B, F, T, Cc, _stream())


def sub_silu_transpose(pre: torch.Tensor, ds: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fwd (ds None): pre (R,F8,C) -> (R, C*F8) = SiLU(pre) in the reference's c*F8+f order.
    bwd: ds (R, C*F8) -> (R,F8,C) = ds^T * SiLU'(pre)."""
    _chk(pre, 'pre', torch.bfloat16)
    R, F8, Cc = pre.shape
    if ds is None:
        out = torch.empty(R, Cc * F8, dtype=torch.bfloat16, device=pre.device)
        _lib.call('sconf_sub_silu_transpose', 0, _p(pre), None, _p(out), R, F8, Cc, _stream())
    else:
        _chk(ds, 'ds', torch.bfloat16)
        out = torch.empty_like(pre)
        _lib.call('sconf_sub_silu_transpose', 1, _p(pre), _p(ds), _p(out), R, F8, Cc, _stream())
    return out


# ------------------------------------------------------------------------------------------------
# CTC
# ------------------------------------------------------------------------------------------------
def ctc_fwd(log_probs: torch.Tensor, targets: torch.Tensor, input_lengths: torch.Tensor, target_lengths: torch.Tensor, blank: int):
    """log_probs (B,N,C) f32; targets (B,Smax) int32; lengths int32 (B,).  Returns nll (B,) f32 and the workspace."""
    _chk(log_probs, 'log_probs', torch.float32); _chk(targets, 'targets', torch.int32)
    _chk(input_lengths, 'input_lengths', torch.int32); _chk(target_lengths, 'target_lengths', torch.int32)
    B, N, Cn = log_probs.shape
    Smax = targets.shape[1]
    L = 2 * Smax + 1
    dev = log_probs.device
    lpg = torch.empty(B, N, L, dtype=torch.float32, device=dev)
    alpha = torch.empty(B, N, L, dtype=torch.float32, device=dev)
    beta = torch.empty(B, N, L, dtype=torch.float32, device=dev)
    nll = torch.empty(B, dtype=torch.float32, device=dev)
    offs = torch.empty(2 * B * N + B, dtype=torch.float64, device=dev)          # per-frame offsets of the renormalised rows + nll in f64
    _lib.call('sconf_ctc_fwd', _p(log_probs), _p(targets), _p(input_lengths), _p(target_lengths), _p(lpg), _p(alpha), _p(beta),
              _p(offs), _p(nll), B, N, Cn, Smax, int(blank), _stream())
    return nll, (lpg, alpha, beta, offs)


def ctc_fwd_logits(logits: torch.Tensor, targets: torch.Tensor, input_lengths: torch.Tensor, target_lengths: torch.Tensor, blank: int):
    """CTC loss straight from the decoder's logits (B,N,C) f32 (log_softmax folded into the emission gather).
    Returns nll (B,) f32 and the workspace (lse (B,N), lpg, alpha, beta, offs) the backward needs."""
    _chk(logits, 'logits', torch.float32); _chk(targets, 'targets', torch.int32)
    _chk(input_lengths, 'input_lengths', torch.int32); _chk(target_lengths, 'target_lengths', torch.int32)
    B, N, Cn = logits.shape
    Smax = targets.shape[1]
    L = 2 * Smax + 1
    dev = logits.device
    lse = torch.empty(B, N, dtype=torch.float32, device=dev)
    lpg = torch.empty(B, N, L, dtype=torch.float32, device=dev)
    alpha = torch.empty(B, N, L, dtype=torch.float32, device=dev)
    beta = torch.empty(B, N, L, dtype=torch.float32, device=dev)
    nll = torch.empty(B, dtype=torch.float32, device=dev)
    offs = torch.empty(2 * B * N + B, dtype=torch.float64, device=dev)
    _lib.call('sconf_ctc_fwd_logits', _p(logits), _p(targets), _p(input_lengths), _p(target_lengths), _p(lse), _p(lpg), _p(alpha),
              _p(beta), _p(offs), _p(nll), B, N, Cn, Smax, int(blank), _stream())
    return nll, (lse, lpg, alpha, beta, offs)


def ctc_bwd_logits(logits, ws, nll, targets, input_lengths, target_lengths, grad_out: Optional[torch.Tensor], blank: int,
                   colsum_into: Optional[torch.Tensor] = None) -> torch.Tensor:
    """d nll / d logits (B,N,C) in bf16 (CTC gradient through log_softmax), scaled by grad_out (B,) f32 if given;
    colsum_into (f32, C elements): += its column sums (the decoder bias gradient), same pass."""
    lse, lpg, alpha, beta, offs = ws
    B, N, Cn = logits.shape
    dl = torch.empty(B, N, Cn, dtype=torch.bfloat16, device=logits.device)
    if grad_out is not None: _chk(grad_out, 'grad_out', torch.float32)
    wsp = None
    if colsum_into is not None:
        _chk(colsum_into, 'colsum_into', torch.float32)
        if colsum_into.numel() != Cn: raise ValueError('colsum_into must have one element per class')
        wsp = torch.empty(int(_lib.load().sconf_ctc_bwd_logits_workspace(B * N, Cn)), dtype=torch.float32, device=logits.device)
    _lib.call('sconf_ctc_bwd_logits', _p(logits), _p(lse), _p(lpg), _p(alpha), _p(beta), _p(offs), _p(nll), _p(targets), _p(input_lengths),
              _p(target_lengths), _p(grad_out), _p(dl), _p(colsum_into), _p(wsp), B, N, Cn, targets.shape[1], int(blank), _stream())
    return dl


def ctc_bwd(log_probs, ws, nll, targets, input_lengths, target_lengths, grad_out: Optional[torch.Tensor], blank: int) -> torch.Tensor:
    lpg, alpha, beta, offs = ws
    B, N, Cn = log_probs.shape
    grad = torch.empty_like(log_probs)
    if grad_out is not None: _chk(grad_out, 'grad_out', torch.float32)
    _lib.call('sconf_ctc_bwd', _p(log_probs), _p(lpg), _p(alpha), _p(beta), _p(offs), _p(nll), _p(targets), _p(input_lengths),
              _p(target_lengths), _p(grad_out), _p(grad), B, N, Cn, targets.shape[1], int(blank), _stream())
    return grad


# ------------------------------------------------------------------------------------------------
# optimiser
# ------------------------------------------------------------------------------------------------
def sumsq_(g: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out (f64 scalar tensor) += sum(g^2)."""
    _chk(g, 'g', torch.float32); _chk(out, 'out', torch.float64)
    _lib.call('sconf_sumsq', _p(g), g.numel(), _p(out), _stream())
    return out


def madgrad_step_(p, g, grad_sum_sq, s, x0, shadow: Optional[torch.Tensor], sumsq: Optional[torch.Tensor], max_norm: float,
                  grad_scale: float, lr: float, momentum: float, eps: float, weight_decay: float, k) -> None:
    """k: the steps applied so far, a Python int or a device int64 tensor (read by the kernel, no host sync).  At k == 0 the
    kernel sets x0 := p."""
    for t, n in ((p, 'p'), (g, 'g'), (grad_sum_sq, 'grad_sum_sq'), (s, 's'), (x0, 'x0')): _chk(t, n, torch.float32)
    k_dev = k if torch.is_tensor(k) else None
    if k_dev is not None: _chk(k_dev, 'k', torch.int64)
    _lib.call('sconf_madgrad_step', _p(p), _p(g), _p(grad_sum_sq), _p(s), _p(x0), _p(shadow), p.numel(), _p(sumsq), float(max_norm),
              float(grad_scale), float(lr), float(momentum), float(eps), float(weight_decay), 0 if k_dev is not None else int(k),
              _p(k_dev), _stream())


def madgrad_advance_(k: torch.Tensor, sumsq: Optional[torch.Tensor], grad_scale: float = 1.0) -> None:
    """k (device int64 scalar) += 1 unless the step was skipped because the gradient norm was not finite."""
    _chk(k, 'k', torch.int64)
    _lib.call('sconf_madgrad_advance', _p(k), _p(sumsq), float(grad_scale), _stream())


# ------------------------------------------------------------------------------------------------
# forward-only inference helpers (sliding-window transcription)
# ------------------------------------------------------------------------------------------------
def overlap_add_exp_(logp: torch.Tensor, acc: torch.Tensor, count: torch.Tensor, pos0: int, stride: int) -> None:
    """acc[pos0 + w*stride + t] += exp(logp[w, t]), count[...] += 1 for W windows logp (W,n,C) f32 (in place, gather form)."""
    _chk(logp, 'logp', torch.float32); _chk(acc, 'acc', torch.float32); _chk(count, 'count', torch.float32)
    W, n, Cc = logp.shape
    _lib.call('sconf_overlap_add_exp', _p(logp), W, n, Cc, int(stride), int(pos0), _p(acc), _p(count), acc.shape[0], _stream())


def overlap_finalize(acc: torch.Tensor, count: torch.Tensor, n_rows: int) -> torch.Tensor:
    """log(acc / count) for the first n_rows rows."""
    _chk(acc, 'acc', torch.float32); _chk(count, 'count', torch.float32)
    out = torch.empty(n_rows, acc.shape[1], dtype=torch.float32, device=acc.device)
    _lib.call('sconf_overlap_finalize', _p(acc), _p(count), _p(out), n_rows, acc.shape[1], _stream())
    return out


def argmax_rows(x: torch.Tensor) -> torch.Tensor:
    """int32 argmax over the last dim of an f32 (M,C) tensor, first index on ties."""
    _chk(x, 'x', torch.float32)
    Cc = x.shape[-1]; M = x.numel() // Cc
    idx = torch.empty(M, dtype=torch.int32, device=x.device)
    _lib.call('sconf_argmax_rows', _p(x), M, Cc, _p(idx), _stream())
    return idx
