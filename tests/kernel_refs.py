"""Plain-PyTorch fp32 references for every op in ``lcasr_amd.hip.ops`` (same names, same signatures).

TEST INFRASTRUCTURE.  Used two ways:
  * GPU parity tests (`-m gpu`): run the HIP op and this reference on the same inputs and compare.
  * CPU host-logic tests: `tests/conftest.py::emulated_ops` monkeypatches `lcasr_amd.hip.ops` with these
    functions so the autograd wiring / module logic of the product package can be checked against the
    oracle and the golden fixtures without a GPU.  The product package itself never imports this file.

All math is done in fp32 on the given (bf16-rounded) inputs; outputs are rounded to the op's output dtype,
mirroring the kernels' "bf16 storage, f32 accumulate" contract.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F

f32 = torch.float32


def require_gpu(t, what='input'):
    return None


def _gelu(x): return F.gelu(x, approximate='tanh')


def _dgelu(x):
    # closed form (no inner autograd: this runs inside Function.forward, possibly under torch.utils.checkpoint, whose saved-tensor
    # hooks would see an inner backward as a reason to recompute the region)
    k = math.sqrt(2.0 / math.pi)
    t = torch.tanh(k * (x + 0.044715 * x ** 3))
    return 0.5 * (1 + t) + 0.5 * x * (1 - t * t) * k * (1 + 3 * 0.044715 * x * x)


def _dsilu(x):
    s = torch.sigmoid(x)
    return s * (1 + x * (1 - s))


def gemm(a, b, layout='nt', bias=None, resid=None, aux=None, act='none', alpha=1.0, out_dtype=torch.bfloat16,
         save_pre=False, split_k=1, accum=None):
    if accum is not None:
        accum += gemm(a, b, layout, alpha=alpha, out_dtype=f32).reshape(accum.shape)
        return accum
    A, Bm = a.to(f32), b.to(f32)
    if layout == 'nt': acc = A @ Bm.t()
    elif layout == 'nn': acc = A @ Bm
    else: acc = A.t() @ Bm
    if bias is not None: acc = acc + bias
    pre = acc.to(torch.bfloat16) if save_pre else None
    if act == 'gelu_dsave':
        pre = _dgelu(acc).to(torch.bfloat16); acc = _gelu(acc)
    elif act == 'mulaux': acc = acc * aux.to(f32)
    elif act == 'gelu': acc = _gelu(acc)
    elif act == 'silu': acc = F.silu(acc)
    elif act == 'dgelu': acc = acc * _dgelu(aux.to(f32))
    elif act == 'dsilu': acc = acc * _dsilu(aux.to(f32))
    acc = acc * alpha
    if resid is not None: acc = acc + resid
    c = acc.to(out_dtype)
    return (c, pre) if save_pre else c


def pick_split_k(M, N, K, n_cus=256):
    return 1


def norm_fwd(x, weight, bias, mode, eps, out_dtype):
    xf = x.to(f32)
    d = xf.shape[-1]
    if mode == 'layer_norm':
        mean = xf.mean(-1); var = xf.var(-1, unbiased=False)
        rstd = torch.rsqrt(var + eps)
        y = (xf - mean[..., None]) * rstd[..., None] * weight + (bias if bias is not None else 0)
    elif mode == 'rms_norm':
        rms = xf.norm(2, dim=-1) * d ** -0.5
        rstd = 1.0 / (rms + eps); mean = torch.zeros_like(rstd)
        y = xf * rstd[..., None] * weight
    else:
        rstd = torch.rsqrt((xf * xf).mean(-1) + eps); mean = torch.zeros_like(rstd)
        y = xf * rstd[..., None] * weight
    return y.to(out_dtype), mean.reshape(-1), rstd.reshape(-1)


def norm_bwd(dy, x, weight, mean, rstd, mode, eps, dres, dx_dtype, dweight, dbias, twin=False):
    xf = x.to(f32).detach().clone().requires_grad_(True)
    w = weight.detach().clone().requires_grad_(True)
    b = torch.zeros_like(weight).requires_grad_(True)
    with torch.enable_grad():
        d = xf.shape[-1]
        if mode == 'layer_norm': y = F.layer_norm(xf, (d,), w, b, eps)
        elif mode == 'rms_norm': y = w * (xf / (xf.norm(2, dim=-1, keepdim=True) * d ** -0.5 + eps))
        else: y = w * xf * torch.rsqrt((xf * xf).mean(-1, keepdim=True) + eps)
        y.backward(dy.to(f32))
    dx = xf.grad
    if dres is not None: dx = dx + dres
    dweight += w.grad
    if dbias is not None and mode == 'layer_norm': dbias += b.grad
    if twin and dx_dtype == f32:
        dx16 = dx.to(torch.bfloat16)
        return dx.to(dx_dtype), dx16, dx16.to(f32).reshape(-1, dx.shape[-1]).sum(0)
    return dx.to(dx_dtype)


def norm2_fwd(x, w1, b1, w2, b2, eps1, eps2, twice=False):
    y1, m1, r1 = norm_fwd(x, w1, b1, 'layer_norm', eps1, f32)
    if not twice:
        h2, m2, r2 = norm_fwd(y1, w2, b2, 'layer_norm', eps2, torch.bfloat16)
        return y1, h2, (m1, r1, m2, r2)
    y2, m2, r2 = norm_fwd(y1, w2, b2, 'layer_norm', eps2, f32)
    h3, m3, r3 = norm_fwd(y2, w2, b2, 'layer_norm', eps2, torch.bfloat16)
    return y1, h3, (m1, r1, m2, r2, m3, r3)


def norm2_bwd(dh2, x, w1, b1, w2, b2, stats, dres, dw1, db1, dw2, db2, twin=False):
    m1, r1, m2, r2 = stats[:4]
    y1 = ((x.to(f32) - m1.view(-1, 1)) * r1.view(-1, 1)).view(x.shape) * w1 + b1
    g = dh2
    if len(stats) == 6:
        m3, r3 = stats[4:]
        y2 = ((y1 - m2.view(-1, 1)) * r2.view(-1, 1)).view(x.shape) * w2 + b2
        g = norm_bwd(dh2, y2, w2, m3, r3, 'layer_norm', 1e-5, None, f32, dw2, db2)
    dy1 = norm_bwd(g, y1, w2, m2, r2, 'layer_norm', 1e-5, dres, f32, dw2, db2)
    return norm_bwd(dy1, x, w1, m1, r1, 'layer_norm', 1e-5, None, f32, dw1, db1, twin=twin)


def cast(x, dtype): return x.to(dtype)


def cast_transpose(w): return w.t().contiguous().to(torch.bfloat16)


def cast_shadows(table, n_entries, total_tiles):
    """Reference of sconf_cast_shadows: the table holds RAW addresses (here: of CPU tensors), exactly like the C ABI."""
    import ctypes
    for src, dst, dst_t, R, C, _ in table[:n_entries].tolist():
        regroup = R < 0                                       # "(h d qkv)" rows -> [q | k | v] (sconf.h)
        R = abs(R)
        w = torch.frombuffer((ctypes.c_float * (R * C)).from_address(src), dtype=torch.float32).view(R, C)
        if regroup: w = w.view(R // 3, 3, C).permute(1, 0, 2).reshape(R, C)
        if dst:
            torch.frombuffer((ctypes.c_uint16 * (R * C)).from_address(dst), dtype=torch.bfloat16).view(R, C).copy_(w)
        if dst_t:
            torch.frombuffer((ctypes.c_uint16 * (R * C)).from_address(dst_t), dtype=torch.bfloat16).view(C, R).copy_(w.t())


def _rot(x, cos, sin, sign=1.0):
    # x (B,N,H,D) f32 ; cos/sin (N, D/2)
    h = x.shape[-1] // 2
    c = cos[None, :, None, :]; s = sin[None, :, None, :] * sign
    x1, x2 = x[..., :h], x[..., h:]
    return torch.cat([x1 * c - x2 * s, x2 * c + x1 * s], dim=-1)


def rotary_qkv_fwd(qkv, cos, sin, B, N, H, D):
    t = qkv.to(f32).reshape(B, N, H, D, 3)
    q, k, v = t[..., 0], t[..., 1], t[..., 2]
    if cos is not None: q, k = _rot(q, cos, sin), _rot(k, cos, sin)
    bf = torch.bfloat16
    return q.contiguous().to(bf), k.contiguous().to(bf), v.contiguous().to(bf)


def rotary_qkv_bwd(dq, dk, dv, cos, sin, B, N, H, D):
    dq, dk, dv = dq.to(f32), dk.to(f32), dv.to(f32)
    if cos is not None: dq, dk = _rot(dq, cos, sin, -1.0), _rot(dk, cos, sin, -1.0)
    return torch.stack([dq, dk, dv], dim=-1).reshape(B * N, H * D * 3).to(torch.bfloat16)


def gemm_qkv_rotary(x, w_regrouped, bias, cos, sin, N, H, D):
    """f32 product, rotation on the f32 result, ONE rounding to bf16 (the fused epilogue's rounding points)."""
    y = x.to(f32) @ w_regrouped.to(f32).t()
    if bias is not None: y = y + bias
    M = y.shape[0]
    y5 = y.view(M // N, N, 3, H, D).clone()
    c, s_ = cos.view(1, N, 1, D // 2), sin.view(1, N, 1, D // 2)
    for w in (0, 1):
        a, b = y5[:, :, w, :, :D // 2].clone(), y5[:, :, w, :, D // 2:].clone()
        y5[:, :, w, :, :D // 2] = a * c - b * s_
        y5[:, :, w, :, D // 2:] = b * c + a * s_
    return y5.view(M, 3 * H * D).to(torch.bfloat16)


def rotary_inplace_(qkv, cos, sin, B, N, H, D):
    t = qkv.view(B, N, 3, H, D)
    t[:, :, 0] = _rot(t[:, :, 0].to(f32), cos, sin).to(qkv.dtype)
    t[:, :, 1] = _rot(t[:, :, 1].to(f32), cos, sin).to(qkv.dtype)
    return qkv


def softmax_fwd(x, log, out_dtype):
    xf = x.to(f32)
    return (F.log_softmax(xf, -1) if log else F.softmax(xf, -1)).to(out_dtype)


def softmax_bwd(y, dy, log, out_dtype, colsum_into=None):
    yf, g = y.to(f32), dy.to(f32)
    if log: dx = g - yf.exp() * g.sum(-1, keepdim=True)
    else: dx = yf * (g - (g * yf).sum(-1, keepdim=True))
    dx = dx.to(out_dtype)
    if colsum_into is not None: colsum_into += dx.to(f32).reshape(-1, dx.shape[-1]).sum(0).view(colsum_into.shape)
    return dx


def rowdot(a, b, bias=None):
    bf = b.to(f32) if bias is None else b.to(f32) - bias
    return (a.to(f32) * bf).sum(-1)


def gemm_softmax_bwd_eligible(M, V, K):
    """Shape rule of the 256-row NT kernels (whole 256 x 256 tiles, whole K-tiles, at least 3/4 of a round of 256 workgroups)."""
    return M % 256 == 0 and V % 256 == 0 and K % 64 == 0 and (M // 256) * (V // 256) * 4 >= 3 * 256


def gemm_softmax_bwd(dy, wt, probs, delta, colsum_into=None):
    dp = dy.to(f32) @ wt.to(f32).t()
    dl = (probs.to(f32) * (dp - delta[:, None])).to(torch.bfloat16)
    if colsum_into is not None: colsum_into += dl.to(f32).sum(0)
    return dl


def colsum_(x, out, alpha=1.0):
    out += alpha * x.to(f32).reshape(-1, x.shape[-1]).sum(0)
    return out


def mask_rows_(x, lengths, B, N):
    v = x.view(B, N, -1)
    pad = torch.arange(N, device=x.device)[None, :] >= lengths[:, None]
    v.masked_fill_(pad[..., None], 0)
    return x


def _attn_mask(B, N, lengths, window, device):
    ok = torch.ones(B, N, N, dtype=torch.bool, device=device)           # [b, q, key]
    if lengths is not None:
        ok &= (torch.arange(N, device=device)[None, None, :] < lengths[:, None, None])
    i = torch.arange(N, device=device)[:, None]; j = torch.arange(N, device=device)[None, :]
    if window[0] >= 0: ok &= (j >= i - window[0])[None]
    if window[1] >= 0: ok &= (j <= i + window[1])[None]
    return ok


def attn_fwd(q, k, v, lengths, window=(-1, -1), scale=None):
    B, N, H, D = q.shape
    sc = scale if scale is not None else D ** -0.5
    s = torch.einsum('bihd,bjhd->bhij', q.to(f32), k.to(f32)) * sc
    ok = _attn_mask(B, N, lengths, window, q.device)
    s = s.masked_fill(~ok[:, None], float('-inf'))
    lse = torch.logsumexp(s, dim=-1)
    p = torch.exp(s - lse[..., None])
    p = torch.nan_to_num(p, nan=0.0)
    o = torch.einsum('bhij,bjhd->bihd', p, v.to(f32))
    if lengths is not None:
        qpad = torch.arange(N, device=q.device)[None, :] >= lengths[:, None]
        o = o.masked_fill(qpad[:, :, None, None], 0.0)
        lse = lse.masked_fill(qpad[:, None, :], float('inf'))
    return o.contiguous().to(torch.bfloat16), lse.contiguous()


def attn_bwd(q, k, v, o, dout, lse, lengths, window=(-1, -1), scale=None, rot=None, out=None):
    B, N, H, D = q.shape
    sc = scale if scale is not None else D ** -0.5
    qf, kf, vf = (t.to(f32).detach().clone().requires_grad_(True) for t in (q, k, v))
    with torch.enable_grad():
        s = torch.einsum('bihd,bjhd->bhij', qf, kf) * sc
        ok = _attn_mask(B, N, lengths, window, q.device)
        s = s.masked_fill(~ok[:, None], float('-inf'))
        dead = ~ok.any(-1)                                  # query rows with no visible key (padding + window)
        s = s.masked_fill(dead[:, None, :, None], 0.0)
        p = torch.softmax(s, dim=-1) * (~dead)[:, None, :, None]
        oo = torch.einsum('bhij,bjhd->bihd', p, vf)
        g = dout.to(f32)
        if lengths is not None:
            qpad = torch.arange(N, device=q.device)[None, :] >= lengths[:, None]
            oo = oo.masked_fill(qpad[:, :, None, None], 0.0)
        oo.backward(g)
    bf = torch.bfloat16
    dq, dk, dv = qf.grad, kf.grad, vf.grad
    if rot is not None:                                     # gradients of the unrotated q, k: transpose of the rotation
        dq, dk = _rot(dq, rot[0], rot[1], -1.0), _rot(dk, rot[0], rot[1], -1.0)
    if out is not None:
        out[0].copy_(dq.to(bf)); out[1].copy_(dk.to(bf)); out[2].copy_(dv.to(bf))
        return out
    return dq.contiguous().to(bf), dk.contiguous().to(bf), dv.contiguous().to(bf)


def _glu_masked(g, lengths, B, N):
    d = g.shape[-1] // 2
    gf = g.to(f32).reshape(B, N, 2 * d)
    a = gf[..., :d] * torch.sigmoid(gf[..., d:])
    if lengths is not None:
        pad = torch.arange(N, device=g.device)[None, :] >= lengths[:, None]
        a = a.masked_fill(pad[..., None], 0.0)
    return a


def glu_dwconv_fwd(g, lengths, w, bias, B, N):
    d = g.shape[-1] // 2
    a = _glu_masked(g, lengths, B, N)
    ks = w.numel() // d
    h = F.conv1d(a.transpose(1, 2), w.reshape(d, 1, ks), bias, padding=(ks - 1) // 2, groups=d).transpose(1, 2).reshape(B * N, d)
    hb = h.to(torch.bfloat16)                       # statistics of the ROUNDED values (as the kernel does)
    stats = torch.stack([hb.double().sum(0), (hb.double() ** 2).sum(0)])
    return hb, stats


def _rmax_dmax(nbt):
    nb = float(nbt)
    return min(max(2 / 35000 * nb + 25 / 35, 1.0), 3.0), min(max(5 / 20000 * nb - 25 / 20, 0.0), 5.0)


def brn_finalize(stats, count, running_mean, running_std, num_batches_tracked, weight, bias, training, eps=1e-3, momentum=0.01):
    d = weight.numel()
    coef = torch.empty(6, d, dtype=f32, device=weight.device)
    if training:
        rmax, dmax = _rmax_dmax(num_batches_tracked)
        mean = (stats[0] / count)
        var = (stats[1] / count - mean * mean).clamp_min(0)
        s = var.sqrt().to(f32) + eps
        mean = mean.to(f32)
        r = (s / running_std).clamp(1 / rmax, rmax)
        dd = ((mean - running_mean) / running_std).clamp(-dmax, dmax)
        coef[0], coef[1], coef[2], coef[3] = mean, s, r, dd
        coef[4] = weight * r / s
        coef[5] = weight * (dd - mean * r / s) + bias
        running_mean += momentum * (mean - running_mean)
        running_std += momentum * (s - running_std)
        num_batches_tracked += 1
    else:
        coef[0], coef[1], coef[2], coef[3] = running_mean, running_std, 1.0, 0.0
        coef[4] = weight / running_std
        coef[5] = bias - weight * running_mean / running_std
    return coef


def affine_silu_fwd(h, coef):
    return F.silu(h.to(f32) * coef[4] + coef[5]).to(torch.bfloat16)


def convmod_bwd(dy, h, g, lengths, w, brn_weight, coef, B, N, training, eps, dw, dbias, dbrn_weight, dbrn_bias, colsum=False):
    d = h.shape[-1]; ks = w.numel() // d
    hf = h.to(f32); gy = dy.to(f32)
    mean, s, r, dd, A, Bc = coef
    dz = gy * _dsilu(hf * A + Bc)
    xh0 = (hf - mean) / s
    S1, S2 = dz.sum(0), (dz * xh0).sum(0)
    dbrn_weight += r * S2 + dd * S1
    dbrn_bias += S1
    M = hf.shape[0]
    if training:
        sigma = s - eps
        k2 = torch.where(sigma > 0, A * (s / sigma) * (S2 / M), torch.zeros_like(s))
        dh = A * dz - A * (S1 / M) - xh0 * k2
    else:
        dh = A * dz
    # dwconv + GLU backward via autograd on the f32 forward
    gf = g.to(f32).detach().clone().requires_grad_(True)
    wf = w.detach().clone().reshape(d, 1, ks).requires_grad_(True)
    bf_ = torch.zeros(d, device=w.device).requires_grad_(True)
    with torch.enable_grad():
        g3 = gf.reshape(B, N, 2 * d)
        a = g3[..., :d] * torch.sigmoid(g3[..., d:])
        if lengths is not None:
            pad = torch.arange(N, device=g.device)[None, :] >= lengths[:, None]
            a = a.masked_fill(pad[..., None], 0.0)
        hh = F.conv1d(a.transpose(1, 2), wf, bf_, padding=(ks - 1) // 2, groups=d).transpose(1, 2).reshape(B * N, d)
        hh.backward(dh)
    dw += wf.grad.reshape(dw.shape)
    dbias += bf_.grad
    return (gf.grad.to(torch.bfloat16), gf.grad.sum(0)) if colsum else gf.grad.to(torch.bfloat16)


def _half(n): return (n - 1) // 2 + 1


def sub_conv0_fwd(x, w, bias):
    xf = x.to(f32).transpose(1, 2).unsqueeze(1)                                  # (B,1,T,F)
    y = F.conv2d(xf, w.reshape(-1, 1, 3, 3), bias, stride=2, padding=1)          # (B,C,T2,F2)
    return y.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)


def _q(t):
    """bf16 operand rounding of the MFMA / v_dot2c kernels (the gradient passes straight through the casts)."""
    return t.to(torch.bfloat16).to(f32)


def sub_dwconv_fwd(x, w, bias):
    # rounding points of dwconv_window_fwd_kernel: SiLU values stored as bf16 pairs, taps as bf16 (v_dot2c_f32_bf16), f32 sums
    Cc = x.shape[-1]
    xf = _q(F.silu(x.to(f32))).permute(0, 3, 1, 2)
    y = F.conv2d(xf, _q(w).reshape(Cc, 1, 3, 3), bias, stride=2, padding=1, groups=Cc)
    return y.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)


def sub_dwconv_bwd(dout, w, pre_in, dw, dbias, colsum_into=None):
    Cc = pre_in.shape[-1]
    p = pre_in.to(f32).detach().clone().requires_grad_(True)
    wf = w.detach().clone().reshape(Cc, 1, 3, 3).requires_grad_(True)
    bf_ = torch.zeros(Cc, device=w.device).requires_grad_(True)
    with torch.enable_grad():
        y = F.conv2d(F.silu(p).permute(0, 3, 1, 2), wf, bf_, stride=2, padding=1, groups=Cc).permute(0, 2, 3, 1)
        y.backward(dout.to(f32))
    dw += wf.grad.reshape(dw.shape); dbias += bf_.grad
    out = p.grad.to(torch.bfloat16)
    if colsum_into is not None: colsum_into += out.to(f32).reshape(-1, Cc).sum(0).view(colsum_into.shape)
    return out


def sub_conv0_bwd_(dpre0, x, dw, dbias):
    Cc = dpre0.shape[-1]
    wf = torch.zeros(Cc, 1, 3, 3, device=dw.device).requires_grad_(True)
    bf_ = torch.zeros(Cc, device=dw.device).requires_grad_(True)
    with torch.enable_grad():
        y = F.conv2d(x.to(f32).transpose(1, 2).unsqueeze(1), wf, bf_, stride=2, padding=1).permute(0, 2, 3, 1)
        y.backward(dpre0.to(f32))
    dw += wf.grad.reshape(dw.shape); dbias += bf_.grad


def _stage01(x, w0, b0, wd, bd):
    # rounding points of the MFMA kernels (subsample_mfma.hip): mel and conv0 taps as bf16 MFMA operands (bias as hi + lo: exact
    # enough to be f32 here), stage-0 activations stored as bf16, depthwise taps as bf16, f32 accumulation everywhere
    Cc = w0.shape[0]
    y0 = F.conv2d(_q(x.to(f32)).transpose(1, 2).unsqueeze(1), _q(w0).reshape(Cc, 1, 3, 3), b0, stride=2, padding=1)
    return F.conv2d(_q(F.silu(y0)), _q(wd).reshape(Cc, 1, 3, 3), bd, stride=2, padding=1, groups=Cc).permute(0, 2, 3, 1)


def sub_stage01_fwd(x, w0, b0, wd, bd):
    return _stage01(x, w0, b0, wd, bd).contiguous().to(torch.bfloat16)


def sub_stage01_bwd_(dd1, x, w0, b0, wd, dw0, db0, dwd, dbd):
    ps = [t.detach().clone().requires_grad_(True) for t in (w0, b0, wd)]
    bdz = torch.zeros_like(b0).requires_grad_(True)
    with torch.enable_grad():
        _stage01(x, ps[0], ps[1], ps[2], bdz).backward(dd1.to(f32))
    dw0 += ps[0].grad.reshape(dw0.shape); db0 += ps[1].grad; dwd += ps[2].grad.reshape(dwd.shape); dbd += bdz.grad


def sub_silu_transpose(pre, ds=None):
    R, F8, Cc = pre.shape
    pf = pre.to(f32)
    if ds is None:
        return F.silu(pf).transpose(1, 2).reshape(R, Cc * F8).contiguous().to(torch.bfloat16)
    g = ds.to(f32).reshape(R, Cc, F8).transpose(1, 2)
    return (g * _dsilu(pf)).contiguous().to(torch.bfloat16)


def ctc_fwd(log_probs, targets, input_lengths, target_lengths, blank):
    nll = F.ctc_loss(log_probs.transpose(0, 1), targets.long(), input_lengths.long(), target_lengths.long(), blank=blank,
                     reduction='none', zero_infinity=False)
    return nll.to(f32), (None, None, None)


def ctc_bwd(log_probs, ws, nll, targets, input_lengths, target_lengths, grad_out, blank):
    lp = log_probs.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        nll_ = F.ctc_loss(lp.transpose(0, 1), targets.long(), input_lengths.long(), target_lengths.long(), blank=blank,
                          reduction='none', zero_infinity=False)
        nll_.backward(grad_out if grad_out is not None else torch.ones_like(nll_))
    return lp.grad


def ctc_fwd_logits(logits, targets, input_lengths, target_lengths, blank):
    nll, _ = ctc_fwd(torch.log_softmax(logits.to(f32), -1), targets, input_lengths, target_lengths, blank)
    return nll, (None, None, None, None)


def ctc_bwd_logits(logits, ws, nll, targets, input_lengths, target_lengths, grad_out, blank, colsum_into=None):
    lg = logits.detach().to(f32).clone().requires_grad_(True)
    with torch.enable_grad():
        nll_ = F.ctc_loss(torch.log_softmax(lg, -1).transpose(0, 1), targets.long(), input_lengths.long(), target_lengths.long(),
                          blank=blank, reduction='none', zero_infinity=False)
        nll_.backward(grad_out if grad_out is not None else torch.ones_like(nll_))
    dl = lg.grad.to(torch.bfloat16)
    if colsum_into is not None: colsum_into += dl.to(f32).reshape(-1, dl.shape[-1]).sum(0)
    return dl


def sumsq_(g, out):
    out += (g.double() ** 2).sum()
    return out


def madgrad_step_(p, g, grad_sum_sq, s, x0, shadow, sumsq, max_norm, grad_scale, lr, momentum, eps, weight_decay, k):
    coef = grad_scale
    if sumsq is not None:
        tot = math.sqrt(float(sumsq)) * abs(grad_scale)
        if not math.isfinite(tot): return
        if max_norm > 0: coef *= min(1.0, max_norm / (tot + 1e-6))
    if lr != 0: lr = lr + eps
    ck = 1 - momentum
    k = int(k)
    lamb = lr * math.sqrt(k + 1)
    if k == 0: x0.copy_(p)
    gv = g * coef
    if weight_decay != 0: gv = gv + weight_decay * p
    grad_sum_sq.addcmul_(gv, gv, value=lamb)
    rms = grad_sum_sq.pow(1 / 3).add_(eps)
    s.add_(gv, alpha=lamb)
    z = x0.addcdiv(s, rms, value=-1)
    p.mul_(1 - ck).add_(z, alpha=ck)
    if shadow is not None: shadow.copy_(p)


def madgrad_advance_(k, sumsq, grad_scale=1.0):
    if sumsq is not None and not math.isfinite(math.sqrt(float(sumsq)) * abs(grad_scale)): return
    k.add_(1)


def overlap_add_exp_(logp, acc, count, pos0, stride):
    W, n, _ = logp.shape
    for w in range(W):                                         # the reference's order: window after window
        p = pos0 + w * stride
        acc[p:p + n] += torch.exp(logp[w]); count[p:p + n] += 1


def overlap_finalize(acc, count, n_rows):
    return torch.log(acc[:n_rows] / count[:n_rows, None])


def argmax_rows(x):
    return torch.argmax(x.reshape(-1, x.shape[-1]), dim=-1).to(torch.int32)
