"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

numpy float64 restatement of the CTC loss the reference trains with:
``torch.nn.CTCLoss(blank=num_classes-1, reduction='sum')`` applied to
``final_posteriors.transpose(0,1)`` (`/root/reference/exp/train.py:104,249`).
The algorithm lives in the third-party dependency ``torch`` (unpinned in the
reference's requirements.txt; this image has 2.10.0+rocm7.0): the published
alpha/beta recursion of Graves et al. 2006 as implemented by ATen's
``ctc_loss_cpu`` — log-space, lattice l' = [blank, l1, blank, l2, ... blank],
transition s-2 → s allowed iff l'_s != blank and l'_s != l'_{s-2}.
Gradient convention follows ATen's native backward: for t < input_length
    grad[t, c] = g * (exp(lp[t,c]) - exp(logsum_{s: l'_s = c}(alpha_t[s]+beta_t[s]) + nll - lp[t,c]))
and 0 for t >= input_length (this equals d loss / d logits when lp =
log_softmax(logits)).

Pinned by tests/test_oracle_golden.py against ``torch.nn.functional.ctc_loss``
run in-process (the reference's own call) and against the golden fixtures.
Pure-python loops: small cases only.
"""
from __future__ import annotations

import numpy as np

NEG_INF = -np.inf


def _lse(*xs):
    m = max(xs)
    if m == NEG_INF:
        return NEG_INF
    return m + np.log(sum(np.exp(x - m) for x in xs))


def ctc_alpha_beta(lp: np.ndarray, target: np.ndarray, T: int, S: int, blank: int):
    """lp: (N, C) log-probs of one sample.  Returns (nll, alpha (T,L), beta (T,L))."""
    L = 2 * S + 1
    lab = np.full(L, blank, dtype=np.int64)
    lab[1::2] = target[:S]
    alpha = np.full((T, L), NEG_INF)
    beta = np.full((T, L), NEG_INF)
    alpha[0, 0] = lp[0, blank]
    if S > 0:
        alpha[0, 1] = lp[0, lab[1]]
    for t in range(1, T):
        for s in range(L):
            a = alpha[t - 1, s]
            b = alpha[t - 1, s - 1] if s >= 1 else NEG_INF
            c = alpha[t - 1, s - 2] if (s >= 2 and lab[s] != blank and lab[s] != lab[s - 2]) else NEG_INF
            alpha[t, s] = _lse(a, b, c) + lp[t, lab[s]]
    ll = _lse(alpha[T - 1, L - 1], alpha[T - 1, L - 2] if L > 1 else NEG_INF)
    beta[T - 1, L - 1] = lp[T - 1, blank]
    if L > 1:
        beta[T - 1, L - 2] = lp[T - 1, lab[L - 2]]
    for t in range(T - 2, -1, -1):
        for s in range(L):
            a = beta[t + 1, s]
            b = beta[t + 1, s + 1] if s + 1 < L else NEG_INF
            c = beta[t + 1, s + 2] if (s + 2 < L and lab[s] != blank and lab[s] != lab[s + 2]) else NEG_INF
            beta[t, s] = _lse(a, b, c) + lp[t, lab[s]]
    return -ll, alpha, beta


def ctc_loss_and_grad(lp_bnc: np.ndarray, targets: np.ndarray, in_len: np.ndarray, tgt_len: np.ndarray,
                      blank: int, grad_out: float = 1.0):
    """Sum-reduced loss and ATen-convention gradient w.r.t. lp (B,N,C)."""
    B, N, C = lp_bnc.shape
    lp = lp_bnc.astype(np.float64)
    grad = np.zeros_like(lp)
    nlls = np.zeros(B)
    for b in range(B):
        T, S = int(in_len[b]), int(tgt_len[b])
        nll, alpha, beta = ctc_alpha_beta(lp[b], targets[b], T, S, blank)
        nlls[b] = nll
        L = 2 * S + 1
        lab = np.full(L, blank, dtype=np.int64)
        lab[1::2] = targets[b][:S]
        for t in range(T):
            lcab = np.full(C, NEG_INF)
            for s in range(L):
                lcab[lab[s]] = _lse(lcab[lab[s]], alpha[t, s] + beta[t, s])
            grad[b, t] = grad_out * (np.exp(lp[b, t]) - np.exp(lcab + nll - lp[b, t]))
    return nlls.sum(), nlls, grad


# ---- the same recursion vectorised over the lattice states (float64 numpy): long contexts ---------------------------------
# ctc_alpha_beta above is a pure-Python double loop (T * L iterations): fine for the fixtures, hopeless at the 131072-frame
# context (T = 16384 frames, L = 8193 states).  These functions restate the identical recursion with one numpy expression per
# frame; tests/test_oracle_golden.py pins them against the loop version (and through it against torch's op).
def _lse3_vec(a, b, c):
    m = np.maximum(np.maximum(a, b), c)
    safe = np.where(np.isfinite(m), m, 0.0)
    with np.errstate(divide='ignore'):
        r = safe + np.log(np.exp(a - safe) + np.exp(b - safe) + np.exp(c - safe))
    return np.where(np.isfinite(m), r, NEG_INF)


def ctc_alpha_beta_vec(lp: np.ndarray, target: np.ndarray, T: int, S: int, blank: int):
    """lp (N, C) float64 log-probs of one sample -> (nll, alpha (T, L), beta (T, L), lab (L,)), float64 throughout."""
    L = 2 * S + 1
    lab = np.full(L, blank, dtype=np.int64)
    lab[1::2] = target[:S]
    skip = np.zeros(L, dtype=bool)                       # s-2 -> s allowed (alpha)
    skip[2:] = (lab[2:] != blank) & (lab[2:] != lab[:-2])
    skip_b = np.zeros(L, dtype=bool)                     # s+2 -> s allowed (beta)
    skip_b[:-2] = (lab[:-2] != blank) & (lab[:-2] != lab[2:])
    em = lp[:T][:, lab]                                  # (T, L) emissions
    alpha = np.full((T, L), NEG_INF)
    beta = np.full((T, L), NEG_INF)
    alpha[0, 0] = em[0, 0]
    if L > 1:
        alpha[0, 1] = em[0, 1]
    pad = np.full(2, NEG_INF)
    for t in range(1, T):
        p = np.concatenate([pad, alpha[t - 1]])
        alpha[t] = _lse3_vec(p[2:], p[1:-1], np.where(skip, p[:-2], NEG_INF)) + em[t]
    ll = _lse(alpha[T - 1, L - 1], alpha[T - 1, L - 2] if L > 1 else NEG_INF)
    beta[T - 1, L - 1] = em[T - 1, L - 1]
    if L > 1:
        beta[T - 1, L - 2] = em[T - 1, L - 2]
    for t in range(T - 2, -1, -1):
        p = np.concatenate([beta[t + 1], pad])
        beta[t] = _lse3_vec(p[:-2], p[1:-1], np.where(skip_b, p[2:], NEG_INF)) + em[t]
    return -ll, alpha, beta, lab


def ctc_loss_and_grad_vec(lp_nc: np.ndarray, target: np.ndarray, T: int, S: int, blank: int, grad_out: float = 1.0, out_dtype=np.float64):
    """One sample: (nll, grad (N, C)) with the ATen convention of ctc_loss_and_grad; grad is returned in out_dtype (the sums are
    float64) so that a (16384, 4096) gradient need not be held twice in float64."""
    lp = np.asarray(lp_nc, dtype=np.float64)
    N, C = lp.shape
    nll, alpha, beta, lab = ctc_alpha_beta_vec(lp, target, T, S, blank)
    grad = np.zeros((N, C), dtype=out_dtype)
    for t in range(T):
        with np.errstate(over='ignore'):
            occ = np.bincount(lab, weights=np.exp(alpha[t] + beta[t] + nll - lp[t, lab]), minlength=C)
        grad[t] = grad_out * (np.exp(lp[t]) - occ)
    return nll, grad
