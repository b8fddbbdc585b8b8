"""Shared helpers for the model-level tests (CPU emulation and GPU parity)."""
import numpy as np
import torch

from conftest import golden_cfg, golden_state_dict, load_golden

TINY_CASES = ['tiny_ln_ragged', 'tiny_ln_equal', 'tiny_rms_ragged', 'tiny_ln_odd']


def build_from_fixture(fx, device='cpu'):
    from lcasr_amd.models.sconformer_xl import SCConformerXL
    m = SCConformerXL(**golden_cfg(fx))
    m.load_state_dict(golden_state_dict(fx))
    return m.to(device).train()


def run_step(m, fx, device='cpu', fused_loss=False):
    """forward + CTC(sum) + the reference loss scaling (exp/train.py:275) + backward.  Returns a dict of CPU results.
    fused_loss: the head and the loss as one operator (model(..., ctc_targets=...)); no log-probabilities then ('logp' is None)."""
    from lcasr_amd.losses import CTCLoss
    x = torch.from_numpy(fx['x']).to(device)
    ln = torch.from_numpy(fx['lengths']).to(device)
    B, _, T = x.shape
    tg, tl = torch.from_numpy(fx['targets']).to(device), torch.from_numpy(fx['target_lengths']).to(device)
    if fused_loss:
        out = m(x, length=ln, ctc_targets=(tg, tl))
        lp, loss = None, out['ctc_nll'].sum()
    else:
        out = m(x, length=ln)
        lp = out['final_posteriors']
        loss = CTCLoss(blank=m.decoder.num_classes - 1, reduction='sum')(lp.transpose(0, 1), tg, out['length'], tl)
    (loss / (T * B) * 100).backward()
    if device != 'cpu':
        torch.cuda.synchronize()
    return dict(logp=None if lp is None else lp.detach().float().cpu(), length=out['length'].cpu(), loss=float(loss),
                grads={k: p.grad.detach().float().cpu() for k, p in m.named_parameters()},
                buffers={k: v.detach().float().cpu() for k, v in m.state_dict().items() if 'batch_norm.running' in k or 'num_batches' in k})


def grad_errors(grads, ref_grads):
    """max |g - ref| per tensor, relative to max(|ref|_max, 1e-3 * global max)."""
    gmax = max(float(np.abs(v).max()) for v in ref_grads.values())
    out = {}
    for k, g in grads.items():
        r = torch.as_tensor(ref_grads[k])
        out[k] = float((g - r).abs().max()) / max(float(r.abs().max()), 1e-3 * gmax)
    return out


def rel_l2_errors(grads, ref_grads, floor=1e-3):
    """Per-tensor relative L2 error ||g - ref|| / max(||ref||, floor * rms_all * sqrt(numel)): a statistic of the whole tensor
    (a few flipped bf16 ulps do not move it, a wrong kernel does).  The floor - `floor` times the root-mean-square over ALL
    gradient elements of the model - only matters for tensors that are analytically zero in the reference (the depthwise-conv
    bias in front of BatchRenorm), whose reference values are cancellation noise."""
    tot = sum(float(torch.as_tensor(v).double().pow(2).sum()) for v in ref_grads.values())
    cnt = sum(int(np.prod(np.shape(v))) for v in ref_grads.values())
    rms = (tot / max(cnt, 1)) ** 0.5
    out = {}
    for k, g in grads.items():
        r = torch.as_tensor(ref_grads[k]).double().reshape(-1)
        gg = torch.as_tensor(g).double().reshape(-1)
        out[k] = float((gg - r).norm()) / max(float(r.norm()), floor * rms * r.numel() ** 0.5)
    return out


def strided_like_fixture(g, cap):
    """The 'gs.' entries of the scalar fixtures (oracle/make_golden.py::strided): flatten, keep every ceil(numel/cap)-th element."""
    flat = g.reshape(-1)
    step = max(1, -(-flat.numel() // cap))
    return flat[::step]


class BlockCapture:
    """Forward hooks on the modules the reference fixtures were captured on (oracle/make_golden.py::run_case): the
    subsampler output, every layer output, and the ff1 / attend / conv BRANCHES (ConformerLayer.forward calls those modules
    with residual=True, so the branch is output - input of the f32 residual stream)."""

    def __init__(self, m):
        self.caps, self.hooks = {}, []
        self._hook(m.subsampling, 'sub.out', False)
        for i, l in enumerate(m.layers):
            self._hook(l, f'layers.{i}.out', False)
            for name in ('ff1', 'attend', 'conv'):
                self._hook(getattr(l, name), f'layers.{i}.{name}.branch', True)

    def _hook(self, mod, key, branch):
        def f(_m, inp, out):
            out = out[0] if isinstance(out, tuple) else out
            self.caps[key] = ((out - inp[0]) if branch else out).detach().float().cpu()
        self.hooks.append(mod.register_forward_hook(f))

    def remove(self):
        for h in self.hooks: h.remove()
