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


def run_step(m, fx, device='cpu'):
    """forward + CTC(sum) + the reference loss scaling (exp/train.py:275) + backward.  Returns a dict of CPU results."""
    from lcasr_amd.losses import CTCLoss
    x = torch.from_numpy(fx['x']).to(device)
    ln = torch.from_numpy(fx['lengths']).to(device)
    out = m(x, length=ln)
    lp = out['final_posteriors']
    B, _, T = x.shape
    loss = CTCLoss(blank=m.decoder.num_classes - 1, reduction='sum')(
        lp.transpose(0, 1), torch.from_numpy(fx['targets']).to(device), out['length'], torch.from_numpy(fx['target_lengths']).to(device))
    (loss / (T * B) * 100).backward()
    if device != 'cpu':
        torch.cuda.synchronize()
    return dict(logp=lp.detach().float().cpu(), length=out['length'].cpu(), loss=float(loss),
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
