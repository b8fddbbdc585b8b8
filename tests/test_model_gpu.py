"""GPU parity of the whole SConformerXL hot path (forward + CTC + backward) through libsconf_hip.so against
  (1) the golden fixtures generated from the imported reference (fp32 CPU path), and
  (2) the CPU emulation of the same bf16-storage contract (tests/kernel_refs.py), which isolates kernel bugs from
      bf16 rounding noise.

Tolerances: north_star asks for logits and CTC loss within 1e-3 "bf16 tolerance" of the reference CPU path.  The
reference's OWN bf16-autocast path differs from its fp32 path by 0.25 max / 0.04 mean abs in log-probs and 3-5e-4
relative in the loss, so: CTC loss relative error <= 1e-3 at the benchmark configs (2e-3 on the
64-wide tiny model, whose activations carry more relative bf16 noise), log-probs mean |d| <= 0.05 and max |d| <= 0.35.
"""
import numpy as np
import pytest
import torch

from common_model import (TINY_CASES, BlockCapture, build_from_fixture, grad_errors, rel_l2_errors, run_step, strided_like_fixture)
from conftest import golden_cfg, load_golden

# Gradient parity metric (round 2): per-tensor relative L2, ||g - ref|| / ||ref|| (common_model.rel_l2_errors), against the
# reference's fp32 gradients.  What bounds it is bf16 storage noise carried through the stack and amplified by every
# BatchRenorm (a division by a batch standard deviation, and in the backward the removal of the mean and of the x-hat
# component of the incoming gradient).  The calibration is the reference's OWN bf16-autocast path against its fp32 path on
# the same cases (tests/golden/ref_bf16_noise.npz, oracle/make_golden.py::bf16_noise_case): median 0.04-0.09, worst live
# tensor 0.14-0.17 on the tiny model and 0.66-0.75 at configs 1 / 2.  Measured here on MI355X: median 0.019-0.053, worst
# 0.06-0.10 (tiny), 0.12-0.14 (c1), 0.11-0.12 (c2).  The MEDIAN is itself a noisy statistic: the same model with one subsampler
# kernel exchanged for another that rounds at different points moves it from 0.020 to 0.052 (c1) and from 0.027 to 0.053
# (c2) in either direction (A/B runs of round 2) - a perturbation of the first stage shifts every gradient
# coherently.  Bounds: worst tensor 0.15 AND below the reference's own worst; median 0.065 AND below the reference's own median.  GPU against the CPU emulation of the same rounding points measures the SAME size as
# either against fp32 (0.025 median / 0.07 worst): two bf16 evaluations that differ only in accumulation order are two
# independent realisations of the rounding noise (an ulp flips where a value sits on a rounding boundary), not a tighter pair.
GRAD_L2_WORST, GRAD_L2_MEDIAN, GRAD_L2_VS_EMULATION = 0.15, 0.065, 0.15


def _check_grad_l2(tag, errs, noise_case=None):
    _report(tag, errs)
    worst, med = max(errs.values()), float(np.median(list(errs.values())))
    assert worst < GRAD_L2_WORST and med < GRAD_L2_MEDIAN, sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    if noise_case is not None:                                   # never noisier than the reference's own bf16 path
        nz = load_golden('ref_bf16_noise')
        theirs = dict(zip(nz[noise_case + '.names'].tolist(), nz[noise_case + '.grad_l2'].tolist()))
        live = [v for k, v in theirs.items() if not k.endswith('depthwise_conv.bias')]
        print(f'[{tag}] reference bf16-autocast vs its fp32: median {np.median(live):.4f}, worst {max(live):.4f}')
        assert worst <= max(live) and med <= float(np.median(live)), (worst, max(live), med, float(np.median(live)))


def _report(tag, errs):
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
    print(f'[{tag}] gradient rel-L2: median {float(np.median(list(errs.values()))):.4f}, worst ' + ', '.join(f'{k} {v:.4f}' for k, v in top))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from lcasr_amd.hip import _lib
    _lib.load()


@pytest.mark.parametrize('case', TINY_CASES)
def test_fused_head_ctc_loss_vs_reference_fixture_and_two_call_path(case):
    """model(..., ctc_targets=...) - head, log_softmax and CTC as one operator (what Trainer.step runs) - against the reference
    fixture (loss, every gradient, BatchRenorm buffers) and against this package's own two-call path (posteriors, then CTCLoss)."""
    fx = load_golden(case)
    m = build_from_fixture(fx, 'cuda')
    r = run_step(m, fx, 'cuda', fused_loss=True)
    assert r['logp'] is None and torch.equal(r['length'], torch.from_numpy(fx['out_length']))
    assert abs(r['loss'] - float(fx['loss'])) / float(fx['loss']) < 2e-3, (r['loss'], float(fx['loss']))
    _check_grad_l2(case + ' (fused loss)', rel_l2_errors(r['grads'], {k[2:]: fx[k] for k in fx.files if k.startswith('g.')}), noise_case=case)
    m2 = build_from_fixture(fx, 'cuda')
    r2 = run_step(m2, fx, 'cuda')
    assert abs(r['loss'] - r2['loss']) / r2['loss'] < 2e-6, (r['loss'], r2['loss'])
    errs = rel_l2_errors(r['grads'], r2['grads'])
    _report(case + ' fused vs two-call', errs)
    assert max(errs.values()) < 0.05, sorted(errs.items(), key=lambda kv: -kv[1])[:5]        # same bf16 dlogits up to an ulp (7e-6 at the decoder), amplified through the bf16 backward
    for k, v in r['buffers'].items():
        assert torch.equal(v, r2['buffers'][k]), k


def test_fused_passes_match_the_separate_kernels(monkeypatch):
    """norm_out + decoder norm in one kernel, rotary in the qkv GEMM epilogue and head + CTC as one operator against the same
    step with all three switched off (SCONF_NORM2=0, SCONF_QKV_ROT_EPILOGUE_OFF, two-call loss): same loss, same gradients up to
    the bf16 rounding points that move (one rounding of q, k instead of two; d loss / d logits identical)."""
    fx = load_golden('tiny_ln_ragged')
    r_on = run_step(build_from_fixture(fx, 'cuda'), fx, 'cuda', fused_loss=True)
    monkeypatch.setenv('SCONF_NORM2', '0'); monkeypatch.setenv('SCONF_QKV_ROT_EPILOGUE_OFF', '1')
    r_off = run_step(build_from_fixture(fx, 'cuda'), fx, 'cuda', fused_loss=False)
    assert abs(r_on['loss'] - r_off['loss']) / r_off['loss'] < 2e-4, (r_on['loss'], r_off['loss'])
    errs = rel_l2_errors(r_on['grads'], r_off['grads'])
    _report('fused vs separate kernels', errs)
    assert max(errs.values()) < 0.03 and float(np.median(list(errs.values()))) < 0.005, sorted(errs.items(), key=lambda kv: -kv[1])[:5]   # measured 0.010 / 0.001


@pytest.mark.parametrize('case', TINY_CASES)
def test_tiny_model_vs_reference_fixture(case):
    fx = load_golden(case)
    m = build_from_fixture(fx, 'cuda')
    r = run_step(m, fx, 'cuda')
    assert torch.equal(r['length'], torch.from_numpy(fx['out_length']))
    d = (r['logp'] - torch.from_numpy(fx['logp'])).abs()
    assert float(d.max()) < 0.35 and float(d.mean()) < 0.05, (float(d.max()), float(d.mean()))
    assert abs(r['loss'] - float(fx['loss'])) / float(fx['loss']) < 2e-3, (r['loss'], float(fx['loss']))
    errs = rel_l2_errors(r['grads'], {k[2:]: fx[k] for k in fx.files if k.startswith('g.')})
    _check_grad_l2(case, errs, noise_case=case)
    for k in fx.files:
        if k.startswith('buf.'):
            got = m.state_dict()[k[4:]].float().cpu()
            assert float((got - torch.from_numpy(fx[k]).float()).abs().max()) < 2e-3, k


@pytest.mark.parametrize('case', TINY_CASES + ['tiny_ln_brn'])
def test_per_block_outputs_vs_reference_captures(case):
    """The fixtures carry the reference's output of every block (forward hooks in oracle/make_golden.py): subsampler, the ff1 /
    attention / conv-module branches and the output of each layer.  Same hooks on the same modules here.  Blocks in front of the
    first BatchRenorm (subsampler, layer-0 ff1 and attention) see only bf16 operand rounding: <= 1.5e-2 of the block's max
    (measured 0.5-1.0e-2).  From the first conv module on, the BatchRenorm's division by a batch standard deviation amplifies
    the rounding of its input (the reference's own bf16-autocast path does the same): relative L2 <= 0.10, max <= 0.12 of the
    block's max (measured with the CPU emulation of the same rounding points: 0.03-0.07)."""
    fx = load_golden(case)
    m = build_from_fixture(fx, 'cuda')
    bc = BlockCapture(m)
    run_step(m, fx, 'cuda')
    bc.remove()
    keys = [k[4:] for k in fx.files if k.startswith('cap.')]
    assert sorted(keys) == sorted(bc.caps), (sorted(keys), sorted(bc.caps))
    rows = []
    for k in keys:
        ref, got = torch.from_numpy(fx['cap.' + k]), bc.caps[k]
        mx = float((got - ref).abs().max()) / float(ref.abs().max())
        l2 = float((got - ref).norm() / ref.norm())
        rows.append((k, mx, l2))
        clean = k in ('sub.out', 'layers.0.ff1.branch', 'layers.0.attend.branch')
        assert mx < (1.5e-2 if clean else 0.12) and l2 < (1.2e-2 if clean else 0.10), (k, mx, l2)
    print(f'[{case}] per-block (max/|ref|max, rel-L2): ' + ', '.join(f'{k} {a:.4f}/{b:.4f}' for k, a, b in rows))


@pytest.mark.parametrize('case', TINY_CASES)
def test_ctc_gradient_vs_reference_dlogp(case):
    """`dlogp` in the fixtures is the reference's d(scaled loss)/d(log-probs) (torch.nn.CTCLoss backward).  (1) The HIP CTC
    kernels fed the REFERENCE's log-probs must reproduce it to f32 accuracy: <= 2e-3 of its max (measured ~1e-5).  (2) End to
    end (the HIP model's own log-probs) the difference is the log-prob noise times the posterior: relative L2 reported, <= 0.15."""
    from lcasr_amd.losses import CTCLoss
    fx = load_golden(case)
    ref = torch.from_numpy(fx['dlogp'])
    B, _, T = fx['x'].shape
    lp = torch.from_numpy(fx['logp']).cuda().requires_grad_(True)
    tg, tl = torch.from_numpy(fx['targets']).cuda(), torch.from_numpy(fx['target_lengths']).cuda()
    ol = torch.from_numpy(fx['out_length']).cuda()
    loss = CTCLoss(blank=lp.shape[-1] - 1, reduction='sum')(lp.transpose(0, 1), tg, ol, tl)
    (loss / (T * B) * 100).backward()
    assert abs(float(loss) - float(fx['loss'])) / float(fx['loss']) < 1e-5
    d = float((lp.grad.cpu() - ref).abs().max()) / float(ref.abs().max())
    assert d < 2e-3, d
    m = build_from_fixture(fx, 'cuda')
    out = m(torch.from_numpy(fx['x']).cuda(), length=torch.from_numpy(fx['lengths']).cuda())
    lp2 = out['final_posteriors']; lp2.retain_grad()
    loss2 = CTCLoss(blank=lp.shape[-1] - 1, reduction='sum')(lp2.transpose(0, 1), tg, out['length'], tl)
    (loss2 / (T * B) * 100).backward()
    l2 = float((lp2.grad.cpu() - ref).norm() / ref.norm())
    print(f'[{case}] dlogp: CTC kernels on reference log-probs max|d|/max {d:.2e}; end-to-end rel-L2 {l2:.4f}')
    assert l2 < 0.15, l2


@pytest.mark.parametrize('case', ['tiny_ln_ragged', 'tiny_rms_ragged'])
def test_tiny_model_vs_cpu_emulation(case, monkeypatch):
    """Same bf16 rounding points on both sides: what remains is accumulation order and fast exp/tanh, amplified
    where a value sits on a bf16 rounding boundary (one flipped bf16 ulp = 0.4 % of that activation)."""
    import kernel_refs
    import lcasr_amd.functional as Fn
    fx = load_golden(case)
    r_gpu = run_step(build_from_fixture(fx, 'cuda'), fx, 'cuda')
    monkeypatch.setattr(Fn, 'ops', kernel_refs)
    Fn.clear_weight_cache()
    r_cpu = run_step(build_from_fixture(fx, 'cpu'), fx, 'cpu')
    Fn.clear_weight_cache()
    d = (r_gpu['logp'] - r_cpu['logp']).abs()
    assert float(d.max()) < 0.2 and float(d.mean()) < 0.02, (float(d.max()), float(d.mean()))
    assert abs(r_gpu['loss'] - r_cpu['loss']) / r_cpu['loss'] < 5e-4
    errs = rel_l2_errors(r_gpu['grads'], {k: v.numpy() for k, v in r_cpu['grads'].items()})
    _report(case + ' vs emulation', errs)
    assert max(errs.values()) < GRAD_L2_VS_EMULATION, sorted(errs.items(), key=lambda kv: -kv[1])[:5]


def test_c1_config_from_seed():
    """BASELINE config 1 (6L/256D/8H, B=2, T=1024): weights from torch.manual_seed(12345), reference loss 1882.50."""
    from lcasr_amd.losses import CTCLoss
    from lcasr_amd.models.sconformer_xl import SCConformerXL
    fx = load_golden('c1_scalars')
    torch.manual_seed(12345)
    m = SCConformerXL(**golden_cfg(fx)).cuda().train()
    x = torch.from_numpy(fx['x']).cuda()
    out = m(x, length=torch.from_numpy(fx['lengths']).cuda())
    lp = out['final_posteriors']
    loss = CTCLoss(blank=4095, reduction='sum')(lp.transpose(0, 1), torch.from_numpy(fx['targets']).cuda(), out['length'],
                                                torch.from_numpy(fx['target_lengths']).cuda())
    (loss / (1024 * 2) * 100).backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - float(fx['loss'])) / float(fx['loss']) < 1e-3, (float(loss), float(fx['loss']))
    d = (lp[:, ::17, ::97].float().cpu() - torch.from_numpy(fx['logp_slice'])).abs()
    assert float(d.max()) < 0.35 and float(d.mean()) < 0.05
    _check_scalar_fixture_grads('c1', fx, m)


def _check_scalar_fixture_grads(tag, fx, m):
    """c1 / c2 fixtures: per-tensor gradient norms, and a strided sample (<= 8192 elements, oracle/make_golden.py::strided) of
    EVERY gradient tensor of the reference: relative L2 of the sample against the same sample of the HIP gradient."""
    ref = {k[6:]: float(fx[k]) for k in fx.files if k.startswith('gnorm.')}
    big = max(ref.values())
    for k, p in m.named_parameters():
        gn = float(p.grad.double().norm())
        if ref[k] > 0.01 * big:
            assert abs(gn - ref[k]) / ref[k] < 0.10, (k, gn, ref[k])
    cap = int(fx['gs_cap'])
    got = {k: strided_like_fixture(p.grad.detach().float().cpu(), cap) for k, p in m.named_parameters()}
    errs = rel_l2_errors(got, {k[3:]: fx[k] for k in fx.files if k.startswith('gs.')})
    assert len(errs) == len(list(m.parameters()))
    _check_grad_l2(tag, errs, noise_case=tag + '_scalars')


def test_c2_config_from_seed():
    """BASELINE config 2 (6L/768D/6H, T=1024 -> N=128 tokens, B=2): the only end-to-end run of the FOUR-wave head_dim-128
    attention kernels (N < 256) and of the 768-wide model at a reference-checkable size.  Reference loss 1897.02."""
    from lcasr_amd.losses import CTCLoss
    from lcasr_amd.models.sconformer_xl import SCConformerXL
    fx = load_golden('c2_scalars')
    torch.manual_seed(12345)
    m = SCConformerXL(**golden_cfg(fx)).cuda().train()
    assert m.d_model == 768 and m.head_dim == 128
    out = m(torch.from_numpy(fx['x']).cuda(), length=torch.from_numpy(fx['lengths']).cuda())
    lp = out['final_posteriors']
    assert lp.shape == (2, 128, 4096)
    loss = CTCLoss(blank=4095, reduction='sum')(lp.transpose(0, 1), torch.from_numpy(fx['targets']).cuda(), out['length'],
                                                torch.from_numpy(fx['target_lengths']).cuda())
    (loss / (1024 * 2) * 100).backward()
    torch.cuda.synchronize()
    rel = abs(float(loss) - float(fx['loss'])) / float(fx['loss'])
    d = (lp[:, ::17, ::97].float().cpu() - torch.from_numpy(fx['logp_slice'])).abs()
    print(f'[c2] loss {float(loss):.3f} vs {float(fx["loss"]):.3f} (rel {rel:.2e}); log-prob slice max|d| {float(d.max()):.3f} mean {float(d.mean()):.4f}')
    assert rel < 1e-3, (float(loss), float(fx['loss']))
    assert float(d.max()) < 0.35 and float(d.mean()) < 0.05
    _check_scalar_fixture_grads('c2', fx, m)


def test_c4_activation_checkpointing_matches_reference_and_plain_run():
    """BASELINE config 4's switches (checkpoint_every_n_layers=1, ff_checkpoint_lvl=2; exp_set_seq_rotary_base_9l.yaml:52-53)
    on the tiny model with live BatchRenorm clamps (fixtures tiny_ln_ckpt / tiny_ln_brn, generated from the reference with and
    without the switches).  What the reference does, and this must mirror: same forward values; the layer is run a second time
    inside the backward, in train mode, so BatchRenorm's running statistics move TWICE per step (num_batches_tracked += 2) and
    the recompute's r / d clamps are taken from the once-moved statistics - the gradients are those of the recompute."""
    fc, fp = load_golden('tiny_ln_ckpt'), load_golden('tiny_ln_brn')
    mc, mp = build_from_fixture(fc, 'cuda'), build_from_fixture(fp, 'cuda')
    assert mc.checkpoint_every_n_layers == 1 and mc.layers[0].ff1.fn.fn.checkpoint_lvl == 2 and mp.checkpoint_every_n_layers == 0
    rc, rp = run_step(mc, fc, 'cuda'), run_step(mp, fp, 'cuda')
    assert torch.equal(rc['logp'], rp['logp']) and rc['loss'] == rp['loss']          # checkpointing never changes a forward value
    for r, fx, m, tag in ((rc, fc, mc, 'ckpt'), (rp, fp, mp, 'plain')):
        assert abs(r['loss'] - float(fx['loss'])) / float(fx['loss']) < 2e-3
        errs = rel_l2_errors(r['grads'], {k[2:]: fx[k] for k in fx.files if k.startswith('g.')})
        _report('c4-switches ' + tag, errs)
        assert max(errs.values()) < GRAD_L2_WORST, sorted(errs.items(), key=lambda kv: -kv[1])[:5]
        for k in fx.files:
            if k.startswith('buf.'):
                got = m.state_dict()[k[4:]].float().cpu()
                assert float((got - torch.from_numpy(fx[k]).float()).abs().max()) < 2e-3, (tag, k)
    nbt = [int(l.conv.fn.batch_norm.num_batches_tracked) for l in mc.layers]
    assert nbt == [30002, 30002] and [int(l.conv.fn.batch_norm.num_batches_tracked) for l in mp.layers] == [30001, 30001]
    # the two runs differ where the reference's do: the running statistics (moved twice vs once)
    d = float((mc.layers[0].conv.fn.batch_norm.running_mean - mp.layers[0].conv.fn.batch_norm.running_mean).abs().max())
    assert d > 1e-3, d


def test_c4_shape_trainer_step_with_checkpointing_matches_plain():
    """BASELINE config 4 itself (9L/768D/6H, T = 16384, per-layer checkpointing + ff_checkpoint_lvl 2) through the TRAINING DRIVER (direct gradient writes into the flat
    buffer, parked bf16 twins, BatchRenorm buffers mutated in forward) - the combination torch.utils.checkpoint could break
    silently.  Fresh BatchRenorm buffers (r = 1, d = 0 whatever the statistics), so the recompute reproduces the forward
    bit for bit and the step must equal the un-checkpointed one: loss equal, updated parameters equal up to atomics order."""
    from lcasr_amd.models.sconformer_xl import SCConformerXL
    from lcasr_amd.train import Trainer, synthetic_batch
    # (round 3: the full 9-layer shape of exp_set_seq_rotary_base_9l.yaml:27-54, so that no BASELINE shape is test-virgin; round 2 ran 3 layers)
    base = dict(vocab_size=4095, n_layers=9, d_model=768, n_heads=6, head_dim=128, use_rotary=True, rotary_base_freq=1500000,
                decoder_norm=True, self_conditioning=True, default_norm='layer_norm')
    res = []
    for extra in (dict(), dict(checkpoint_every_n_layers=1, ff_checkpoint_lvl=2)):
        torch.manual_seed(12345)
        m = SCConformerXL(**base, **extra).cuda().train()
        tr = Trainer(m, lr=3e-3, global_batch=4)
        batch = synthetic_batch(4, 16384, 4095, seed=5)            # 8192 tokens: activations (not the 1.2 GB of state) set the peak
        torch.cuda.reset_peak_memory_stats()
        loss = float(tr.step(*batch))
        res.append((loss, tr.opt.flat[0].data.clone(), [int(l.conv.fn.batch_norm.num_batches_tracked) for l in m.layers],
                    torch.cuda.max_memory_allocated()))
        del tr, m
    (l0, p0, n0, mem0), (l1, p1, n1, mem1) = res
    assert l0 == l1, (l0, l1)
    d = (p0 - p1).abs()
    assert float(d.max()) <= 2e-2 and float(d.mean()) <= 1e-4, (float(d.max()), float(d.mean()))
    assert n0 == [1] * 9 and n1 == [2] * 9
    assert mem1 < mem0, (mem0, mem1)                                                 # and it does save activation memory


def test_eval_mode_and_determinism():
    fx = load_golden('tiny_ln_ragged')
    m = build_from_fixture(fx, 'cuda').eval()
    x, ln = torch.from_numpy(fx['x']).cuda(), torch.from_numpy(fx['lengths']).cuda()
    with torch.no_grad():
        a = m(x, length=ln)['final_posteriors']
        b = m(x, length=ln)['final_posteriors']
    assert torch.equal(a, b), 'forward is not bitwise reproducible'
    assert float((a.exp().sum(-1) - 1).abs().max()) < 1e-3


def test_full_size_properties_c3_shape():
    """BASELINE config 3 shape (6L/768D/6H, T=16384, B=1): size-independent properties of the hot path."""
    from lcasr_amd.losses import CTCLoss
    from lcasr_amd.models.sconformer_xl import SCConformerXL
    torch.manual_seed(12345)
    m = SCConformerXL(vocab_size=4095, n_layers=6, d_model=768, n_heads=6, head_dim=128, use_rotary=True, rotary_base_freq=1500000,
                      decoder_norm=True, self_conditioning=True, default_norm='layer_norm').cuda().train()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 80, 16384, generator=g).cuda()
    tg = torch.randint(0, 4095, (1, 512), generator=g).cuda()
    out = m(x)
    lp = out['final_posteriors']
    assert lp.shape == (1, 2048, 4096) and int(out['length'][0]) == 2048
    assert float((lp.exp().sum(-1) - 1).abs().max()) < 1e-3                       # rows are distributions
    loss = CTCLoss(blank=4095, reduction='sum')(lp.transpose(0, 1), tg, out['length'], torch.tensor([512]).cuda())
    lp.retain_grad()
    loss.backward()
    torch.cuda.synchronize()
    assert np.isfinite(float(loss)) and float(loss) > 0
    # CTC grad rows sum to ~0.  Round 2 allowed 6e-2 here (plain f32 log-space rows drift with |alpha| ~ 8 N, as
    # torch's f32 op does); the lattice rows are renormalised per frame now (csrc/ctc.hip) and the sums are exact to 1e-3.
    assert float(lp.grad.sum(-1).abs().max()) < 1e-3
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())


@pytest.mark.gpu
def test_benchmark_shape_step_same_through_both_gemm_kernels(monkeypatch):
    """BASELINE config 3 at the benchmark's token count (B = 8 x T = 16384 -> 16384 tokens per GEMM, where the 256-row kernels
    take every projection, dgrad and wgrad): one Trainer step with them (default) and one with the 128x128 kernel forced.  The
    GEMMs are bit-identical, so loss and updated parameters may differ only by the order of the float atomics elsewhere."""
    from lcasr_amd.models.sconformer_xl import SCConformerXL
    from lcasr_amd.train import Trainer, synthetic_batch
    res = []
    for force_old in (False, True):
        if force_old: monkeypatch.setenv('SCONF_GEMM_NO_256', '1')
        else: monkeypatch.delenv('SCONF_GEMM_NO_256', raising=False)
        torch.manual_seed(12345)
        m = SCConformerXL(vocab_size=4095, n_layers=6, d_model=768, n_heads=6, head_dim=128, use_rotary=True, rotary_base_freq=1500000,
                          decoder_norm=True, self_conditioning=True, default_norm='layer_norm').cuda().train()
        tr = Trainer(m, lr=3e-3, global_batch=8)
        batch = synthetic_batch(8, 16384, 4095, seed=3)
        losses = [float(tr.step(*batch)) for _ in range(2)]
        res.append((losses, tr.opt.flat[0].data.clone()))
        del tr, m
    (l_new, p_new), (l_old, p_old) = res
    assert all(np.isfinite(l_new)) and abs(l_new[0] - l_old[0]) <= 1e-5 * abs(l_old[0]), (l_new, l_old)
    assert abs(l_new[1] - l_old[1]) <= 2e-3 * abs(l_old[1]), (l_new, l_old)      # after one optimiser step on each side
    d = (p_new - p_old).abs()
    assert float(d.max()) <= 2e-2 and float(d.mean()) <= 1e-4, (float(d.max()), float(d.mean()))


@pytest.mark.gpu
def test_fetch_logits_batched_windows_and_greedy_on_gpu():
    """f3: sliding-window inference through the HIP path vs the reference's fetch_logits output (fixture), batched windows
    vs one-at-a-time, and greedy decoding (HIP argmax) vs the reference decoder's token ids."""
    from lcasr_amd.eval.utils import fetch_logits
    from lcasr_amd.decoding.greedy import GreedyCTCDecoder
    fx = load_golden('infer_tiny')
    m = build_from_fixture(fx, 'cuda').eval()

    class Tok:
        def vocab_size(self): return int(fx['cfg.vocab_size'])

    class Args: config = {'audio_chunking': {'size': 512, 'overlap': 128}}

    spec = torch.from_numpy(fx['spec'])
    dec = GreedyCTCDecoder(tokenizer=None, blank_id=m.decoder.num_classes - 1)
    for ci, (sl, ov) in enumerate(fx['cases'].tolist()):
        seq = fetch_logits(Args, m, spec, sl, ov, Tok(), use_tqdm=False, batched=False)
        bat = fetch_logits(Args, m, spec, sl, ov, Tok(), use_tqdm=False, batched=True, max_batch=3)
        ref = fx[f'logits.{ci}']
        assert seq.shape == ref.shape == bat.shape
        d = np.abs(seq - ref)
        assert float(d.max()) < 0.3 and float(d.mean()) < 0.03, (sl, ov, float(d.max()), float(d.mean()))
        assert float(np.abs(seq - bat).max()) < 2e-3, float(np.abs(seq - bat).max())   # same kernels, other batch size
        assert dec(torch.from_numpy(ref).cuda(), decode=False) == fx[f'greedy.{ci}'].tolist()


@pytest.mark.gpu
def test_train_recording_chunked_ragged_shrinking_batch():
    """f2: a batch of long recordings of different lengths, chunk by chunk through the HIP path: rows drop out as recordings
    end, last chunks are ragged (per-sample lengths), the loss normaliser is the constant chunk_size * batch."""
    from lcasr_amd.train import Trainer
    from lcasr_amd.utils.dataloading import chunk_spectogram, plan_chunks
    from oracle import sconformer_ref as O
    fx = load_golden('tiny_ln_ragged')
    m = build_from_fixture(fx, 'cuda')
    sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    tr = Trainer(m, lr=1e-3, global_batch=3)
    g = torch.Generator().manual_seed(1)
    lens = torch.tensor([1000, 530, 256])
    audio = torch.randn(3, 80, 1000, generator=g)
    for b, l in enumerate(lens.tolist()): audio[b, :, l:] = 0
    V = int(fx['cfg.vocab_size'])
    seen = []

    def targets(ix, c):
        n = c['audio'].shape[0]
        tl = ((c['audio_lengths'] // 8) // 4).clamp(min=1)
        tg = torch.randint(0, V, (n, int(tl.max())), generator=g)
        seen.append((ix, n, c['audio'].shape[-1], c['audio_lengths'].tolist(), tg, tl))
        return tg, tl

    losses = tr.train_recording(audio.cuda(), lens.cuda(), 256, 64, lambda ix, c: tuple(t.cuda() for t in targets(ix, {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in c.items()})))
    plan = plan_chunks(chunk_spectogram(audio, 256, 64), lens.clone(), 64)
    assert [s[1] for s in seen] == [int(p['selection_mask'].sum()) for p in plan] == [3, 3, 2, 1, 1, 1]
    assert seen[-1][2] == 40 and seen[2][3] == [256, 146]                   # ragged last chunk; ragged lengths inside a chunk
    assert len(losses) == len(plan) and all(torch.isfinite(l) for l in losses)
    # first chunk == one oracle step on the same weights / inputs
    cfg = O.make_config(**golden_cfg(fx))
    oloss, _, _ = O.train_step_loss(sd0, cfg, audio[:, :, :256], torch.tensor(seen[0][3]), seen[0][4], seen[0][5])
    assert abs(float(losses[0]) - float(oloss)) / float(oloss) < 3e-3, (float(losses[0]), float(oloss))


def test_subsampler_over_2pow31_elements_matches_two_halves():
    """At 128 x 16384-frame samples the stage-1 tensors of the subsampler hold 2.7e9 elements (> 2^31): every offset that
    spans samples must be 64-bit.  The subsampler has no cross-sample operation, so the full batch must reproduce the two
    halves: outputs row for row, parameter gradients as their sum."""
    import lcasr_amd.functional as Fn
    from lcasr_amd.components.subsampling import ConvSubsampling
    torch.manual_seed(3)
    sub = ConvSubsampling(subsampling='dw_striding', conv_channels=256, activation=torch.nn.SiLU(), subsampling_factor=8, feat_in=80,
                          feat_out=768, norm_out=False).cuda()
    B, T = 128, 16384
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, T, 80, generator=g).cuda()                                   # (B, T, F) as ConvSubsampling.forward takes it
    gy = (torch.randn(B, T // 8, 768, generator=g) * 0.1).cuda()
    lengths = torch.full((B,), T, device='cuda')

    def run(sl):
        for p in sub.parameters(): p.grad = None
        Fn.refresh_weight_shadows()
        y, _ = sub(x[sl], lengths=lengths[sl])
        y.backward(gy[sl])
        torch.cuda.synchronize()
        return y.detach(), [p.grad.detach().clone() for p in sub.parameters()]

    y_full, g_full = run(slice(0, B))
    assert y_full.numel() * 0 == 0 and torch.isfinite(y_full).all()
    y_a, g_a = run(slice(0, B // 2))
    y_b, g_b = run(slice(B // 2, B))
    assert torch.equal(y_full[:B // 2], y_a) and torch.equal(y_full[B // 2:], y_b)          # no cross-sample arithmetic at all
    for (n, _), gf, ga, gb in zip(sub.named_parameters(), g_full, g_a, g_b):
        ref = ga.double() + gb.double()
        err = float((gf.double() - ref).abs().max()) / (float(ref.abs().max()) + 1e-12)
        assert err < 2e-3, (n, err)                                                           # split-K / atomic order only


def test_full_size_properties_c5_shape():
    """BASELINE config 5 shape (3L/2048D/16H, 512 subsampler channels, T=131072 -> N=16384 tokens, B=1, ff_checkpoint_lvl 2):
    the 20-minute context through forward + CTC + backward; size-independent properties (no reference value exists at this size:
    the reference's CPU attention alone would need a 17 GB score matrix per layer)."""
    from lcasr_amd.losses import CTCLoss
    from lcasr_amd.models.sconformer_xl import SCConformerXL
    torch.manual_seed(12345)
    m = SCConformerXL(vocab_size=4095, n_layers=3, d_model=2048, n_heads=16, head_dim=128, subsampling_conv_channels=512, use_rotary=True,
                      rotary_base_freq=1500000, decoder_norm=True, self_conditioning=True, default_norm='layer_norm',
                      ff_checkpoint_lvl=2).cuda().train()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 80, 131072, generator=g).cuda()
    S = 4096
    tg = torch.randint(0, 4095, (1, S), generator=g).cuda()
    out = m(x)
    lp = out['final_posteriors']
    assert lp.shape == (1, 16384, 4096) and int(out['length'][0]) == 16384
    assert float((lp.exp().sum(-1) - 1).abs().max()) < 1e-3                       # rows are distributions
    loss = CTCLoss(blank=4095, reduction='sum')(lp.transpose(0, 1), tg, out['length'], torch.tensor([S]).cuda())
    lp.retain_grad()
    loss.backward()
    torch.cuda.synchronize()
    assert np.isfinite(float(loss)) and float(loss) > 0
    # CTC gradient rows sum to ~0.  Round 2 observed 0.36-0.50 here and allowed 0.75: plain f32 log-space alpha/beta reach
    # |-8.3 * 16384| ~ 1.4e5, where one f32 ulp is 1.6e-2 (torch's f32 CTC has the same limit).  With the lattice rows
    # renormalised per frame and the offsets in f64 the sums are exact to 1e-3 at this length too (value-level parity against the
    # f64 oracle: tests/test_kernels_gpu.py::test_ctc_at_the_131072_frame_context_against_the_f64_oracle).
    assert float(lp.grad.sum(-1).abs().max()) < 1e-3
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())
    # the forward is deterministic: same input, same weights -> same bits (eval of determinism at the 131072-frame size)
    with torch.no_grad():
        for l in m.layers: l.conv.fn.batch_norm.num_batches_tracked.zero_(); l.conv.fn.batch_norm.running_mean.zero_(); l.conv.fn.batch_norm.running_std.fill_(1.0)
        a = m(x)['final_posteriors']
        for l in m.layers: l.conv.fn.batch_norm.num_batches_tracked.zero_(); l.conv.fn.batch_norm.running_mean.zero_(); l.conv.fn.batch_norm.running_std.fill_(1.0)
        b = m(x)['final_posteriors']
    assert torch.equal(a, b)


def test_checkpoint_resume_on_device(tmp_path):
    """f4 (general.py:97-172) through the HIP path: Trainer 2 steps -> save_model -> a NEW model (different init) + new MADGRAD
    -> load_checkpoint -> step 3 equals the uninterrupted run's step 3.  The optimiser state lives in flat device buffers and the
    parameters are views into one; both must survive the round trip through the reference's per-parameter state_dict layout."""
    from lcasr_amd.models.sconformer_xl import SCConformerXL
    from lcasr_amd.train import Trainer, synthetic_batch
    from lcasr_amd.utils.general import load_checkpoint, save_model
    fx = load_golden('tiny_ln_ragged')
    cfg = golden_cfg(fx)
    batches = [synthetic_batch(2, 256, int(fx['cfg.vocab_size']), seed=s) for s in (1, 2, 3)]

    def fresh(seed):
        torch.manual_seed(seed)
        return Trainer(SCConformerXL(**cfg).cuda().train(), lr=3e-3, global_batch=2)

    a = fresh(12345)
    la = [float(a.step(*b)) for b in batches]
    pa = a.opt.flat[0].data.clone()
    b_ = fresh(12345)
    lb = [float(b_.step(*b)) for b in batches[:2]]
    config = {'checkpointing': {'dir': str(tmp_path)}}
    save_model(b_.model, b_.opt, None, 2, config, seen_ids=[4, 5], epoch=0)
    c = fresh(999)                                                               # other weights, empty optimiser state
    seen, step, epoch = load_checkpoint(None, c.model, c.opt, path=str(tmp_path), device='cuda')
    assert (seen, step, epoch) == ([4, 5], 2, 0) and c.opt.k == 2
    assert torch.equal(c.opt.flat[0].data, b_.opt.flat[0].data)                  # parameters are still views into the flat buffer
    for key in ('_gss', '_s', '_x0'):
        assert torch.equal(c.opt.param_groups[0][key], b_.opt.param_groups[0][key]), key
    for (k, v), (_, w) in zip(b_.model.state_dict().items(), c.model.state_dict().items()):
        assert torch.equal(v, w), k                                              # incl. BatchRenorm running buffers
    lc = float(c.step(*batches[2]))
    assert lb == la[:2]
    assert abs(lc - la[2]) <= 1e-6 * abs(la[2]), (lc, la[2])
    d = (c.opt.flat[0].data - pa).abs()
    assert float(d.max()) <= 1e-4, float(d.max())                                # float atomics order in the per-channel reductions only


def test_module_level_forward_seams_and_weight_shadow_staleness():
    """The sub-module `forward()`s of the reference (attention.py:509, fused_dense.py:489, convolution.py:103) take an already
    normalised input; PreNorm(fn)(x) == fn(norm(x)).  And a module-level call after an in-place weight edit / an optimiser step
    must see the new weights (bf16 shadows re-cast on use, not only at the next whole-model forward)."""
    import lcasr_amd.functional as Fn
    fx = load_golden('tiny_ln_ragged')
    m = build_from_fixture(fx, 'cuda')
    l0 = m.layers[0]
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 32, 64, generator=g).cuda()
    lens = torch.tensor([32, 20], device='cuda', dtype=torch.int32)
    pad = torch.arange(32, device='cuda')[None, :] >= lens[:, None]
    cos, sin = m.rotary_pos_emb(32, x.device)

    class Rot: pass
    rot = Rot(); rot.cos, rot.sin, rot.learned = cos, sin, False
    xn = l0.attend.norm(x)
    a_mod = l0.attend.fn(xn, attn_mask=~pad, length=lens, pad_mask=pad, rotary_emb_fn=rot)
    a_pre = l0.attend(x, lengths=lens, rotary=m.rotary_pos_emb.tables(32, x.device))
    assert float((a_mod - a_pre).abs().max()) < 2e-2 * float(a_pre.abs().max())          # xn passes through one more bf16 rounding
    f_mod = l0.ff1.fn.fn(l0.ff1.fn.norm(x)) * 0.5
    assert float((f_mod - l0.ff1(x)).abs().max()) < 2e-2 * float(f_mod.abs().max())
    c_mod = l0.conv.fn(l0.conv.norm(x), pad_mask=pad)
    c_pre = l0.conv(x, lengths=lens)
    assert float((c_mod - c_pre).abs().max()) < 3e-2 * float(c_pre.abs().max())
    # gradients flow through the seam
    xr = xn.detach().requires_grad_(True)
    l0.attend.fn(xr, pad_mask=pad, rotary_emb_fn=rot).sum().backward()
    assert torch.isfinite(xr.grad).all() and float(xr.grad.abs().sum()) > 0 and l0.attend.fn.qkv_proj.weight.grad is not None
    # staleness: edit a weight in place, call the module again without a model forward in between
    before = l0.ff1(x).detach().clone()
    with torch.no_grad():
        l0.ff1.fn.fn.fc2.weight.mul_(2.0)
    after = l0.ff1(x).detach()
    assert float((after - 2 * before).abs().max()) < 2e-2 * float(after.abs().max())
    Fn.bump_weight_epoch()                                                              # what MADGRAD.step does after its kernel
    l0.ff1.fn.fn.fc2.weight.data.mul_(0.5)                                              # .data edits do not bump the version counter
    again = l0.ff1(x).detach()
    assert float((again - before).abs().max()) < 2e-2 * float(before.abs().max())


def test_ctc_rejects_or_poisons_invalid_arguments():
    """torch.nn.CTCLoss checks its arguments on the host.  Here host tensors are checked the same way (ValueError); device
    tensors are not read back (that would stall the launch queue): the kernels poison the offending sample - nll NaN, its
    gradient rows NaN - and never index with the bad value; the other samples are unaffected."""
    from lcasr_amd.losses import CTCLoss
    g = torch.Generator().manual_seed(2)
    lp = torch.log_softmax(torch.randn(3, 40, 32, generator=g), -1).cuda().requires_grad_(True)
    ctc = CTCLoss(blank=31, reduction='none')
    tg = torch.randint(0, 31, (3, 6), generator=g)
    il, tl = torch.tensor([40, 30, 40]), torch.tensor([6, 4, 5])
    ok = ctc(lp.transpose(0, 1), tg.cuda(), il.cuda(), tl.cuda())
    for bad_t, bad_il, bad_tl in ((tg.clone().index_put_((torch.tensor([1]), torch.tensor([2])), torch.tensor(32)), il, tl),
                                  (tg, torch.tensor([40, 41, 40]), tl), (tg, il, torch.tensor([6, 7, 5]))):
        with pytest.raises(ValueError):
            ctc(lp.transpose(0, 1), bad_t, bad_il, bad_tl)                                # host tensors: checked
        nll = ctc(lp.transpose(0, 1), bad_t.cuda(), bad_il.cuda(), bad_tl.cuda())         # device tensors: poisoned sample
        assert torch.isnan(nll[1]) and torch.equal(nll[[0, 2]], ok[[0, 2]])
        gr, = torch.autograd.grad(nll[[0, 2]].sum() + nll[1], lp)
        assert torch.isnan(gr[1, :30]).all() and torch.isfinite(gr[0]).all() and torch.isfinite(gr[2]).all()


@pytest.mark.parametrize('fused_loss', [False, True])
def test_reference_default_vocabulary_129_classes_on_the_device(fused_loss):
    """the reference's default vocab_size=128 (129 classes, sconformer_xl.py:34) runs on the HIP path with the class
    dimension padded to 144 inside the decoder (zero weight rows, -1e30 bias): output shapes are the reference's, loss, log-probs
    and every gradient match the oracle on the same weights (the CPU twin with emulated ops is in test_host_logic.py)."""
    from test_host_logic import _run_129_class_step
    loss, ref_loss, lp, ref_lp, grads, ref_grads = _run_129_class_step('cuda', fused_loss)
    torch.cuda.synchronize()
    assert abs(loss - ref_loss) / ref_loss < 2e-3, (loss, ref_loss)
    if lp is not None:
        d = (lp.float().cpu() - ref_lp).abs()
        assert float(d.mean()) < 0.05 and float(d.max()) < 0.35, (float(d.mean()), float(d.max()))
    errs = rel_l2_errors(grads, ref_grads)
    _report(f'129 classes (fused_loss={fused_loss})', errs)
    assert max(errs.values()) < GRAD_L2_WORST, sorted(errs.items(), key=lambda kv: -kv[1])[:5]


def test_two_ranks_on_the_hip_path_match_single_process_sum_of_shards(tmp_path):
    """VERDICT r2 #6a: the data-parallel step on the REAL kernels.  Two fresh child processes (`python -m torch.distributed.run`,
    started with subprocess - never an exec of this GPU-initialised process) share the one GPU of the box over gloo
    (tests/ddp_gpu_worker.py), each back-propagating its shard of a 4-sample batch with direct gradient writes into its flat
    buffer + bucketed all-reduce + fused MADGRAD, 3 steps.  Rank 0's parameters must equal one process that back-propagates the
    two shards one after the other into the same gradient buffer (what tests/test_ddp_gloo.py checks on the CPU with emulated
    ops).  No RCCL here (one device); the collective calls, their order and the hooks are the same code."""
    import os
    import socket
    import subprocess
    import sys
    from ddp_gpu_worker import global_batch
    from lcasr_amd.losses import CTCLoss
    from lcasr_amd.optim import MADGRAD
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]
    out = str(tmp_path / 'r')
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), os.path.join(here, 'ddp_gpu_worker.py'), out], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    r0, r1 = torch.load(out + '.0'), torch.load(out + '.1')
    assert torch.equal(r0['data'], r1['data']), 'replicas diverged'
    fx = load_golden('tiny_ln_ragged')
    m, V = build_from_fixture(fx, 'cuda'), int(fx['cfg.vocab_size'])
    opt = MADGRAD(m.parameters(), lr=3e-3)
    x, ln, tg, tl = global_batch(V)
    ref_losses = []
    for _ in range(3):
        step = []
        for sl in (slice(0, 2), slice(2, 4)):
            loss = m(x[sl].cuda(), length=ln[sl].cuda(), ctc_targets=(tg[sl].cuda(), tl[sl].cuda()))['ctc_nll'].sum()   # Trainer.step's operator
            (loss / (256 * 4) * 100).backward()
            step.append(float(loss))
        opt.step(max_norm=0.8); opt.zero_grad()
        ref_losses.append(step)
    torch.cuda.synchronize()
    assert ref_losses[0][0] == pytest.approx(r0['losses'][0], rel=1e-5) and ref_losses[0][1] == pytest.approx(r1['losses'][0], rel=1e-5)
    assert r0['nbt'] == 3                                                       # rank 0's BatchRenorm saw one shard per step
    d = (r0['data'] - opt.flat[0].data.cpu()).abs()
    scale = float(opt.flat[0].data.abs().max())
    print(f'[2 ranks, one GPU, gloo] max |param diff| {float(d.max()):.2e} (max |param| {scale:.2f}); exposed all-reduce wait {r0["wait_ms"]} ms/step')
    assert float(d.max()) < 1e-5, float(d.max())                                # float-atomics order in the per-channel reductions only (measured 1.5e-8)
    for i in (1, 2):
        assert ref_losses[i][0] == pytest.approx(r0['losses'][i], rel=2e-3) and ref_losses[i][1] == pytest.approx(r1['losses'][i], rel=2e-3)
