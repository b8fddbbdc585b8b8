"""ORACLE — TEST INFRASTRUCTURE ONLY (development container only).

Imports the reference implementation from /root/reference (CPU, fp32), checks
the restatement in oracle/sconformer_ref.py against it, and writes the golden
fixtures under tests/golden/ that pin both the oracle and the HIP path.  This
script is inert on the GPU box (no /root/reference there); the fixtures it
writes are plain .npz data (inputs + expected outputs), loadable with
numpy.load(allow_pickle=False).

Loader shim: SURVEY.md appendix A.5 (skips the package __init__s that pull
torchaudio/librosa/jiwer, which are not needed on the hot path).

Usage:  python oracle/make_golden.py
"""
from __future__ import annotations

import os
import sys
import types
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = '/root/reference'
GOLD = os.path.join(ROOT, 'tests', 'golden')


def load_reference():
    sys.path.insert(0, REF)
    for name, path in [('lcasr', REF + '/lcasr'), ('lcasr.utils', REF + '/lcasr/utils'),
                       ('lcasr.models', REF + '/lcasr/models'), ('lcasr.optim', REF + '/lcasr/optim')]:
        m = types.ModuleType(name); m.__path__ = [path]; sys.modules[name] = m
    oc = types.ModuleType('omegaconf'); oc2 = types.ModuleType('omegaconf.omegaconf')
    oc2.OmegaConf = object; oc.omegaconf = oc2; oc.OmegaConf = object
    sys.modules['omegaconf'] = oc; sys.modules['omegaconf.omegaconf'] = oc2
    warnings.filterwarnings('ignore')
    from lcasr.models.sconformer_xl import SCConformerXL
    from lcasr.components.attention import attention_ref
    from lcasr.optim.madgrad import MADGRAD
    return SCConformerXL, attention_ref, MADGRAD


TINY = dict(vocab_size=127, n_layers=2, d_model=64, n_heads=2, head_dim=32, subsampling_conv_channels=32,
            use_rotary=True, rotary_base_freq=1500000, decoder_norm=True, self_conditioning=True, bias_in_ff=False)
C1 = dict(vocab_size=4095, n_layers=6, d_model=256, n_heads=8, head_dim=32, subsampling_conv_channels=256,
          use_rotary=True, rotary_base_freq=1500000, decoder_norm=True, self_conditioning=True, bias_in_ff=False,
          default_norm='layer_norm')
C2 = dict(C1, d_model=768, n_heads=6, head_dim=128)           # BASELINE configs[1]: 6L/768D/6H, seq=1024


def synth(B, T, V, lengths=None, seed=0):
    """SURVEY §8(d) synthetic inputs: mel ~ N(0,1) (B,80,T), targets uniform [0,V), S=N/4."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 80, T, generator=g)
    N = T // 8
    S = max(N // 4, 1)
    tg = torch.randint(0, V, (B, S), generator=g)
    ln = torch.tensor(lengths if lengths is not None else [T] * B)
    # per-sample target length: a quarter of that sample's token count (always CTC-feasible)
    tl = torch.tensor([max(1, min(S, ((int(l) - 1) // 8 + 1) // 4)) for l in ln], dtype=torch.long)
    return x, ln, tg, tl


def np_sd(sd):
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


GS_CAP = 8192          # 'gs.' entries: every gradient tensor, flattened and strided down to at most this many elements


def strided(v, cap=GS_CAP):
    flat = v.reshape(-1)
    step = max(1, -(-flat.numel() // cap))
    return flat[::step]


def perturb_brn(model, nbt=30000):
    """Move the BatchRenorm buffers off their init (running_mean 0, running_std 1, nbt 0 -> r = 1, d = 0 whatever the
    statistics) so that the r / d clamps are live: rmax(30000) = 2.43, dmax = 5."""
    g = torch.Generator().manual_seed(77)
    with torch.no_grad():
        for l in model.layers:
            bn = l.conv.fn.batch_norm
            bn.running_mean.copy_(0.05 * torch.randn(bn.running_mean.shape, generator=g))
            bn.running_std.copy_(0.15 + 0.3 * torch.rand(bn.running_std.shape, generator=g))
            bn.num_batches_tracked.fill_(nbt)


def run_case(SC, kw, B, T, lengths, tag, save_all_grads=True, save_weights=True, max_abs_tol=2e-5, prep=None, strided_grads=False):
    from oracle import sconformer_ref as O
    torch.manual_seed(12345)                                   # exp/train.py:363
    model = SC(**kw)
    model.train()
    if prep is not None:
        prep(model)
    V = kw['vocab_size']
    x, ln, tg, tl = synth(B, T, V, lengths)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}

    caps = {}
    hooks = []
    def hook(name):
        def f(mod, inp, out):
            caps[name] = (out[0] if isinstance(out, tuple) else out).detach().clone()
        return f
    hooks.append(model.subsampling.register_forward_hook(hook('sub.out')))
    for i, l in enumerate(model.layers):
        hooks.append(l.register_forward_hook(hook(f'layers.{i}.out')))
        hooks.append(l.attend.register_forward_hook(hook(f'layers.{i}.attend.branch')))
        hooks.append(l.conv.register_forward_hook(hook(f'layers.{i}.conv.branch')))
        hooks.append(l.ff1.register_forward_hook(hook(f'layers.{i}.ff1.branch')))

    out = model(x, length=ln)
    lp = out['final_posteriors']
    ctc = torch.nn.CTCLoss(blank=model.decoder.num_classes - 1, reduction='sum')   # exp/train.py:104
    loss = ctc(lp.transpose(0, 1), tg, out['length'], tl).sum()                     # exp/train.py:249
    scaled = loss / (T * B) * 100                                                   # exp/train.py:275
    lp.retain_grad()
    scaled.backward()
    for h in hooks: h.remove()
    ref_grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    sd1 = {k: v.clone() for k, v in model.state_dict().items()}

    # ---- oracle on the same weights -------------------------------------
    cfg = O.make_config(**kw)
    sdo = {k: v.clone().requires_grad_(v.is_floating_point() and k in ref_grads) for k, v in sd0.items()}
    nb, cap = {}, {}
    oloss, oscaled, oout = O.train_step_loss(sdo, cfg, x, ln, tg, tl, new_buffers=nb)
    oout['final_posteriors'].retain_grad()
    oscaled.backward()
    def md(a, b): return float((a - b).abs().max())
    errs = {'logp': md(oout['final_posteriors'], lp), 'loss': abs(float(oloss) - float(loss)) / abs(float(loss))}
    # dw-conv bias grads are analytically 0 (BatchRenorm removes the mean): compare on an absolute floor
    gmax = max(float(v.abs().max()) for v in ref_grads.values())
    zero_keys = [k for k in ref_grads if k.endswith('depthwise_conv.bias')]        # pure cancellation noise
    errs['grad'] = max(md(sdo[k].grad, ref_grads[k]) / max(float(ref_grads[k].abs().max()), 1e-3 * gmax)
                       for k in ref_grads if k not in zero_keys)
    errs['grad0'] = max(md(sdo[k].grad, ref_grads[k]) / gmax for k in zero_keys)
    errs['buf'] = max(md(nb[k].float(), sd1[k].float()) for k in nb)
    assert (oout['length'] == out['length']).all()
    print(f'[{tag}] loss={float(loss):.6f} oracle-vs-reference:', {k: f'{v:.2e}' for k, v in errs.items()})
    assert errs['logp'] < max_abs_tol and errs['loss'] < 1e-5 and errs['grad'] < 1e-3 and errs['grad0'] < 1e-4 and errs['buf'] < 1e-6, errs

    fx = dict(x=x.numpy(), lengths=ln.numpy(), targets=tg.numpy(), target_lengths=tl.numpy(),
              out_length=out['length'].numpy(), loss=np.float64(float(loss)), scaled_loss=np.float64(float(scaled)))
    for k, v in kw.items():
        fx['cfg.' + k] = np.array(v)
    if save_weights:
        for k, v in np_sd(sd0).items(): fx['w.' + k] = v
        fx['logp'] = lp.detach().numpy()
        fx['dlogp'] = lp.grad.detach().numpy()
        for k, v in caps.items(): fx['cap.' + k] = v.numpy()
        for k in nb: fx['buf.' + k] = sd1[k].numpy()
    else:
        fx['logp_slice'] = lp.detach()[:, ::17, ::97].numpy()
        fx['logp_sum'] = np.float64(float(lp.detach().double().sum()))
        fx['logp_abs_sum'] = np.float64(float(lp.detach().double().abs().sum()))
    if save_all_grads:
        for k, v in ref_grads.items(): fx['g.' + k] = v.numpy()
    else:
        for k, v in ref_grads.items(): fx['gnorm.' + k] = np.float64(float(v.double().norm()))
    if strided_grads:                                          # a strided sample of EVERY gradient tensor (relative-L2 parity)
        for k, v in ref_grads.items(): fx['gs.' + k] = strided(v).numpy().copy()
        fx['gs_cap'] = np.array(GS_CAP)
    np.savez_compressed(os.path.join(GOLD, tag + '.npz'), **fx)
    return model


def bf16_noise_case(SC):
    """What "bf16 tolerance" means for this model: the reference's OWN bf16-autocast path (exp/train.py:236, here
    torch.autocast('cpu', bfloat16) - the CPU-runnable form of the same autocast policy) against its fp32 path, on the cases
    the HIP path is checked on.  Stored: loss / log-prob differences and the per-tensor relative L2 of every gradient
    (tests/common_model.py::rel_l2_errors).  The HIP path must not be noisier than this."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from common_model import rel_l2_errors
    fx = {}
    cases = {'tiny_ln_ragged': (dict(TINY, default_norm='layer_norm'), 2, 256, [256, 200]),
             'tiny_ln_equal': (dict(TINY, default_norm='layer_norm'), 2, 256, None),
             'tiny_rms_ragged': (dict(TINY, default_norm='rms_norm'), 2, 256, [256, 200]),
             'tiny_ln_odd': (dict(TINY, default_norm='layer_norm'), 3, 1000, [1000, 1023 - 40, 17 * 8]),
             'c1_scalars': (C1, 2, 1024, None), 'c2_scalars': (C2, 2, 1024, None)}

    def run(kw, B, T, lengths, bf16):
        torch.manual_seed(12345)
        m = SC(**kw); m.train()
        x, ln, tg, tl = synth(B, T, kw['vocab_size'], lengths)
        with torch.autocast('cpu', dtype=torch.bfloat16, enabled=bf16):
            out = m(x, length=ln)
            lp = out['final_posteriors']
        loss = torch.nn.CTCLoss(blank=m.decoder.num_classes - 1, reduction='sum')(lp.float().transpose(0, 1), tg, out['length'], tl)
        (loss / (T * B) * 100).backward()
        return float(loss), lp.detach().float(), {k: p.grad.detach().float() for k, p in m.named_parameters()}

    for tag, (kw, B, T, ln) in cases.items():
        l0, lp0, g0 = run(kw, B, T, ln, False)
        l1, lp1, g1 = run(kw, B, T, ln, True)
        e = rel_l2_errors(g1, {k: v.numpy() for k, v in g0.items()})
        d = (lp1 - lp0).abs()
        names = sorted(e)
        fx[tag + '.names'] = np.array(names)
        fx[tag + '.grad_l2'] = np.array([e[k] for k in names])
        fx[tag + '.loss_rel'] = np.float64(abs(l1 - l0) / l0)
        fx[tag + '.logp_max'] = np.float64(float(d.max())); fx[tag + '.logp_mean'] = np.float64(float(d.mean()))
        live = [e[k] for k in names if not k.endswith('depthwise_conv.bias')]
        print(f'[bf16 noise of the reference, {tag}] loss rel {fx[tag + ".loss_rel"]:.2e}, log-prob max {float(d.max()):.3f} mean {float(d.mean()):.4f}, '
              f'gradient rel-L2 median {np.median(live):.4f} worst {max(live):.4f}')
    np.savez_compressed(os.path.join(GOLD, 'ref_bf16_noise.npz'), **fx)


def chunk_case():
    """f2: `chunk_spectogram` (lcasr/utils/dataloading.py:14-25) run on integer-valued spectrograms.  The module's top-level
    imports need torchaudio (absent here), so the function's own definition is taken out of the file's syntax tree and
    executed as is (it only touches torch); nothing of the file is written anywhere."""
    import ast
    src = open(REF + '/lcasr/utils/dataloading.py').read()
    node = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == 'chunk_spectogram')
    ns = {'torch': torch}
    exec(compile(ast.Module(body=[node], type_ignores=[]), REF + '/lcasr/utils/dataloading.py', 'exec'), ns)
    ref_fn = ns['chunk_spectogram']
    fx = {}
    cases = [(1000, 256, 64), (1000, 256, 0), (1000, 400, 8), (513, 512, 1), (77, 128, 32), (2048, 2048, 0), (300, 100, 99)]
    fx['cases'] = np.array(cases)
    for ci, (T, size, ov) in enumerate(cases):
        spec = torch.arange(2 * 3 * T, dtype=torch.float32).view(2, 3, T)
        chunks = ref_fn(spec, size, ov)
        fx[f'n.{ci}'] = np.array(len(chunks))
        fx[f'widths.{ci}'] = np.array([c.shape[-1] for c in chunks])
        fx[f'first.{ci}'] = np.array([float(c[0, 0, 0]) for c in chunks])          # = start frame of each chunk
        fx[f'sum.{ci}'] = np.array([float(c.double().sum()) for c in chunks])
    np.savez_compressed(os.path.join(GOLD, 'chunks.npz'), **fx)
    print('[chunks] wrote', len(cases), 'cases')


def attention_cases(attention_ref):
    """F3/F4: flash-attn semantics (incl. local window) from attention.py:330-410 — the only
    CPU-runnable definition of window_size in the reference."""
    from oracle import sconformer_ref as O
    g = torch.Generator().manual_seed(3)
    fx = {}
    for name, (B, N, H, D, win, lens) in {
        'full_d128': (1, 256, 2, 128, (-1, -1), None),
        'full_d32': (2, 192, 4, 32, (-1, -1), [192, 100]),
        'win_d32': (2, 160, 2, 32, (16, 16), None),
        'win_asym_d128': (1, 128, 2, 128, (24, 8), None),
    }.items():
        q, k, v = (torch.randn(B, N, H, D, generator=g) for _ in range(3))
        kpm = None
        if lens is not None:
            kpm = torch.arange(N)[None, :] < torch.tensor(lens)[:, None]
        o, _ = attention_ref(q, k, v, query_padding_mask=kpm, key_padding_mask=kpm, window_size=win, upcast=True)
        fx[name + '.q'], fx[name + '.k'], fx[name + '.v'], fx[name + '.o'] = q.numpy(), k.numpy(), v.numpy(), o.numpy()
        fx[name + '.window'] = np.array(win)
        fx[name + '.lens'] = np.array(lens if lens is not None else [N] * B)
    np.savez_compressed(os.path.join(GOLD, 'attention.npz'), **fx)
    print('[attention] wrote', len(fx), 'arrays')


def madgrad_case(MADGRAD):
    from oracle import madgrad_ref as M
    g = torch.Generator().manual_seed(5)
    shapes = [(37,), (8, 16), (3, 5, 7)]
    params = [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in shapes]
    opt = MADGRAD(params, lr=3e-3, momentum=0.9, weight_decay=0.0, eps=1e-6)
    fx = {}
    for i, p in enumerate(params): fx[f'p0.{i}'] = p.detach().numpy().copy()
    # oracle state
    st = [dict(p=p.detach().numpy().copy(), gss=np.zeros(p.shape, np.float32), s=np.zeros(p.shape, np.float32),
               x0=p.detach().numpy().copy()) for p in params]
    for step in range(4):
        grads = [torch.randn(*s, generator=g) * (10.0 if step == 1 else 0.1) for s in shapes]
        for i, gr in enumerate(grads): fx[f'g{step}.{i}'] = gr.numpy().copy()
        for p, gr in zip(params, grads): p.grad = gr.clone()
        torch.nn.utils.clip_grad_norm_(params, 0.8)             # exp/train.py:55
        opt.step()
        coef, _ = M.clip_coef([x.numpy() for x in grads], 0.8)
        for i, gr in enumerate(grads):
            s_ = st[i]
            s_['p'], s_['gss'], s_['s'] = M.madgrad_step(s_['p'], gr.numpy() * np.float32(coef), s_['gss'], s_['s'], s_['x0'], step, 3e-3)
            err = np.abs(s_['p'] - params[i].detach().numpy()).max()
            assert err < 1e-6, (step, i, err)
        for i, p in enumerate(params): fx[f'p{step + 1}.{i}'] = p.detach().numpy().copy()
    np.savez_compressed(os.path.join(GOLD, 'madgrad.npz'), **fx)
    print('[madgrad] oracle matches reference; wrote fixture')


def schedule_and_checkpoint_case(MADGRAD):
    """f2/f4: the reference's SequenceWarmupManager / CosineLRScheduler step sequences, chunk_spectogram, and the layout of
    MADGRAD.state_dict() after two steps (so a resumed optimiser can be checked against steps 3-4 of madgrad.npz)."""
    from lcasr.utils.scheduling import SequenceWarmupManager, CosineLRScheduler
    fx = {}
    cfgs = [dict(increase_every=5, stop_after=23, start_after=3, initial_sequence_length=512, initial_batch_size=64, max_sequence_length=6000),
            dict(increase_every=4, stop_after=100, start_after=0, initial_sequence_length=2048, initial_batch_size=3, max_sequence_length=16384,
                 increase_by_multiplier=1.5, batch_size_multiplier=0.7),
            dict(increase_every=-1, stop_after=10, start_after=0, initial_sequence_length=2048, initial_batch_size=8, max_sequence_length=4096)]
    for ci, c in enumerate(cfgs):
        m = SequenceWarmupManager(**c)
        rows = []
        for i in range(60):
            ch, sl, bs = m.step(steps=1 + (i % 3 == 0))
            rows.append([int(ch), sl, bs, m.cur_position, m.steps_since_last_increase])
        fx[f'swm.{ci}'] = np.array(rows, dtype=np.int64)
        fx[f'swm_cfg.{ci}'] = np.array([c.get(k, d) for k, d in (('increase_every', 0), ('stop_after', 0), ('start_after', 0), ('initial_sequence_length', 0),
                                       ('initial_batch_size', 0), ('max_sequence_length', 0), ('increase_by_multiplier', 2.0), ('batch_size_multiplier', 0.5))], dtype=np.float64)
    p = torch.nn.Parameter(torch.zeros(3))
    opt = torch.optim.SGD([p], lr=1.0)
    sch = CosineLRScheduler(opt, warmup_steps=7, peak_value=3e-3, final_value=0.0)
    lrs = []
    for i in range(12):
        opt.step(); sch.step(); lrs.append(sch.get_last_lr()[0])
    sch.set_cosine_schedule(total_recordings=40, cur_podcast=12)
    for i in range(30):
        opt.step(); sch.step(); lrs.append(sch.get_last_lr()[0])
    fx['cosine_lrs'] = np.array(lrs, dtype=np.float64)
    fx['cosine_state_keys'] = np.array(sorted(k for k in sch.state_dict().keys()))
    # optimiser state after 2 of the 4 steps of madgrad.npz
    mg = np.load(os.path.join(GOLD, 'madgrad.npz'))
    shapes = [(37,), (8, 16), (3, 5, 7)]
    params = [torch.nn.Parameter(torch.from_numpy(mg[f'p0.{i}'].copy())) for i in range(3)]
    opt = MADGRAD(params, lr=3e-3, momentum=0.9, weight_decay=0.0, eps=1e-6)
    for step in range(2):
        for i, q in enumerate(params): q.grad = torch.from_numpy(mg[f'g{step}.{i}'].copy())
        torch.nn.utils.clip_grad_norm_(params, 0.8)
        opt.step()
    sd = opt.state_dict()
    assert sorted(map(str, sd['state'].keys())) == ['0', '1', '2', 'k'], sd['state'].keys()
    for i in range(3):
        for name in ('grad_sum_sq', 's', 'x0'):
            fx[f'opt.state.{i}.{name}'] = sd['state'][i][name].numpy()
        assert sorted(sd['state'][i].keys()) == ['grad_sum_sq', 's', 'x0']
    fx['opt.k'] = sd['state']['k'].numpy()
    g0 = sd['param_groups'][0]
    fx['opt.group_keys'] = np.array(sorted(g0.keys()))
    fx['opt.group_params'] = np.array(g0['params'])
    np.savez_compressed(os.path.join(GOLD, 'schedules.npz'), **fx)
    print('[schedules/checkpoint] wrote fixture; optimizer group keys:', sorted(g0.keys()), 'scheduler keys:', sorted(sch.state_dict().keys()))


def infer_case(SC):
    """Sliding-window inference: the reference's own fetch_logits + GreedyCTCDecoder on a tiny eval model.
    lcasr.utils.audio_tools (torchaudio/librosa) is only used by unrelated helpers of that file: a two-function stand-in
    module lets lcasr/eval/utils.py import."""
    from oracle import sconformer_ref as O, infer_ref as I
    at = types.ModuleType('lcasr.utils.audio_tools'); at.total_frames = lambda s: int(s * 100); at.total_seconds = lambda f: f / 100
    sys.modules['lcasr.utils.audio_tools'] = at
    for name, path in [('lcasr.eval', REF + '/lcasr/eval'), ('lcasr.decoding', REF + '/lcasr/decoding')]:
        m = types.ModuleType(name); m.__path__ = [path]; sys.modules[name] = m
    from lcasr.eval.utils import fetch_logits
    from lcasr.decoding.greedy import GreedyCTCDecoder
    kw = dict(TINY, default_norm='layer_norm')
    torch.manual_seed(12345)
    model = SC(**kw)
    g = torch.Generator().manual_seed(11)
    model.train()
    with torch.no_grad():
        for _ in range(3):                                       # move the BatchRenorm running statistics off their init
            model(torch.randn(2, 80, 256, generator=g))
    model.eval(); model.device = 'cpu'
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    spec = torch.randn(1, 80, 1000, generator=g)

    class Tok:                                                   # fetch_logits only asks the tokenizer for its size
        def vocab_size(self): return kw['vocab_size']

    class Args: config = {'audio_chunking': {'size': 512, 'overlap': 128}}

    cfg = O.make_config(**kw)
    fx = dict(spec=spec.numpy())
    for k, v in kw.items(): fx['cfg.' + k] = np.array(v)
    for k, v in np_sd(sd).items(): fx['w.' + k] = v
    dec = GreedyCTCDecoder(tokenizer=None, blank_id=model.decoder.num_classes - 1)
    cases = [(256, 64), (256, 0), (2048, 0), (-1, -1), (320, 160), (1000, 0), (500, 0)]
    fx['cases'] = np.array(cases)
    for ci, (sl, ov) in enumerate(cases):
        ref = fetch_logits(Args, model, spec.clone(), sl, ov, Tok(), use_tqdm=False)
        sl_r, ov_r = (512, 128) if sl == -1 else (sl, ov)
        mine = I.fetch_logits(sd, cfg, spec, sl_r, ov_r, kw['vocab_size'])
        err = float(np.abs(ref - mine).max())
        ids = dec(torch.from_numpy(ref), decode=False)
        assert ids == I.greedy_decode(ref, model.decoder.num_classes - 1)
        print(f'[infer seq_len={sl} overlap={ov}] rows={ref.shape[0]} oracle-vs-reference max|d|={err:.2e} greedy tokens={len(ids)}')
        assert ref.shape == mine.shape and err < 2e-5, err
        fx[f'logits.{ci}'] = ref
        fx[f'greedy.{ci}'] = np.array(ids, dtype=np.int64)
    np.savez_compressed(os.path.join(GOLD, 'infer_tiny.npz'), **fx)


def main():
    if len(sys.argv) > 1 and sys.argv[1] in ('infer', 'sched'):  # regenerate only one of the later fixtures
        SC, _, MADGRAD = load_reference()
        torch.set_num_threads(8)
        return infer_case(SC) if sys.argv[1] == 'infer' else schedule_and_checkpoint_case(MADGRAD)
    if len(sys.argv) > 1 and sys.argv[1] == 'r2':                # the fixtures added in round 2 only
        SC, _, _ = load_reference()
        torch.set_num_threads(8)
        run_case(SC, C1, 2, 1024, None, 'c1_scalars', save_all_grads=False, save_weights=False, max_abs_tol=2e-4, strided_grads=True)
        run_case(SC, C2, 2, 1024, None, 'c2_scalars', save_all_grads=False, save_weights=False, max_abs_tol=5e-4, strided_grads=True)
        run_case(SC, dict(TINY, default_norm='layer_norm', checkpoint_every_n_layers=1, ff_checkpoint_lvl=2), 2, 256, [256, 200],
                 'tiny_ln_ckpt', prep=perturb_brn)
        run_case(SC, dict(TINY, default_norm='layer_norm'), 2, 256, [256, 200], 'tiny_ln_brn', prep=perturb_brn)
        bf16_noise_case(SC)
        return chunk_case()
    if len(sys.argv) > 1 and sys.argv[1] == 'noise':
        SC, _, _ = load_reference()
        torch.set_num_threads(8)
        return bf16_noise_case(SC)
    assert os.path.isdir(REF), 'reference not present: this script only runs in the development container'
    os.makedirs(GOLD, exist_ok=True)
    SC, attention_ref, MADGRAD = load_reference()
    torch.set_num_threads(8)
    run_case(SC, dict(TINY, default_norm='layer_norm'), 2, 256, [256, 200], 'tiny_ln_ragged')
    run_case(SC, dict(TINY, default_norm='layer_norm'), 2, 256, None, 'tiny_ln_equal')
    run_case(SC, dict(TINY, default_norm='rms_norm'), 2, 256, [256, 200], 'tiny_rms_ragged')
    run_case(SC, dict(TINY, default_norm='layer_norm'), 3, 1000, [1000, 1023 - 40, 17 * 8], 'tiny_ln_odd')
    run_case(SC, C1, 2, 1024, None, 'c1_scalars', save_all_grads=False, save_weights=False, max_abs_tol=2e-4, strided_grads=True)
    run_case(SC, C2, 2, 1024, None, 'c2_scalars', save_all_grads=False, save_weights=False, max_abs_tol=5e-4, strided_grads=True)
    # BASELINE config 4's switches (exp_set_seq_rotary_base_9l.yaml:52-53) on the tiny model, BatchRenorm clamps live
    run_case(SC, dict(TINY, default_norm='layer_norm', checkpoint_every_n_layers=1, ff_checkpoint_lvl=2), 2, 256, [256, 200],
             'tiny_ln_ckpt', prep=perturb_brn)
    run_case(SC, dict(TINY, default_norm='layer_norm'), 2, 256, [256, 200], 'tiny_ln_brn', prep=perturb_brn)
    bf16_noise_case(SC)
    chunk_case()
    attention_cases(attention_ref)
    madgrad_case(MADGRAD)
    infer_case(SC)
    schedule_and_checkpoint_case(MADGRAD)


if __name__ == '__main__':
    main()
