#!/usr/bin/env python
"""Headline benchmark: SConformerXL training step (forward + CTC + backward + clip + MADGRAD [+ RCCL gradient
all-reduce]) on synthetic mel, BASELINE.json config 3: 6L/768D/6H rotary theta=1.5M, seq=16384, bf16 compute.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line (rank 0): metric = spectrogram-frames/sec (whole job), plus
  roofline     : the dominant kernel (NT bf16 MFMA GEMM) timed live with HIP events on its stream over the timed region,
                 algorithmic FLOPs = 2*M*N*K per launch, peak = 2.5 PFLOP/s dense bf16;
                 roofline.attention = the attention forward / backward launches of the same timed region against the same peak
                 (algorithmic 4 N^2 H D per sample forward, 10 N^2 H D backward), roofline.hbm = an HBM-bound kernel of the same
                 region (the pre-norm forward, f32 rows in / bf16 rows out) in GB/s against the 8 TB/s peak;
  cpu_baseline : the CPU oracle (oracle/sconformer_ref.py, fp32, all host cores) on a bounded sample of the same
                 workload (N=1, rank 0 only); cpu_baseline_t1024: the same for the 1024-frame context;
  sweep        : (N=1, after the timed region of the headline) the other north-star contexts through the same code, each at its
                 configured batch: c2 (T=1024), c4 (9L, T=16384, per-layer checkpointing), c5 (3L/2048D, T=131072);
  dist         : (N>1) backend, gradient bytes all-reduced per step and rank, and how long the step's stream waited for the
                 all-reduces it could not hide behind the backward.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # BASELINE.json configs[2]: the configuration the metric is quoted on
    'c3': dict(model=dict(vocab_size=4095, n_layers=6, d_model=768, n_heads=6, head_dim=128, subsampling_conv_channels=256,
                          use_rotary=True, rotary_base_freq=1500000, decoder_norm=True, self_conditioning=True,
                          default_norm='layer_norm', bias_in_ff=False), T=16384, batch=128,
               # per-GPU batch: the reference trains this length with 22 recordings per 80 GB GPU (exp/configs/16_ds.yaml:92,
               # constant 360 k frames per batch) = 79 per 288 GB; the batch sizes that give every GEMM whole rounds of 256x256
               # tiles are multiples of 64, and 128 (131 GiB peak) is the faster of the two that fit comfortably: frames/s at
               # B = 16 / 32 / 64 / 96 / 128 on the v11 kernels ~ 5.3 / 5.6 / 5.96 / 5.7 / 6.08 M.  At 128 the subsampler's stage-1
               # tensors pass 2^31 elements; tests/test_model_gpu.py::test_subsampler_over_2pow31_elements_matches_two_halves
               # pins the 64-bit offsets.
               name='6L/768D/6H SConformerXL, seq=16384, rotary theta=1.5M'),
    'c2': dict(model=dict(vocab_size=4095, n_layers=6, d_model=768, n_heads=6, head_dim=128, subsampling_conv_channels=256,
                          use_rotary=True, rotary_base_freq=1500000, decoder_norm=True, self_conditioning=True,
                          default_norm='layer_norm', bias_in_ff=False), T=1024, batch=1024,
               # the reference trains 1024-frame chunks 352 to an 80 GB GPU (exp/configs/README.md:85-93) = 1267 to 288 GB; 1024 keeps
               # every GEMM in whole rounds of 256-row tiles (66 GiB peak).  (B = 64: 16 ms/step, 4.1 M frames/s - launch-bound.)
               name='6L/768D/6H SConformerXL, seq=1024'),
    'c1': dict(model=dict(vocab_size=4095, n_layers=6, d_model=256, n_heads=8, head_dim=32, subsampling_conv_channels=256,
                          use_rotary=True, rotary_base_freq=1500000, decoder_norm=True, self_conditioning=True,
                          default_norm='layer_norm', bias_in_ff=False), T=1024, batch=2,
               name='6L/256D/8H SConformerXL, seq=1024'),
    # BASELINE.json configs[3]: exp/configs/paper_templates/exp_set_seq_rotary_base_9l.yaml:27-54 (per-layer activation
    # checkpointing + ff_checkpoint_lvl 2)
    'c4': dict(model=dict(vocab_size=4095, n_layers=9, d_model=768, n_heads=6, head_dim=128, subsampling_conv_channels=256,
                          use_rotary=True, rotary_base_freq=1500000, decoder_norm=True, self_conditioning=True,
                          default_norm='layer_norm', bias_in_ff=False, checkpoint_every_n_layers=1, ff_checkpoint_lvl=2),
               T=16384, batch=128, name='9L/768D/6H SConformerXL, seq=16384, per-layer checkpointing'),     # 3.73 M frames/s (batch 64: 3.65 M)
    # BASELINE.json configs[4]: exp_set_seq_rotary_base_3l_2048.yaml:27-53 (20-minute context)
    'c5': dict(model=dict(vocab_size=4095, n_layers=3, d_model=2048, n_heads=16, head_dim=128, subsampling_conv_channels=512,
                          use_rotary=True, rotary_base_freq=1500000, decoder_norm=True, self_conditioning=True,
                          default_norm='layer_norm', bias_in_ff=False, ff_checkpoint_lvl=2),
               T=131072, batch=16,      # 16 recordings of 20 minutes: 157 GiB peak; 1.87 M frames/s against 1.80 M at batch 8 (the lattice kernel
               # of the CTC loss runs 2 workgroups per sample)
               name='3L/2048D/16H SConformerXL, seq=131072'),
}
PEAK_BF16_DENSE = 2.5e15        # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16 MFMA
PEAK_HBM = 8.0e12               # MI355X_MICROARCH.md: 8.0 TB/s HBM3E (spec; 6.3 TB/s is what a copy reaches)


GEMM_KERNELS = {0: 'gemm_kernel<NT> (128x128 tile)',
                1: 'gemm256_kernel<false, 2, *> (NT, 256x256 tile; one instantiation per epilogue kind - the family row of the profile)',
                2: 'gemm192_kernel<*> (NT, 256x192 tile, 3 phases per K-tile)', 3: 'gemm256_kernel<true, 2> (TN, 256x256 tile)'}


class GemmTimer:
    """HIP-event timing, on the stream it is launched on, of every launch of the dominant GEMM kernel: the variant
    (sconf_gemm_variant tells which kernel a problem runs on) that carries the most FLOPs in a warm-up step.
    Events are created and recorded once BEFORE the timed region (event creation grows a driver pool and stalls the
    stream for tens of ms when it happens mid-run); inside the timed region events are only re-recorded."""

    def __init__(self):
        self.pool, self.used, self.enabled, self.count_only, self.calls = [], 0, False, False, 0
        self.log = []                                                 # (variant, flops) per timed launch
        self.warm_flops = {}                                          # variant -> flops of one warm-up step
        self.only = None                                              # the variant timed in the timed region
        self.xpool, self.xused, self.xlog = [], 0, []                 # attention / norm launches

    def install(self):
        import lcasr_amd.hip.ops as ops
        from lcasr_amd.hip import _lib
        inner = ops.gemm
        timer = self
        acts = ops.ACT

        def gemm(a, b, layout='nt', **kw):
            if not (timer.enabled or timer.count_only):
                return inner(a, b, layout, **kw)
            if layout == 'nt': (m, k), n = a.shape, b.shape[0]
            elif layout == 'nn': (m, k), n = a.shape, b.shape[1]
            else: (k, m), n = a.shape, b.shape[1]
            var = _lib.load().sconf_gemm_variant(ops.LAYOUT[layout], m, n, k, a.stride(0), b.stride(0), int(kw.get('split_k', 1)),
                                                 acts[kw.get('act', 'none')], int(kw.get('resid') is not None or kw.get('accum') is not None and kw.get('split_k', 1) == 1),
                                                 int(bool(kw.get('save_pre'))))
            if layout == 'nn': var = 0
            if timer.count_only:                                      # warm-up step: which kernel carries the most FLOPs?
                timer.warm_flops[var] = timer.warm_flops.get(var, 0.0) + 2.0 * m * n * k
                timer.calls += 1
                return inner(a, b, layout, **kw)
            if var != timer.only or timer.used + 2 > len(timer.pool):
                return inner(a, b, layout, **kw)
            e0, e1 = timer.pool[timer.used], timer.pool[timer.used + 1]
            timer.used += 2
            e0.record()
            out = inner(a, b, layout, **kw)
            e1.record()
            nbytes = 2.0 * (m * k + n * k) + m * n * (4 if kw.get('out_dtype') == torch.float32 or kw.get('accum') is not None else 2)
            nbytes += m * n * (2 * bool(kw.get('save_pre')) + 4 * (kw.get('resid') is not None) + 2 * (kw.get('aux') is not None))
            timer.log.append((var, 2.0 * m * n * k, nbytes))
            return out
        ops.gemm = gemm
        inner_sb = ops.gemm_softmax_bwd

        def gemm_softmax_bwd(dy, wt, probs, delta, colsum_into=None):     # a 256x256 NT launch too (softmax backward in its epilogue)
            if not (timer.enabled or timer.count_only): return inner_sb(dy, wt, probs, delta, colsum_into)
            (m, k), n = dy.shape, wt.shape[0]
            if timer.count_only:
                timer.warm_flops[1] = timer.warm_flops.get(1, 0.0) + 2.0 * m * n * k
                timer.calls += 1
                return inner_sb(dy, wt, probs, delta, colsum_into)
            if timer.only != 1 or timer.used + 2 > len(timer.pool): return inner_sb(dy, wt, probs, delta, colsum_into)
            e0, e1 = timer.pool[timer.used], timer.pool[timer.used + 1]
            timer.used += 2
            slab = torch.empty(2 * (m // 256), n, dtype=torch.float32, device=dy.device)      # (the wrapper's own allocation, outside the events)
            dl = torch.empty(m, n, dtype=torch.bfloat16, device=dy.device)
            e0.record()
            _lib.call('sconf_gemm_softmax_bwd', dy.data_ptr(), wt.data_ptr(), probs.data_ptr(), delta.data_ptr(), dl.data_ptr(), slab.data_ptr(),
                      m, n, k, dy.stride(0), wt.stride(0), probs.stride(0), torch.cuda.current_stream().cuda_stream)
            e1.record()
            if colsum_into is not None: ops.colsum_(slab, colsum_into)
            timer.log.append((1, 2.0 * m * n * k, 2.0 * (m * k + n * k) + 4.0 * m * n + 4.0 * m + 4.0 * slab.numel()))
            return dl
        ops.gemm_softmax_bwd = gemm_softmax_bwd
        # attention forward / backward and one HBM-bound kernel (the pre-norm forward), timed the same way: ~30 more event
        # pairs per step
        inner_af, inner_ab, inner_nf = ops.attn_fwd, ops.attn_bwd, ops.norm_fwd

        def attn_fwd(q, k, v, *a, **kw):
            if not timer.enabled: return inner_af(q, k, v, *a, **kw)
            B_, N_, H_, D_ = q.shape
            return timer._timed('attn_fwd', 4.0 * N_ * N_ * H_ * D_ * B_, 0.0, lambda: inner_af(q, k, v, *a, **kw))

        def attn_bwd(q, k, v, *a, **kw):
            if not timer.enabled: return inner_ab(q, k, v, *a, **kw)
            B_, N_, H_, D_ = q.shape
            return timer._timed('attn_bwd', 10.0 * N_ * N_ * H_ * D_ * B_, 0.0, lambda: inner_ab(q, k, v, *a, **kw))

        def norm_fwd(x, w, b, mode, eps, out_dtype):
            if not timer.enabled or x.dtype != torch.float32 or out_dtype != torch.bfloat16: return inner_nf(x, w, b, mode, eps, out_dtype)
            return timer._timed('norm_fwd', 0.0, x.numel() * 6.0, lambda: inner_nf(x, w, b, mode, eps, out_dtype))
        ops.attn_fwd, ops.attn_bwd, ops.norm_fwd = attn_fwd, attn_bwd, norm_fwd

    def _timed(self, tag, flops, nbytes, fn):
        if self.xused + 2 > len(self.xpool):
            return fn()
        e0, e1 = self.xpool[self.xused], self.xpool[self.xused + 1]
        self.xused += 2
        e0.record(); out = fn(); e1.record()
        self.xlog.append((tag, flops, nbytes))
        return out

    def extra_summary(self):
        per = {}
        for i, (tag, fl, nb) in enumerate(self.xlog):
            ms = self.xpool[2 * i].elapsed_time(self.xpool[2 * i + 1])
            d = per.setdefault(tag, [0, 0.0, 0.0, 0.0]); d[0] += 1; d[1] += ms; d[2] += fl; d[3] += nb
        att, hbm = {}, None
        for tag, key in (('attn_fwd', 'fwd'), ('attn_bwd', 'bwd')):
            if tag in per:
                n, ms, fl, _ = per[tag]
                att[key] = dict(achieved=round(fl / (ms * 1e-3) / 1e12, 1), unit='TFLOP/s', frac=round(fl / (ms * 1e-3) / PEAK_BF16_DENSE, 4),
                                launches=n, avg_launch_us=round(ms * 1e3 / n, 1))
        if att:
            att['peak'] = PEAK_BF16_DENSE / 1e12
            att['flops'] = 'algorithmic: forward 4 N^2 H D per sample, backward 10 N^2 H D (dQ + dK/dV kernels together; they execute 14)'
        if 'norm_fwd' in per:
            n, ms, _, nb = per['norm_fwd']
            hbm = dict(bound='hbm', kernel='norm_fwd_kernel (LayerNorm, f32 rows in, bf16 rows out)', achieved=round(nb / (ms * 1e-3) / 1e9, 1),
                       peak=PEAK_HBM / 1e9, unit='GB/s', frac=round(nb / (ms * 1e-3) / PEAK_HBM, 4), launches=n, avg_launch_us=round(ms * 1e3 / n, 1),
                       algorithmic_bytes_per_launch=int(nb / n))
        return att or None, hbm

    def prepare(self, n_launches):
        # events only around the dominant kernel's launches: ~100 event records per step instead of ~350, which would
        # themselves cost about 1 ms per step
        self.only = min(self.warm_flops, key=lambda v: (-self.warm_flops[v], v)) if self.warm_flops else None
        self.pool = [torch.cuda.Event(enable_timing=True) for _ in range(2 * n_launches)]
        self.xpool = [torch.cuda.Event(enable_timing=True) for _ in range(2 * 64 * max(1, n_launches // max(self.calls, 1)))]
        for e in self.pool + self.xpool:
            e.record()                                                # force lazy creation now
        torch.cuda.synchronize()

    def summary(self):
        if not self.used:
            return None
        per = {}
        for i, (var, fl, nb) in enumerate(self.log):
            ms = self.pool[2 * i].elapsed_time(self.pool[2 * i + 1])
            d = per.setdefault(var, [0, 0.0, 0.0, 0.0]); d[0] += 1; d[1] += ms; d[2] += fl; d[3] += nb
        top = max(per, key=lambda v: per[v][1])
        n, ms, fl, nb = per[top]
        ach = fl / (ms * 1e-3)
        tot = sum(self.warm_flops.values()) or 1.0
        others = {GEMM_KERNELS.get(v, str(v)): round(f / tot, 3) for v, f in sorted(self.warm_flops.items())}
        # HBM bytes per launch of this kernel from the PMC passes kept under profiles/ (FETCH_SIZE / WRITE_SIZE with the gfx950
        # correction, see the file's header); the operand / output bytes of the same launches are computed here
        traffic, src = None, None
        try:
            tname = next(n for n in ('r03_hbm_traffic.json', 'r02_hbm_traffic.json') if os.path.exists(os.path.join(ROOT, 'profiles', n)))
            tj = json.load(open(os.path.join(ROOT, 'profiles', tname)))
            key = GEMM_KERNELS.get(top, '').split(' (')[0]                      # family rows are written by tools/hbm_traffic.py
            if key in tj['kernels']:
                traffic, src = tj['kernels'][key]['hbm_bytes_per_launch'], f'profiles/{tname} (PMC, same batch)'
        except Exception:
            pass
        return dict(bound='mfma', kernel=GEMM_KERNELS.get(top, str(top)), achieved=round(ach / 1e12, 2), peak=PEAK_BF16_DENSE / 1e12,
                    unit='TFLOP/s', frac=round(ach / PEAK_BF16_DENSE, 4), traffic=traffic, traffic_unit='bytes per launch',
                    traffic_source=src, algorithmic_bytes_per_launch=int(nb / n), launches=n,
                    avg_launch_us=round(ms * 1e3 / n, 2), gemm_flop_share_by_kernel=others)


def host_cores() -> int:
    """CPU threads this process may really use: cgroup quota if set, else the affinity mask, capped at the one-GPU
    box share (16) so the oracle is not oversubscribed on a 256-thread host it only owns a slice of."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get('SCONF_CPU_THREADS', '16'))))


def cpu_baseline(cfg_name: str, budget_s: float = 10.0, parity: bool = True):
    """Oracle (CPU fp32 restatement of the reference path) on a bounded sample: B=1 of the same config, forward+CTC+backward
    steps for ~budget_s seconds after one warm-up, all host cores.  parity: also run the HIP path on that very sample."""
    from oracle import sconformer_ref as O
    from lcasr_amd.models.sconformer_xl import SCConformerXL
    cfg = CONFIGS[cfg_name]
    cores = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(12345)
    sd = {k: v.clone() for k, v in SCConformerXL(**cfg['model']).state_dict().items()}
    gk = {k for k, v in sd.items() if v.is_floating_point() and 'running' not in k and 'rotary' not in k}
    sd = {k: (v.requires_grad_(True) if k in gk else v) for k, v in sd.items()}
    T = cfg['T']
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 80, T, generator=g)
    N = T // 8
    tg = torch.randint(0, cfg['model']['vocab_size'], (1, max(N // 4, 1)), generator=g)
    ocfg = O.make_config(**cfg['model'])

    def one_step():
        for v in sd.values():
            if v.requires_grad: v.grad = None
        loss, scaled, _ = O.train_step_loss(sd, ocfg, x, torch.tensor([T]), tg, torch.tensor([tg.shape[1]]))
        scaled.backward()
        return float(loss)

    one_step()                                                   # warm-up (allocator, thread pool)
    t0 = time.perf_counter()
    n = 0
    while True:
        loss = one_step(); n += 1
        if time.perf_counter() - t0 > budget_s or n >= 20: break # bounded: ~10-15 s of CPU work
    dt = (time.perf_counter() - t0) / n
    res = dict(value=round(T / dt, 1), unit='spectrogram-frames/sec', cores=cores, kind='port',
               sample=f'oracle fp32 forward+CTC+backward, B=1 x T={T}, {n} timed steps after 1 warm-up ({dt:.2f} s/step)', loss=round(loss, 3))
    if not parity:
        return res
    # CTC-loss parity at the benchmark size (BASELINE.json metric: "+ CTC-loss parity vs CPU ref"): the HIP path on the very
    # sample the oracle just ran (same seeded weights, same mel, same targets)
    from lcasr_amd.losses import CTCLoss
    torch.manual_seed(12345)
    m = SCConformerXL(**cfg['model']).cuda().train()
    tl = torch.tensor([tg.shape[1]]).cuda()
    with torch.no_grad():
        # the operator the timed step runs (head + log_softmax + CTC in one) ...
        hip_loss = float(m(x.cuda(), length=torch.tensor([T]).cuda(), ctc_targets=(tg.cuda(), tl))['ctc_nll'].sum())
    torch.manual_seed(12345)
    m = SCConformerXL(**cfg['model']).cuda().train()                 # (fresh BatchRenorm buffers: the forward above moved them)
    with torch.no_grad():
        # ... and the reference's two calls (posteriors, then CTCLoss)
        out = m(x.cuda(), length=torch.tensor([T]).cuda())
        hip_loss2 = float(CTCLoss(blank=m.decoder.num_classes - 1, reduction='sum')(out['final_posteriors'].transpose(0, 1), tg.cuda(), out['length'], tl))
    res['hip_loss_same_sample'] = round(hip_loss, 3)
    res['hip_loss_two_call_path'] = round(hip_loss2, 3)
    res['ctc_loss_rel_err'] = float(f'{max(abs(hip_loss - loss), abs(hip_loss2 - loss)) / abs(loss):.3e}')
    return res


def Fn_clear():
    """Release everything the previous model held on the device (bf16 weight shadows are keyed on the parameters)."""
    import gc
    import lcasr_amd.functional as Fn
    Fn.clear_weight_cache()
    gc.collect()
    torch.cuda.empty_cache()


def run_config(name: str, steps: int, warmup: int, batch: int = 0) -> dict:
    """One BASELINE config through Trainer.step on synthetic data: ms per step and frames / s (single GPU, no event timers)."""
    from lcasr_amd.models.sconformer_xl import SCConformerXL
    from lcasr_amd.train import Trainer, synthetic_batch
    cfg = CONFIGS[name]
    B, T = batch or cfg['batch'], cfg['T']
    torch.manual_seed(12345)
    model = SCConformerXL(**cfg['model']).cuda().train()
    trainer = Trainer(model, lr=3e-3, clip_value=0.8, global_batch=B)
    audio, lengths, targets, tl = synthetic_batch(B, T, cfg['model']['vocab_size'], seed=0)
    loss = None
    for _ in range(warmup):
        loss = trainer.step(audio, lengths, targets, tl)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = trainer.step(audio, lengths, targets, tl)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out = dict(workload=cfg['name'], seq_len=T, per_gpu_batch=B, steps=steps, warmup=warmup, ms_per_step=round(dt * 1e3, 2),
               frames_per_sec=round(B * T / dt, 1), loss_last_step=round(float(loss), 3),
               peak_mem_gib=round(torch.cuda.max_memory_allocated() / 2**30, 1))
    del trainer, model, audio
    Fn_clear()
    torch.cuda.reset_peak_memory_stats()
    return out


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` (N > 1) without a launcher: start the N ranks as a CHILD `torch.distributed.run` - before this
    process has touched the GPU, and never by exec - relay its output (rank 0 prints the JSON line) and return its exit code."""
    import socket
    import subprocess
    have = torch.cuda.device_count()                                  # does not initialise the GPU on this image
    if have < n and not os.environ.get('SCONF_SINGLE_DEVICE'):     # (the rehearsal hook puts every rank on cuda:0)
        print(f'bench.py: --gpus {n} but only {have} GPU(s) are visible; refusing to run fewer ranks than asked for', file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', default='c3', choices=list(CONFIGS))
    ap.add_argument('--batch', type=int, default=0, help='per-GPU batch (default: config value)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-sweep', action='store_true', help='skip the c2 / c4 / c5 lines after the headline (N=1, c3 only)')
    ap.add_argument('--fwd-only', action='store_true', help='diagnostic (SURVEY 8d "also report fwd-only"): time the training-mode forward alone')
    ap.add_argument('--per-step', action='store_true', help='diagnostic: print per-step GPU times (HIP events, no extra syncs)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    import torch.distributed as dist
    # rehearsal hook: SCONF_DIST_BACKEND=gloo SCONF_SINGLE_DEVICE=1 runs several ranks on ONE GPU (RCCL refuses duplicate
    # devices) to exercise the multi-rank code path on a one-GPU box; the driver's real runs use nccl (= RCCL over xGMI).
    backend = os.environ.get('SCONF_DIST_BACKEND', 'nccl')
    if os.environ.get('SCONF_SINGLE_DEVICE'):
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'

    import lcasr_amd  # noqa: F401
    from lcasr_amd.models.sconformer_xl import SCConformerXL
    from lcasr_amd.parallel import broadcast_module_state
    from lcasr_amd.train import Trainer, synthetic_batch

    cfg = CONFIGS[args.config]
    B = args.batch or cfg['batch']
    T = cfg['T']
    torch.manual_seed(12345)                                                  # exp/train.py:363
    model = SCConformerXL(**cfg['model']).cuda().train()
    if world > 1:
        broadcast_module_state(model)
    trainer = Trainer(model, lr=3e-3, clip_value=0.8, global_batch=B * world)
    trainer.sync.profile = world > 1                                          # HIP events around the all-reduce waits (dist.exposed_wait_ms_per_step)
    audio, lengths, targets, tl = synthetic_batch(B, T, cfg['model']['vocab_size'], seed=rank)

    timer = GemmTimer()
    timer.install()
    run_step = trainer.step
    if args.fwd_only:
        def run_step(audio, lengths, targets, tl):
            with torch.no_grad():
                return model(audio, length=lengths)['final_posteriors'][0, 0, 0]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    loss = None
    timer.count_only = True
    for i in range(max(args.warmup, 1)):
        loss = run_step(audio, lengths, targets, tl)
        if i == 0:
            timer.count_only = False
    timer.prepare(timer.calls * args.steps)
    import gc
    gc.collect(); gc.disable()                                       # no collector pauses inside the timed region
    sync()
    timer.enabled = True
    t0 = time.perf_counter()
    step_ev = []
    for _ in range(args.steps):
        if args.per_step:
            e = torch.cuda.Event(enable_timing=True); e.record(); step_ev.append(e)
        loss = run_step(audio, lengths, targets, tl)
    if args.per_step:
        e = torch.cuda.Event(enable_timing=True); e.record(); step_ev.append(e)
    sync()
    dt = time.perf_counter() - t0
    if args.per_step and rank == 0:
        print('per-step ms:', [round(a.elapsed_time(b), 1) for a, b in zip(step_ev[:-1], step_ev[1:])], file=sys.stderr)
    timer.enabled = False
    gc.enable()
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device='cuda')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)

    if rank == 0:
        frames = B * T * world * args.steps
        att, hbm = timer.extra_summary()
        roof = timer.summary()
        if roof is not None:
            roof['attention'], roof['hbm'] = att, hbm
        res = {
            'metric': 'spectrogram-frames/sec (6L/768D, seq=16384)' if args.config == 'c3' else f'spectrogram-frames/sec ({args.config}: {cfg["name"]})',
            'value': round(frames / dt, 1), 'unit': 'spectrogram-frames/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 2), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': cfg['name'] + (', forward only (train mode, no_grad)' if args.fwd_only else ', fwd+CTC+bwd+clip+MADGRAD'), 'per_gpu_batch': B, 'global_batch': B * world, 'seq_len': T,
                       'parallelism': f'dp{world}', 'loss_last_step': round(float(loss), 3)},
            'per_gpu_value': round(frames / dt / world, 1),
            'roofline': roof,
        }
        if world > 1:
            # what an N-GPU deviation from linear is made of: the exchange is one all-reduce(SUM) per 64 MiB slice of the flat f32
            # gradient buffer, issued from the backward as soon as a slice is final; `exposed_wait_ms_per_step` is how long the
            # compute stream then still had to wait in GradSync.finish() (HIP events on that stream, this rank)
            sy = trainer.sync
            res['dist'] = {'backend': backend + (' (RCCL over xGMI)' if backend == 'nccl' else ''), 'world_size': dist.get_world_size(),
                           'allreduce_bytes_per_step_per_rank': int(sy.flat_grad.numel() * 4), 'buckets': len(sy.buckets),
                           'exposed_wait_ms_per_step': sy.exposed_wait_ms(args.steps), 'single_device_rehearsal': bool(os.environ.get('SCONF_SINGLE_DEVICE'))}
        if world == 1 and not args.no_cpu_baseline and args.config == 'c5':
            # the oracle writes attention out as an explicit (H, N, N) f32 matrix: 17 GB per layer at N = 16384, not a bounded sample
            res['cpu_baseline'] = None
        elif world == 1 and not args.no_cpu_baseline:
            del trainer, model, audio
            Fn_clear()
            res['cpu_baseline'] = cpu_baseline(args.config)
            res['ctc_loss_rel_err'] = res['cpu_baseline']['ctc_loss_rel_err']
            res['gpu_over_cpu'] = round(res['value'] / res['cpu_baseline']['value'], 1)
            if args.config == 'c3':
                res['cpu_baseline_t1024'] = cpu_baseline('c2', budget_s=4.0, parity=False)
        if world == 1 and args.config == 'c3' and not args.no_sweep and not args.fwd_only:
            # the other north-star contexts (1024 and 131072 frames; the 9-layer model) through the same code path, after - and
            # outside - the timed region of the headline
            trainer = model = audio = None
            Fn_clear()
            res['sweep'] = {name: run_config(name, steps=5, warmup=2) for name in ('c2', 'c4', 'c5')}
            cb = res.get('cpu_baseline_t1024')
            if cb: res['sweep']['c2']['gpu_over_cpu'] = round(res['sweep']['c2']['frames_per_sec'] / cb['value'], 1)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
