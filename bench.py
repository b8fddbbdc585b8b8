#!/usr/bin/env python
"""Headline benchmark: SConformerXL training step (forward + CTC + backward + clip + MADGRAD [+ RCCL gradient
all-reduce]) on synthetic mel, BASELINE.json config 3: 6L/768D/6H rotary theta=1.5M, seq=16384, bf16 compute.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line (rank 0): metric = spectrogram-frames/sec (whole job), plus
  roofline     : the dominant kernel (NT bf16 MFMA GEMM) timed live with HIP events on its stream over the timed region,
                 algorithmic FLOPs = 2*M*N*K per launch, peak = 2.5 PFLOP/s dense bf16;
  cpu_baseline : the CPU oracle (oracle/sconformer_ref.py, fp32, all host cores) on a bounded sample of the same
                 workload (N=1, rank 0 only).
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
                          default_norm='layer_norm', bias_in_ff=False), T=1024, batch=64,
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
               T=16384, batch=64, name='9L/768D/6H SConformerXL, seq=16384, per-layer checkpointing'),
    # BASELINE.json configs[4]: exp_set_seq_rotary_base_3l_2048.yaml:27-53 (20-minute context)
    'c5': dict(model=dict(vocab_size=4095, n_layers=3, d_model=2048, n_heads=16, head_dim=128, subsampling_conv_channels=512,
                          use_rotary=True, rotary_base_freq=1500000, decoder_norm=True, self_conditioning=True,
                          default_norm='layer_norm', bias_in_ff=False, ff_checkpoint_lvl=2),
               T=131072, batch=8, name='3L/2048D/16H SConformerXL, seq=131072'),
}
PEAK_BF16_DENSE = 2.5e15        # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16 MFMA


GEMM_KERNELS = {0: 'gemm_kernel<NT> (128x128 tile)', 1: 'gemm256_kernel<false, 2> (NT, 256x256 tile)',
                2: 'gemm192_kernel (NT, 256x192 tile, 3 phases per K-tile)', 3: 'gemm256_kernel<true, 2> (TN, 256x256 tile)'}


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

    def prepare(self, n_launches):
        # events only around the dominant kernel's launches: ~100 event records per step instead of ~350, which would
        # themselves cost about 1 ms per step
        self.only = min(self.warm_flops, key=lambda v: (-self.warm_flops[v], v)) if self.warm_flops else None
        self.pool = [torch.cuda.Event(enable_timing=True) for _ in range(2 * n_launches)]
        for e in self.pool:
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
            tj = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r02_hbm_traffic.json')))
            key = GEMM_KERNELS.get(top, '').split(' (')[0]
            if key in tj['kernels']:
                traffic, src = tj['kernels'][key]['hbm_bytes_per_launch'], 'profiles/r02_hbm_traffic.json (PMC, same batch)'
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


def cpu_baseline(cfg_name: str):
    """Oracle (CPU fp32 restatement of the reference path) on a bounded sample: B=1 of the same config, one
    forward+CTC+backward after no warm-up, all host cores."""
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
        if time.perf_counter() - t0 > 10.0 or n >= 20: break     # bounded: ~10-15 s of CPU work
    dt = (time.perf_counter() - t0) / n
    res = dict(value=round(T / dt, 1), unit='spectrogram-frames/sec', cores=cores, kind='port',
               sample=f'oracle fp32 forward+CTC+backward, B=1 x T={T}, {n} timed steps after 1 warm-up ({dt:.2f} s/step)', loss=round(loss, 3))
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


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` (N > 1) without a launcher: start the N ranks as a CHILD `torch.distributed.run` - before this
    process has touched the GPU, and never by exec - relay its output (rank 0 prints the JSON line) and return its exit code."""
    import socket
    import subprocess
    have = torch.cuda.device_count()                                  # does not initialise the GPU on this image
    if have < n:
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
        res = {
            'metric': 'spectrogram-frames/sec (6L/768D, seq=16384)' if args.config == 'c3' else f'spectrogram-frames/sec ({args.config}: {cfg["name"]})',
            'value': round(frames / dt, 1), 'unit': 'spectrogram-frames/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 2), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': cfg['name'] + (', forward only (train mode, no_grad)' if args.fwd_only else ', fwd+CTC+bwd+clip+MADGRAD'), 'per_gpu_batch': B, 'global_batch': B * world, 'seq_len': T,
                       'parallelism': f'dp{world}', 'loss_last_step': round(float(loss), 3)},
            'per_gpu_value': round(frames / dt / world, 1),
            'roofline': timer.summary(),
        }
        if world == 1 and not args.no_cpu_baseline and args.config == 'c5':
            # the oracle writes attention out as an explicit (H, N, N) f32 matrix: 17 GB per layer at N = 16384, not a bounded sample
            res['cpu_baseline'] = None
        elif world == 1 and not args.no_cpu_baseline:
            torch.cuda.empty_cache()
            res['cpu_baseline'] = cpu_baseline(args.config)
            res['ctc_loss_rel_err'] = res['cpu_baseline']['ctc_loss_rel_err']
            res['gpu_over_cpu'] = round(res['value'] / res['cpu_baseline']['value'], 1)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
