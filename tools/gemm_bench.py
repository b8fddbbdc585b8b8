"""Micro-benchmark of sconf_gemm_bf16 on the GEMM shapes of BASELINE config 3 (B=16, N=2048 tokens => M=32768).
Random data (zero-filled operands read high on MI355X).  Usage: python tools/gemm_bench.py [tag]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lcasr_amd.hip.ops as ops

M = 32768
SHAPES = [  # layout, (a shape), (b shape), kwargs, label
    ('nt', (M, 768), (3072, 768), {}, 'ff1 fwd'), ('nt', (M, 3072), (768, 3072), {}, 'ff2 fwd'),
    ('nt', (M, 768), (2304, 768), {}, 'qkv fwd'), ('nt', (M, 768), (4096, 768), {}, 'vocab fwd'),
    ('nt', (M, 4096), (768, 4096), {}, 'reproj fwd'), ('nt', (M, 768), (768, 768), {}, 'out/pw2 fwd'),
    ('nn', (M, 3072), (3072, 768), {}, 'ff1 dgrad'), ('nn', (M, 768), (768, 3072), {}, 'ff2 dgrad'),
    ('nn', (M, 4096), (4096, 768), {}, 'vocab dgrad'), ('nn', (M, 768), (768, 4096), {}, 'reproj dgrad'),
    ('tn', (M, 3072), (M, 768), dict(out_dtype=torch.float32), 'ff1 wgrad'), ('tn', (M, 768), (M, 3072), dict(out_dtype=torch.float32), 'ff2 wgrad'),
    ('tn', (M, 4096), (M, 768), dict(out_dtype=torch.float32), 'vocab wgrad'), ('tn', (M, 768), (M, 768), dict(out_dtype=torch.float32), 'out wgrad'),
    ('nt', (4096, 4096), (4096, 4096), {}, '4096^3 nt'), ('nt', (8192, 8192), (8192, 8192), {}, '8192^3 nt'),
]
tag = sys.argv[1] if len(sys.argv) > 1 else ''
tot_f, tot_t = 0.0, 0.0
for layout, sa, sb, kw, label in SHAPES:
    a = torch.randn(*sa, device='cuda').bfloat16(); b = torch.randn(*sb, device='cuda').bfloat16()
    if layout == 'nt': m, k = sa; n = sb[0]
    elif layout == 'nn': m, k = sa; n = sb[1]
    else: k, m = sa; n = sb[1]
    kw = dict(kw)
    if layout == 'tn': kw['split_k'] = ops.pick_split_k(m, n, k)
    for _ in range(3): ops.gemm(a, b, layout, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps): ops.gemm(a, b, layout, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * m * n * k
    if not label.endswith('^3 nt'): tot_f += fl; tot_t += ms
    print(f'{tag:8s} {label:14s} {layout} m={m:6d} n={n:5d} k={k:6d} split={kw.get("split_k",1):2d}  {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s', flush=True)
print(f'{tag:8s} model-shape aggregate: {tot_f/tot_t/1e9:7.1f} TF/s')
