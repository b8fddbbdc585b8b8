"""Diagnostic: per-shape time of every GEMM call inside one training step of BASELINE config 3 (B=16)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lcasr_amd.hip.ops as ops
from bench import CONFIGS
from lcasr_amd.models.sconformer_xl import SCConformerXL
from lcasr_amd.train import Trainer, synthetic_batch
cfg = CONFIGS["c3"]; B = int(os.environ.get("GB", "16"))
torch.manual_seed(12345)
model = SCConformerXL(**cfg['model']).cuda().train()
tr = Trainer(model, global_batch=B)
batch = synthetic_batch(B, cfg['T'], 4095)
for _ in range(2): tr.step(*batch)
torch.cuda.synchronize()
inner = ops.gemm
rec = collections.defaultdict(lambda: [0, 0.0, 0.0])
def gemm(a, b, layout='nt', **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    out = inner(a, b, layout, **kw)
    e1.record(); torch.cuda.synchronize()
    if layout == 'nt': m, k = a.shape; n = b.shape[0]
    elif layout == 'nn': m, k = a.shape; n = b.shape[1]
    else: k, m = a.shape; n = b.shape[1]
    key = (layout, m, n, k, kw.get('act', 'none'), 'f32' if kw.get('out_dtype') == torch.float32 else 'bf16', 'res' if kw.get('resid') is not None else '', 'pre' if kw.get('save_pre') else '', kw.get('split_k', 1))
    r = rec[key]; r[0] += 1; r[1] += e0.elapsed_time(e1); r[2] += 2.0 * m * n * k
    return out
ops.gemm = gemm
import lcasr_amd.functional as Fn
tr.step(*batch)
tot_ms = sum(r[1] for r in rec.values()); tot_f = sum(r[2] for r in rec.values())
for key, r in sorted(rec.items(), key=lambda kv: -kv[1][1]):
    print(f'{r[1]:7.2f} ms  n={r[0]:3d}  {r[2]/r[1]/1e9:7.1f} TF/s  {key}')
print(f'total {tot_ms:.2f} ms, {tot_f/tot_ms/1e9:.1f} TF/s aggregate')
