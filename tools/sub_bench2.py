"""Fused subsampler stage 0->1: MFMA kernels (default) against the VALU kernels (SCONF_SUB_MFMA=0), time and max difference."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lcasr_amd.hip.ops as ops
B, F, T, C = int(os.environ.get("SUB_B", "32")), 80, 16384, int(os.environ.get("SUB_C", "256"))
x = torch.randn(B, F, T, device='cuda')
w0 = torch.randn(C, 9, device='cuda') * 0.3; b0 = torch.randn(C, device='cuda') * 0.1
wd = torch.randn(C, 9, device='cuda') * 0.3; bd = torch.randn(C, device='cuda') * 0.1
def t(fn, n=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
res = {}
for mode in ('0', '1'):
    os.environ['SCONF_SUB_MFMA'] = mode
    d1 = ops.sub_stage01_fwd(x, w0, b0, wd, bd)
    tf = t(lambda: ops.sub_stage01_fwd(x, w0, b0, wd, bd))
    dd1 = (torch.randn(d1.shape, device='cuda', generator=torch.Generator('cuda').manual_seed(1)) * 0.1).bfloat16()
    g = [torch.zeros(C, 9, device='cuda'), torch.zeros(C, device='cuda'), torch.zeros(C, 9, device='cuda'), torch.zeros(C, device='cuda')]
    ops.sub_stage01_bwd_(dd1, x, w0, b0, wd, *g)
    tb = t(lambda: ops.sub_stage01_bwd_(dd1, x, w0, b0, wd, *[torch.zeros_like(v) for v in g]))
    res[mode] = (d1.float(), [v.clone() for v in g])
    print(f'SCONF_SUB_MFMA={mode}: B={B} C={C} fwd {tf:.3f} ms  bwd {tb:.3f} ms')
a, b = res['0'], res['1']
print('fwd max|d|/max', float((a[0] - b[0]).abs().max() / a[0].abs().max()))
for n, u, v in zip(('dw0', 'db0', 'dwd', 'dbd'), a[1], b[1]):
    print(n, 'max|d|/max', float((u - v).abs().max() / u.abs().max()))
