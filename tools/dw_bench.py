"""Stage-2 depthwise conv of the subsampler (B, 4096, 20, 256) -> (B, 2048, 10, 256): window kernels vs the position-centric ones."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lcasr_amd.hip.ops as ops
B, C = int(os.environ.get("SUB_B", "32")), 256
Ti, Fi = 4096, 20
pre1 = torch.randn(B, Ti, Fi, C, device='cuda').bfloat16()
wd = torch.randn(C, 9, device='cuda') * 0.3; bd = torch.randn(C, device='cuda') * 0.1
dd2 = (torch.randn(B, Ti // 2, Fi // 2, C, device='cuda') * 0.1).bfloat16()
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
    y = ops.sub_dwconv_fwd(pre1, wd, bd)
    tf = t(lambda: ops.sub_dwconv_fwd(pre1, wd, bd))
    gw, gb = torch.zeros(C, 9, device='cuda'), torch.zeros(C, device='cuda')
    dx = ops.sub_dwconv_bwd(dd2, wd, pre1, gw, gb)
    tb = t(lambda: ops.sub_dwconv_bwd(dd2, wd, pre1, torch.zeros_like(gw), torch.zeros_like(gb)))
    res[mode] = (y.float(), dx.float(), gw.clone(), gb.clone())
    print(f'SCONF_SUB_MFMA={mode}: B={B} dwconv fwd {tf:.3f} ms  bwd {tb:.3f} ms')
a, b = res['0'], res['1']
for n, u, v in zip(('y', 'dx', 'dw', 'db'), a, b): print(n, 'max|d|/max', float((u - v).abs().max() / u.abs().max()))
