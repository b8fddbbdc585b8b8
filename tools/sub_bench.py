"""Micro-benchmark of the fused subsampler stage 0->1 kernels at config 3 (B=16, T=16384, C=256)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lcasr_amd.hip.ops as ops
B, F, T, C = int(os.environ.get("SUB_B", "16")), 80, 16384, 256
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
d1 = ops.sub_stage01_fwd(x, w0, b0, wd, bd)
for nb in (512, 1024, 2048, 4096, 8192):
    os.environ['SCONF_SUB_FWD_BLOCKS'] = str(nb)
    print(f'fwd blocks {nb}: {t(lambda: ops.sub_stage01_fwd(x, w0, b0, wd, bd)):.3f} ms')
dd1 = torch.randn_like(d1.float()).bfloat16()
g = [torch.zeros(C, 9, device='cuda'), torch.zeros(C, device='cuda'), torch.zeros(C, 9, device='cuda'), torch.zeros(C, device='cuda')]
for cfg in sys.argv[1:] or ['2,1024']:
    os.environ['SCONF_SUB_BWD_CFG'] = cfg
    print(f'bwd cfg {cfg:10s}: {t(lambda: ops.sub_stage01_bwd_(dd1, x, w0, b0, wd, *g)):.3f} ms')
Ti, Fi = 4096, 20
pre1 = torch.randn(B, Ti, Fi, C, device='cuda').bfloat16()
dd2 = torch.randn(B, Ti // 2, Fi // 2, C, device='cuda').bfloat16()
gw, gbv = torch.zeros(C, 9, device='cuda'), torch.zeros(C, device='cuda')
print(f'dwconv fwd: {t(lambda: ops.sub_dwconv_fwd(pre1, wd, bd)):.3f} ms')
for thr in ('256', '512'):
    os.environ['SCONF_SUB_DWBWD_THREADS'] = thr
    for cfg in ('4,512', '4,1024', '8,256', '8,512', '8,1024', '8,2048'):
        os.environ['SCONF_SUB_DWBWD_CFG'] = cfg
        print(f'dwconv bwd threads {thr} cfg {cfg}: {t(lambda: ops.sub_dwconv_bwd(dd2, wd, pre1, gw, gbv)):.3f} ms')
