"""Times the conv-module kernels (GLU + depthwise conv k=9 + BatchRenorm + SiLU, forward and backward) at the benchmark shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lcasr_amd.hip.ops as ops
B, N, d, ks = (int(sys.argv[1]) if len(sys.argv) > 1 else 64), 2048, 768, 9
def t(fn, n=5):
    best = 1e9
    for _ in range(3):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best
g = torch.randn(B * N, 2 * d, device='cuda').bfloat16(); w = torch.randn(d, ks, device='cuda') * 0.2; bias = torch.randn(d, device='cuda') * 0.1
h, stats = ops.glu_dwconv_fwd(g, None, w, bias, B, N)
rm, rs, nbt = torch.zeros(d, device='cuda'), torch.ones(d, device='cuda'), torch.zeros((), dtype=torch.int64, device='cuda')
bw, bb = torch.ones(d, device='cuda'), torch.zeros(d, device='cuda')
coef = ops.brn_finalize(stats, B * N, rm, rs, nbt, bw, bb, True)
dy = torch.randn(B * N, d, device='cuda').bfloat16()
gs = [torch.zeros(d, ks, device='cuda'), torch.zeros(d, device='cuda'), torch.zeros(d, device='cuda'), torch.zeros(d, device='cuda')]
print(f'glu_dwconv_fwd {t(lambda: ops.glu_dwconv_fwd(g, None, w, bias, B, N))*1e3:7.1f} us')
print(f'affine_silu    {t(lambda: ops.affine_silu_fwd(h, coef))*1e3:7.1f} us')
print(f'convmod_bwd    {t(lambda: ops.convmod_bwd(dy, h, g, None, w, bw, coef, B, N, True, 1e-3, *gs, colsum=True))*1e3:7.1f} us (reduce + finalize + dwconv_glu_bwd + small adds)')
