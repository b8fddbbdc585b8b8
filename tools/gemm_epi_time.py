"""Times the model's NT GEMM calls with their real epilogues (plain / bias / f32 + residual / gelu' save / * aux) at M = 131072."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lcasr_amd.hip.ops as ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
def t(fn, n=8):
    best = 1e9
    for _ in range(3):
        for _ in range(2): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best
cases = [('qkv           plain', 2304, 768, {}), ('pw1      bf16+bias', 1536, 768, dict(bias=True)), ('ff1     gelu_dsave', 3072, 768, dict(act='gelu_dsave', save_pre=True)),
         ('ff2        f32+res', 768, 3072, dict(resid=True, out_dtype=torch.float32, alpha=0.5)), ('out        f32+res', 768, 768, dict(resid=True, out_dtype=torch.float32)),
         ('pw2   f32+res+bias', 768, 768, dict(resid=True, out_dtype=torch.float32, bias=True)), ('vocab    bf16+bias', 4096, 768, dict(bias=True)),
         ('reproj f32+res+bias', 768, 4096, dict(resid=True, out_dtype=torch.float32, bias=True)), ('ff dgrad    mulaux', 3072, 768, dict(act='mulaux', aux=True, alpha=0.5)),
         ('ff dgrad2    plain', 768, 3072, {}), ('head   f32 + bias', 4096, 768, dict(out_dtype=torch.float32, bias=True))]
tot = 0.0
for name, n, k, kw in cases:
    a = torch.randn(M, k, device='cuda').bfloat16(); b = torch.randn(n, k, device='cuda').bfloat16()
    kw = dict(kw)
    if kw.get('bias'): kw['bias'] = torch.randn(n, device='cuda')
    if kw.get('aux'): kw['aux'] = torch.randn(M, n, device='cuda').bfloat16()
    if kw.get('resid'): kw['resid'] = torch.randn(M, n, device='cuda')
    ms = t(lambda: ops.gemm(a, b, 'nt', **kw)); tot += ms
    print(f'{name:20s} n={n:5d} k={k:5d}: {ms*1e3:7.1f} us {2.0*M*n*k/ms/1e9:6.0f} TF', flush=True)
print(f'sum {tot*1e3:.1f} us')
