import os, sys
sys.path.insert(0, '/root/repo')
import torch
import lcasr_amd.hip.ops as ops
def t(fn, n=10):
    best = 1e9
    for _ in range(3):
        for _ in range(2): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best
M = 131072
for lay, m, n, k, sp in [('nt', M, 3072, 768, 1), ('nt', M, 768, 3072, 1), ('nt', M, 4096, 768, 1), ('nt', M, 768, 4096, 1), ('nt', M, 2304, 768, 1), ('tn', 3072, 768, M, 7), ('tn', 4096, 768, M, 5)]:
    if lay == 'tn': a = torch.randn(k, m, device='cuda').bfloat16(); b = torch.randn(k, n, device='cuda').bfloat16(); kw = dict(out_dtype=torch.float32, split_k=sp)
    else: a = torch.randn(m, k, device='cuda').bfloat16(); b = torch.randn(n, k, device='cuda').bfloat16(); kw = {}
    line = f'{lay} {m}x{n}x{k}:'
    for gm in ('2', '4', '8', '16', '32'):
        os.environ['SCONF_GEMM_GM'] = gm
        ms = t(lambda: ops.gemm(a, b, lay, **kw))
        line += f'  gm{gm} {2.0*m*n*k/ms/1e9:5.0f}'
    print(line, flush=True)
