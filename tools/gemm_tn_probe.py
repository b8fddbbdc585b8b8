"""Diagnostic: TN vs NT main-loop efficiency of the 256-row GEMM kernel at square long-K shapes (no split-K, plain epilogue)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
for m, n, k in [(4096, 4096, 4096), (8192, 8192, 8192), (3072, 768, 32768 // 7 // 64 * 64 * 7)]:
    a = torch.randn(m, k, device='cuda').bfloat16(); b = torch.randn(n, k, device='cuda').bfloat16()
    at = a.t().contiguous(); bt = b.t().contiguous()
    for lay, A, B, kw in [('nt', a, b, {}), ('nt f32', a, b, dict(out_dtype=torch.float32)), ('tn f32', at, bt, dict(out_dtype=torch.float32)),
                          ('tn f32 split7', at, bt, dict(out_dtype=torch.float32, split_k=7))]:
        if 'split' in lay and m != 3072: continue
        ms = t(lambda: ops.gemm(A, B, lay.split()[0], **kw))
        print(f'{m}x{n}x{k} {lay:14s} {ms*1e3:8.1f} us {2.0*m*n*k/ms/1e9:7.0f} TF', flush=True)
