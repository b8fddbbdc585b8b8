"""Diagnostic: TN (weight-gradient) main-loop rate of the 256-row GEMM kernel; with a probe build (make PROBE=1) SCONF_GEMM_DEBUG=2
removes the epilogue."""
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
K = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
for m, n, sp in [(3072, 768, 7), (768, 3072, 7), (4096, 4096, 1), (4096, 768, 5)]:
    k = K if sp > 1 else 4096
    a = torch.randn(k, m, device='cuda').bfloat16(); b = torch.randn(k, n, device='cuda').bfloat16()
    for dbg in ('0', '2'):
        os.environ['SCONF_GEMM_DEBUG'] = dbg
        ms = t(lambda: ops.gemm(a, b, 'tn', out_dtype=torch.float32, split_k=sp))
        print(f'tn {m}x{n}x{k} split {sp} debug={dbg}: {ms*1e3:8.1f} us {2.0*m*n*k/ms/1e9:6.0f} TF', flush=True)
    if sp == 1:
        at = a.t().contiguous(); bt = b.t().contiguous()
        for dbg in ('0', '2'):
            os.environ['SCONF_GEMM_DEBUG'] = dbg
            ms = t(lambda: ops.gemm(at, bt, 'nt', out_dtype=torch.float32))
            print(f'nt {m}x{n}x{k} debug={dbg}: {ms*1e3:8.1f} us {2.0*m*n*k/ms/1e9:6.0f} TF', flush=True)
