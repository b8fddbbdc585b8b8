"""Diagnostic: the vendor BLAS (torch.matmul -> hipBLASLt / rocBLAS) on the model's GEMM shapes beside sconf_gemm_bf16 (plain epilogue).
Not part of the product path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lcasr_amd.hip.ops as ops
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
M = int(os.environ.get("BLAS_M", "262144"))
print('NT  y = x W^T (bf16 out)')
for n, k in [(3072, 768), (768, 3072), (2304, 768), (768, 768), (4096, 768), (768, 4096), (768, 2304), (1536, 768), (768, 1536)]:
    x = torch.randn(M, k, device='cuda').bfloat16(); w = torch.randn(n, k, device='cuda').bfloat16()
    a = t(lambda: ops.gemm(x, w, 'nt')); b = t(lambda: torch.nn.functional.linear(x, w))
    print(f'  {M}x{n}x{k}: sconf {2.0*M*n*k/a/1e9:5.0f} TF   blas {2.0*M*n*k/b/1e9:5.0f} TF', flush=True)
print('TN  dW = dy^T x (f32 out)')
for m, n, sp in [(3072, 768, 7), (768, 3072, 7), (768, 768, 28), (4096, 768, 5), (2304, 768, 9)]:
    dy = torch.randn(M, m, device='cuda').bfloat16(); x = torch.randn(M, n, device='cuda').bfloat16()
    a = t(lambda: ops.gemm(dy, x, 'tn', out_dtype=torch.float32, split_k=sp))
    b = t(lambda: torch.matmul(dy.t(), x))
    print(f'  {m}x{n}x{M}: sconf {2.0*M*n*m/a/1e9:5.0f} TF (split {sp})   blas(bf16 out) {2.0*M*n*m/b/1e9:5.0f} TF', flush=True)
