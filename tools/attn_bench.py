"""Micro-benchmark of the attention kernels at BASELINE config 3 shape (B=16, N=2048, H=6, D=128), random data."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lcasr_amd.hip.ops as ops
B, N, H, D = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (16, 2048, 6, 128)))
reps = int(os.environ.get('REPS', '10'))
q, k, v, do = (torch.randn(B, N, H, D, device='cuda').bfloat16() for _ in range(4))
o, lse = ops.attn_fwd(q, k, v, None)
def t(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
fl = 4.0 * B * H * N * N * D
ms = t(lambda: ops.attn_fwd(q, k, v, None))
print(f'attn fwd  {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s (algorithmic 4BHN^2D)')
ms = t(lambda: ops.attn_bwd(q, k, v, o, do, lse, None))
print(f'attn bwd  {ms*1e3:8.1f} us  {2.5*fl/ms/1e9:7.1f} TF/s (algorithmic 10BHN^2D; executed 14BHN^2D = {3.5*fl/ms/1e9:7.1f})')
ms2 = t(lambda: (ops.attn_fwd(q, k, v, None), ops.attn_bwd(q, k, v, o, do, lse, None)))
print(f'attn fwd+bwd {ms2*1e3:8.1f} us')
