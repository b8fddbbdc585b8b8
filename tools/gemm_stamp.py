"""Stamp build of the 256-row GEMM (make EXTRA=-DSCONF_GEMM_STAMP, SCONF_GEMM_STAMP_PRINT=1): cycle shares per K-tile on a few shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lcasr_amd.hip.ops as ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
for (n, k) in ((3072, 768), (768, 3072), (4096, 768)):
    a = torch.randn(M, k, device='cuda').bfloat16(); b = torch.randn(n, k, device='cuda').bfloat16()
    os.environ.pop('SCONF_GEMM_STAMP_PRINT', None)
    for _ in range(3): ops.gemm(a, b, 'nt')
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.gemm(a, b, 'nt')
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f'NT M={M} N={n} K={k}: {ms*1e3:.1f} us  {2.0*M*n*k/ms/1e9:.0f} TF/s (stamp build)', flush=True)
    os.environ['SCONF_GEMM_STAMP_PRINT'] = '1'
    ops.gemm(a, b, 'nt'); torch.cuda.synchronize()
a = torch.randn(4096, 4096, device='cuda').bfloat16(); b = torch.randn(4096, 4096, device='cuda').bfloat16()
ops.gemm(a, b, 'nt'); torch.cuda.synchronize()
