"""Micro-benchmark of the row-norm kernels at the model's shape (32768 x 768, f32 residual stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lcasr_amd.hip.ops as ops
M, d = int(os.environ.get('NORM_M', '32768')), int(os.environ.get('NORM_D', '768'))
x = torch.randn(M, d, device='cuda'); w = torch.randn(d, device='cuda'); b = torch.randn(d, device='cuda')
y, mean, rstd = ops.norm_fwd(x, w, b, 'layer_norm', 1e-5, torch.bfloat16)
dres = torch.randn(M, d, device='cuda'); dw = torch.zeros(d, device='cuda'); db = torch.zeros(d, device='cuda')
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for gdt in (torch.bfloat16, torch.float32):
    dy = torch.randn(M, d, device='cuda').to(gdt)
    us = timeit(lambda: ops.norm_bwd(dy, x, w, mean, rstd, 'layer_norm', 1e-5, dres, torch.float32, dw, db))
    nbytes = M * d * (dy.element_size() + 4 + 4 + 4)
    print(f'norm_bwd dy={gdt} : {us:7.1f} us  {nbytes/us/1e6:6.2f} TB/s')
us = timeit(lambda: ops.norm_fwd(x, w, b, 'layer_norm', 1e-5, torch.bfloat16))
print(f'norm_fwd f32->bf16: {us:7.1f} us  {M*d*6/us/1e6:6.2f} TB/s')
if len(sys.argv) > 1:
    dy = torch.randn(M, d, device='cuda').bfloat16()
    for cfg in sys.argv[1:]:
        os.environ['SCONF_NORM_BWD_GRID'] = cfg            # max workgroups
        us = timeit(lambda: ops.norm_bwd(dy, x, w, mean, rstd, 'layer_norm', 1e-5, dres, torch.float32, dw, db))
        print(f'cfg {cfg:12s}: {us:7.1f} us  {M*d*14/us/1e6:6.2f} TB/s')
