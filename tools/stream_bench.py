"""Achieved HBM rate of the streaming (non-MFMA) kernels at the bench shape (config 3, B = 128: 262144 rows), beside a plain device copy.
Buffers rotate over NSET sets so that no call finds its inputs in the 256 MB Infinity Cache.
usage: python tools/stream_bench.py [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lcasr_amd.hip.ops as ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
d, C, NSET = 768, 4096, 3
dev = 'cuda'

def timeit(fn, n=12):
    for i in range(3): fn(i % NSET)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i % NSET)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

def report(name, us, nbytes):
    print(f'{name:34s} {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s (algorithmic bytes {nbytes / 1e6:.0f} MB)', flush=True)

xs = [torch.randn(M, d, device=dev) for _ in range(NSET)]
ys = [torch.empty(M, d, device=dev) for _ in range(NSET)]
report('torch copy f32', timeit(lambda i: ys[i].copy_(xs[i])), M * d * 8)
w = torch.randn(d, device=dev); b = torch.randn(d, device=dev)
report('norm_fwd f32->bf16', timeit(lambda i: ops.norm_fwd(xs[i], w, b, 'layer_norm', 1e-5, torch.bfloat16)), M * d * 6)
report('norm_fwd f32->f32', timeit(lambda i: ops.norm_fwd(xs[i], w, b, 'layer_norm', 1e-5, torch.float32)), M * d * 8)
_, mean, rstd = ops.norm_fwd(xs[0], w, b, 'layer_norm', 1e-5, torch.bfloat16)
dw = torch.zeros(d, device=dev); db = torch.zeros(d, device=dev)
dy16 = [torch.randn(M, d, device=dev).bfloat16() for _ in range(NSET)]
report('norm_bwd dy bf16 +dres', timeit(lambda i: ops.norm_bwd(dy16[i], xs[i], w, mean, rstd, 'layer_norm', 1e-5, ys[i], torch.float32, dw, db)), M * d * 14)
report('norm_bwd dy bf16 +dres +twin', timeit(lambda i: ops.norm_bwd(dy16[i], xs[i], w, mean, rstd, 'layer_norm', 1e-5, ys[i], torch.float32, dw, db, twin=True)), M * d * 16)
report('norm_bwd dy f32 (no dres)', timeit(lambda i: ops.norm_bwd(ys[i], xs[i], w, mean, rstd, 'layer_norm', 1e-5, None, torch.float32, dw, db)), M * d * 12)
report('cast f32->bf16', timeit(lambda i: ops.cast(xs[i], torch.bfloat16)), M * d * 6)
del ys
H, D = 6, 128
B, N = M // 2048, 2048
qkv = [torch.randn(M, 3 * d, device=dev).bfloat16() for _ in range(NSET)]
ang = torch.rand(N, D // 2, device=dev) * 6.28
cos, sin = torch.cos(ang).contiguous(), torch.sin(ang).contiguous()
try:
    report('rotary_inplace (q,k of qkv)', timeit(lambda i: ops.rotary_inplace_(qkv[i], cos, sin, B, N, H, D)), M * 2 * d * 4)
except Exception as e:
    print('rotary_inplace skipped:', e)
del qkv
lg = [torch.randn(M, C, device=dev).bfloat16() for _ in range(NSET)]
report('softmax_fwd bf16->bf16', timeit(lambda i: ops.softmax_fwd(lg[i], False, torch.bfloat16)), M * C * 4)
pr = ops.softmax_fwd(lg[0], False, torch.bfloat16)
report('softmax_bwd bf16', timeit(lambda i: ops.softmax_bwd(pr, lg[i], False, torch.bfloat16)), M * C * 6)
