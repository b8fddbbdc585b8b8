"""Interleaved A/B of the attention kernels in ONE process (boxes of the pool differ by several per cent): the round-2 8-wave
forward (SCONF_ATTN_RC=0) as the box's yardstick, the current forward, the current backward.  B N H D as attn_bench.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lcasr_amd.hip.ops as ops
B, N, H, D = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (128, 2048, 6, 128)))
reps, rounds = int(os.environ.get('REPS', '10')), int(os.environ.get('ROUNDS', '5'))
q, k, v, do = (torch.randn(B, N, H, D, device='cuda').bfloat16() for _ in range(4))
o, lse = ops.attn_fwd(q, k, v, None)
def t(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
def fwd(rc):
    os.environ['SCONF_ATTN_RC'] = rc
    return t(lambda: ops.attn_fwd(q, k, v, None))
res = {'fwd_r2': [], 'fwd': [], 'bwd': []}
for _ in range(rounds):
    res['fwd_r2'].append(fwd('0')); res['fwd'].append(fwd('1')); res['bwd'].append(t(lambda: ops.attn_bwd(q, k, v, o, do, lse, None)))
fl = 4.0 * B * H * N * N * D
for name, mult in (('fwd_r2', 1.0), ('fwd', 1.0), ('bwd', 2.5)):
    xs = sorted(res[name]); med = xs[len(xs) // 2]
    print(f'{name:7s} median {med*1e3:8.1f} us  min {xs[0]*1e3:8.1f} us  {mult*fl/med/1e9:7.1f} TF/s algorithmic ({mult*fl/med/1e9/2500:.3f} of 2.5 PF)')
print(f'fwd / fwd_r2 = {sorted(res["fwd"])[rounds//2] / sorted(res["fwd_r2"])[rounds//2]:.3f}   bwd / fwd_r2 = {sorted(res["bwd"])[rounds//2] / sorted(res["fwd_r2"])[rounds//2]:.3f}')
