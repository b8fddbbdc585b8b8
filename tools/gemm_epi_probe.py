"""Diagnostic: what the fused epilogues cost on the model's short-K shapes (SCONF_GEMM_DEBUG: 1 = no stores, 2 = no epilogue).
Needs a probe build of the library:  make -C long-context-asr_amd/csrc clean && make -C long-context-asr_amd/csrc PROBE=1"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lcasr_amd.hip.ops as ops
M = 32768
cases = [(3072, 768, 'plain', {}), (3072, 768, 'gelu_dsave', dict(act='gelu_dsave', save_pre=True, bias=True)), (3072, 768, 'mulaux', dict(act='mulaux', aux=True)),
         (768, 3072, 'plain', {}), (768, 3072, 'f32+res', dict(resid=True, out_dtype=torch.float32)), (768, 768, 'plain', {}), (768, 768, 'f32+res', dict(resid=True, out_dtype=torch.float32)),
         (4096, 768, 'plain', {}), (768, 4096, 'plain', {})]
for n, k, name, kw in cases:
    a = torch.randn(M, k, device='cuda').bfloat16(); b = torch.randn(n, k, device='cuda').bfloat16()
    kw = dict(kw)
    if kw.get('bias'): kw['bias'] = torch.randn(n, device='cuda')
    if kw.get('aux'): kw['aux'] = torch.randn(M, n, device='cuda').bfloat16()
    if kw.get('resid'): kw['resid'] = torch.randn(M, n, device='cuda')
    line = f'n={n:5d} k={k:5d} {name:11s}'
    for dbg in ('0', '1', '2', 'old'):
        os.environ.pop('SCONF_GEMM_NO_256', None); os.environ['SCONF_GEMM_DEBUG'] = dbg if dbg != 'old' else '0'
        if dbg == 'old': os.environ['SCONF_GEMM_NO_256'] = '1'
        best = 1e9
        for rnd in range(3):
            for _ in range(2): ops.gemm(a, b, 'nt', **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ops.gemm(a, b, 'nt', **kw)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        line += f'  [{dbg}] {best*1e3:7.1f} us {2.0*M*n*k/best/1e9:6.0f} TF'
    print(line, flush=True)
