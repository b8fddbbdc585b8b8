"""Checks the 256x256-tile GEMM kernel against the 128x128 one (bitwise: same K order per accumulator) and against an
fp32 torch product, then times both on the model's shapes.  Usage: python tools/gemm256_check.py [--bench-only]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lcasr_amd.hip.ops as ops


def both(fn, width=None):
    os.environ.pop('SCONF_GEMM_NO_256', None)
    if width: os.environ['SCONF_GEMM_256_WIDTH'] = str(width)
    new = fn()
    os.environ.pop('SCONF_GEMM_256_WIDTH', None)
    os.environ['SCONF_GEMM_NO_256'] = '1'
    old = fn()
    os.environ.pop('SCONF_GEMM_NO_256', None)
    return new, old


def check():
    torch.manual_seed(0)
    ok = True
    for (m, n, k) in [(256, 256, 64), (256, 192, 64), (256, 768, 128), (512, 256, 192), (256, 768, 768), (4096, 768, 3072), (32768, 768, 768), (8192, 2304, 768)]:
        a = torch.randn(m, k, device='cuda').bfloat16(); b = torch.randn(n, k, device='cuda').bfloat16()
        bias = torch.randn(n, device='cuda'); resid = torch.randn(m, n, device='cuda'); aux = torch.randn(m, n, device='cuda').bfloat16()
        for name, kw in [('plain', {}), ('bias+gelu_dsave', dict(bias=bias, act='gelu_dsave', save_pre=True)),
                         ('resid f32', dict(resid=resid, alpha=0.5, out_dtype=torch.float32)), ('mulaux', dict(aux=aux, act='mulaux'))]:
            for rep in range(4):
                new, old = both(lambda: ops.gemm(a, b, 'nt', **kw), width=(192 if n % 192 == 0 and rep % 2 else 256))
                new = new if isinstance(new, tuple) else (new,); old = old if isinstance(old, tuple) else (old,)
                same = all(torch.equal(x, y) for x, y in zip(new, old))
                if not same:
                    d = max((x.float() - y.float()).abs().max().item() for x, y in zip(new, old))
                    nbad = sum((x != y).sum().item() for x, y in zip(new, old))
                    print(f'NT {m}x{n}x{k} {name}: MISMATCH vs 128 kernel  max|d|={d:.4g} nbad={nbad}'); ok = False
                    break
            else:
                ref = a.float() @ b.float().t()
                if name == 'plain':
                    err = (new[0].float() - ref).abs().max().item() / ref.abs().max().item()
                    print(f'NT {m}x{n}x{k} {name}: identical to 128 kernel; rel err vs fp32 {err:.2e}')
                else:
                    print(f'NT {m}x{n}x{k} {name}: identical to 128 kernel')
    for (m, n, k, sp) in [(256, 256, 64, 1), (768, 768, 4096, 4), (3072, 768, 32768, 7), (768, 3072, 32768, 7), (768, 768, 32768, 28)]:
        a = torch.randn(k, m, device='cuda').bfloat16(); b = torch.randn(k, n, device='cuda').bfloat16()
        for rep in range(3):
            new, old = both(lambda: ops.gemm(a, b, 'tn', out_dtype=torch.float32, split_k=sp))
            if not torch.equal(new, old):
                print(f'TN {m}x{n}x{k} split {sp}: MISMATCH max|d|={(new - old).abs().max().item():.4g} nbad={(new != old).sum().item()}'); ok = False
                break
        else:
            ref = a.float().t() @ b.float()
            print(f'TN {m}x{n}x{k} split {sp}: identical to 128 kernel; rel err vs fp32 {(new - ref).abs().max().item() / ref.abs().max().item():.2e}')
    print('CHECK', 'OK' if ok else 'FAILED')
    return ok


def bench():
    M = 32768
    shapes = [('nt', M, 3072, 768, 'ff1 fwd'), ('nt', M, 768, 3072, 'ff2 fwd'), ('nt', M, 2304, 768, 'qkv fwd'), ('nt', M, 768, 2304, 'qkv dgrad'),
              ('nt', M, 4096, 768, 'vocab fwd'), ('nt', M, 768, 4096, 'reproj fwd'), ('nt', M, 768, 768, 'out/pw2'),
              ('nt', M, 1536, 768, 'pw1 fwd'), ('nt', M, 768, 1536, 'pw1 dgrad'),
              ('tn', 3072, 768, M, 'ff1 wgrad'), ('tn', 768, 3072, M, 'ff2 wgrad'), ('tn', 4096, 768, M, 'vocab wgrad'),
              ('tn', 768, 768, M, 'out wgrad'), ('tn', 2304, 768, M, 'qkv wgrad'),
              ('nt', 4096, 4096, 4096, '4096^3'), ('nt', 8192, 8192, 8192, '8192^3')]
    agg = {'new': [0.0, 0.0], 'old': [0.0, 0.0], 'w192': [0.0, 0.0], 'w192p4': [0.0, 0.0], 'auto': [0.0, 0.0]}
    for layout, m, n, k, label in shapes:
        if layout == 'nt':
            a = torch.randn(m, k, device='cuda').bfloat16(); b = torch.randn(n, k, device='cuda').bfloat16()
        else:
            a = torch.randn(k, m, device='cuda').bfloat16(); b = torch.randn(k, n, device='cuda').bfloat16()
        res = {}
        for which in ('new', 'old', 'w192', 'w192p4', 'auto'):
            os.environ.pop('SCONF_GEMM_NO_256', None); os.environ.pop('SCONF_GEMM_256_WIDTH', None); os.environ.pop('SCONF_GEMM_192_4PHASE', None)
            if which == 'old': os.environ['SCONF_GEMM_NO_256'] = '1'
            elif which == 'new': os.environ['SCONF_GEMM_256_WIDTH'] = '256'
            elif which in ('w192', 'w192p4'):
                if layout != 'nt' or n % 192: res[which] = (float('nan'), 0); continue
                os.environ['SCONF_GEMM_256_WIDTH'] = '192'
                if which == 'w192p4': os.environ['SCONF_GEMM_192_4PHASE'] = '1'
            kw = {}
            if layout == 'tn':
                kw = dict(out_dtype=torch.float32)
                if which in ('old',): kw['split_k'] = ops.pick_split_k(m, n, k)
                else: kw['split_k'] = max(1, 256 // ((m // 256) * (n // 256)))
            best = 1e9
            for rnd in range(3):
                for _ in range(2): ops.gemm(a, b, layout, **kw)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): ops.gemm(a, b, layout, **kw)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 10)
            res[which] = (best, kw.get('split_k', 1))
            if not label.endswith('^3') and which not in ('w192', 'w192p4'):
                agg[which][0] += 2.0 * m * n * k; agg[which][1] += best
        os.environ.pop('SCONF_GEMM_NO_256', None); os.environ.pop('SCONF_GEMM_256_WIDTH', None); os.environ.pop('SCONF_GEMM_192_4PHASE', None)
        fl = 2.0 * m * n * k
        print(f'{label:12s} {layout} m={m:6d} n={n:5d} k={k:6d}  256: {res["new"][0]*1e3:7.1f} us {fl/res["new"][0]/1e9:6.0f} TF (split {res["new"][1]:2d})  '
              f'192: {res["w192"][0]*1e3:7.1f} us {fl/res["w192"][0]/1e9:6.0f} TF (4-phase {fl/res["w192p4"][0]/1e9:6.0f})  auto: {fl/res["auto"][0]/1e9:6.0f} TF  '
              f'128: {res["old"][0]*1e3:7.1f} us {fl/res["old"][0]/1e9:6.0f} TF (split {res["old"][1]:2d})', flush=True)
    for w in ('new', 'old', 'auto'): print(f'aggregate {w}: {agg[w][0]/agg[w][1]/1e9:7.1f} TF/s')


if __name__ == '__main__':
    assert torch.cuda.is_available()
    if '--bench-only' in sys.argv or check():
        bench()
