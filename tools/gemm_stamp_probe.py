"""Probe build only (make -C long-context-asr_amd/csrc clean && make -C long-context-asr_amd/csrc PROBE=1): per-item time stamps of
the 256x256 GEMM kernel - how long a workgroup's epilogue takes and how far apart in time the workgroups' epilogues are."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lcasr_amd.hip.ops as ops
from lcasr_amd.hip import _lib
lib = _lib.load()
lib.sconf_gemm_probe_stamps.argtypes = [ctypes.c_void_p]
M = 131072
cases = {'gelu_dsave': (3072, 768, dict(act='gelu_dsave', save_pre=True)), 'plain': (3072, 768, {}), 'mulaux': (3072, 768, dict(act='mulaux', aux=True)), 'bias': (1536, 768, dict(bias=True)),
         'f32res768': (768, 768, dict(resid=True, out_dtype=torch.float32))}
for name in sys.argv[1:] or ['gelu_dsave', 'plain']:
    n, k, kw = cases[name]
    a = torch.randn(M, k, device='cuda').bfloat16(); b = torch.randn(n, k, device='cuda').bfloat16()
    kw = dict(kw)
    if kw.get('bias'): kw['bias'] = torch.randn(n, device='cuda')
    if kw.get('resid'): kw['resid'] = torch.randn(M, n, device='cuda')
    if kw.get('aux'): kw['aux'] = torch.randn(M, n, device='cuda').bfloat16()
    os.environ['SCONF_GEMM_256_WIDTH'] = '256'
    for stagger, mode in ((0, 0), (8, 1)):
        os.environ['SCONF_GEMM_STAGGER'] = str(stagger); os.environ['SCONF_GEMM_STAGGER_MODE'] = str(mode)
        st = torch.zeros(256, 64, 4, dtype=torch.int64, device='cuda')
        for _ in range(3): ops.gemm(a, b, 'nt', **kw)
        lib.sconf_gemm_probe_stamps(ctypes.c_void_p(st.data_ptr()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.gemm(a, b, 'nt', **kw); e1.record(); torch.cuda.synchronize()
        lib.sconf_gemm_probe_stamps(None)
        s = st.cpu().numpy()
        items = int((s[0, :, 1] != 0).sum())
        sel = s[:, 2:items - 1]                                            # steady-state items
        epi = sel[..., 2] - sel[..., 1]; drain = sel[..., 3] - sel[..., 2]
        period = np.diff(s[:, 1:items, 1], axis=1)                        # cycles between consecutive epilogue starts of a workgroup
        rt = sel[..., 0].astype(np.float64) * 10e-3                        # us (100 MHz)
        spread = rt.std(axis=0).mean()                                     # how far apart the workgroups start the same item's epilogue
        xs = np.array([rt[x::8].std(axis=0).mean() for x in range(8)]).mean()
        print(f'{name:10s} stagger {stagger} mode {mode}: {e0.elapsed_time(e1)*1e3:7.1f} us  items/WG {items}  epilogue issue {epi.mean():7.0f} cyc (p10 {np.percentile(epi,10):.0f} p90 {np.percentile(epi,90):.0f})'
              f'  store drain {drain.mean():6.0f} cyc  item period {period.mean():7.0f} cyc  start spread: all WGs {spread:5.2f} us, within an XCD {xs:5.2f} us', flush=True)
