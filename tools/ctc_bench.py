"""Micro-benchmark of the CTC kernels at config 3 (B=16, N=2048 frames, 512 labels, 4096 classes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lcasr_amd.hip.ops as ops
B, N, C, S = 16, 2048, 4096, 512
lp = torch.log_softmax(torch.randn(B, N, C, device='cuda'), -1)
tg = torch.randint(0, C - 1, (B, S), device='cuda', dtype=torch.int32)
il = torch.full((B,), N, device='cuda', dtype=torch.int32); tl = torch.full((B,), S, device='cuda', dtype=torch.int32)
def t(fn, n=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for nt in sys.argv[1:] or ['0']:
    os.environ['SCONF_CTC_THREADS'] = nt
    nll, ws = ops.ctc_fwd(lp, tg, il, tl, C - 1)
    print(f'threads {nt}: fwd {t(lambda: ops.ctc_fwd(lp, tg, il, tl, C - 1)):.3f} ms  bwd {t(lambda: ops.ctc_bwd(lp, ws, nll, tg, il, tl, None, C - 1)):.3f} ms  nll[0]={float(nll[0]):.3f}')
# head + loss as one operator: the logits forms beside the separate operators (B from argv via env CTC_B, default 128)
B2 = int(os.environ.get('CTC_B', '128'))
lg = torch.randn(B2, N, C, device='cuda')
tg2 = torch.randint(0, C - 1, (B2, S), device='cuda', dtype=torch.int32)
il2 = torch.full((B2,), N, device='cuda', dtype=torch.int32); tl2 = torch.full((B2,), S, device='cuda', dtype=torch.int32)
os.environ.pop('SCONF_CTC_THREADS', None)
nll, ws = ops.ctc_fwd_logits(lg, tg2, il2, tl2, C - 1)
cs = torch.zeros(C, device='cuda')
print(f'B={B2} logits form: fwd (lse + gather + lattice) {t(lambda: ops.ctc_fwd_logits(lg, tg2, il2, tl2, C - 1)):.3f} ms   '
      f'bwd (d nll / d logits, bf16 + column sums) {t(lambda: ops.ctc_bwd_logits(lg, ws, nll, tg2, il2, tl2, None, C - 1, colsum_into=cs)):.3f} ms')
lp2 = ops.softmax_fwd(lg, True, torch.float32)
nll2, ws2 = ops.ctc_fwd(lp2, tg2, il2, tl2, C - 1)
gr = ops.ctc_bwd(lp2, ws2, nll2, tg2, il2, tl2, None, C - 1)
print(f'B={B2} separate: log_softmax {t(lambda: ops.softmax_fwd(lg, True, torch.float32)):.3f} + ctc_fwd {t(lambda: ops.ctc_fwd(lp2, tg2, il2, tl2, C - 1)):.3f} ms   '
      f'ctc_bwd {t(lambda: ops.ctc_bwd(lp2, ws2, nll2, tg2, il2, tl2, None, C - 1)):.3f} + log_softmax bwd {t(lambda: ops.softmax_bwd(lp2, gr, True, torch.bfloat16, colsum_into=cs)):.3f} ms')
