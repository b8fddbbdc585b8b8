import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import lcasr_amd.hip.ops as ops
from lcasr_amd.hip import _lib
import kernel_refs as R
torch.manual_seed(0)
B,N,H,D=1,512,1,128
q,k,v,do=(torch.randn(B,N,H,D).bfloat16() for _ in range(4))
o,lse=ops.attn_fwd(q.cuda(),k.cuda(),v.cuda(),None)
orf,lser=R.attn_fwd(q,k,v,None,(-1,-1))
print('fwd err', float((o.float().cpu()-orf.float()).abs().max()), float((lse.cpu()-lser).abs().max()))
# call bwd through ops but keep the workspace: replicate ops.attn_bwd
import ctypes
dq=torch.empty(B,N,H,D,dtype=torch.bfloat16,device='cuda'); dk=torch.empty_like(dq); dv=torch.empty_like(dq)
delta=torch.full((2,B,H,N),7.0,dtype=torch.float32,device='cuda')
_p=ops._p; s3=ops._strides3
qc,kc,vc,doc=q.cuda(),k.cuda(),v.cuda(),do.cuda()
_lib.call('sconf_attn_bwd', _p(qc),_p(kc),_p(vc),_p(o),_p(doc),_p(lse),_p(delta),_p(dq),_p(dk),_p(dv),None,B,N,H,D,s3(qc),s3(kc),s3(vc),s3(o),s3(doc),s3(dq),s3(dk),s3(dv),-1,-1,D**-0.5,None,None,ops._stream())
torch.cuda.synchronize()
dl=(do.float()*orf.float()).sum(-1).permute(0,2,1)   # (B,H,N)
print('stat0 vs -delta', float((delta[0].cpu()+dl).abs().max()), 'stat1 vs -lse2', float((delta[1].cpu()+lser*1.4426950408889634).abs().max()))
dqr,dkr,dvr=R.attn_bwd(q,k,v,orf,do,lser,None,(-1,-1))
for n,a,b in (('dq',dq,dqr),('dk',dk,dkr),('dv',dv,dvr)):
    print(n, 'max err', float((a.float().cpu()-b.float()).abs().max()), 'max ref', float(b.float().abs().max()))
