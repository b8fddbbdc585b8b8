"""Lists the calls of selected lcasr_amd.hip.ops functions (shape + calling function) during one training step of config 3.
usage: python tools/op_trace.py cast colsum_ ..."""
import os, sys, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lcasr_amd.hip.ops as ops
from lcasr_amd.models.sconformer_xl import SCConformerXL
from lcasr_amd.train import Trainer, synthetic_batch
names = sys.argv[1:] or ['cast']
log = collections.Counter()
live = [False]
def wrap(name):
    fn = getattr(ops, name)
    def w(*a, **k):
        if live[0]:
            fr = [f for f in traceback.extract_stack()[:-1] if 'functional.py' in f.filename or 'train.py' in f.filename or 'optim.py' in f.filename]
            where = ' < '.join(f'{f.name}:{f.lineno}' for f in fr[-2:])
            log[(name, tuple(a[0].shape), str(a[0].dtype), where)] += 1
        return fn(*a, **k)
    setattr(ops, name, w)
for n in names: wrap(n)
torch.manual_seed(12345)
kw = dict(vocab_size=4095, use_rotary=True, rotary_base_freq=1500000, decoder_norm=True, self_conditioning=True, default_norm='layer_norm',
          bias_in_ff=False, n_layers=6, d_model=768, n_heads=6, head_dim=128, subsampling_conv_channels=256)
model = SCConformerXL(**kw).cuda().train()
tr = Trainer(model, global_batch=4)
batch = synthetic_batch(4, 16384, 4095)
tr.step(*batch); tr.step(*batch)
live[0] = True
tr.step(*batch)
torch.cuda.synchronize()
for k, v in sorted(log.items(), key=lambda kv: -kv[1]): print(v, *k)
