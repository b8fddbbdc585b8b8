"""Run a few training steps of a BASELINE.json config on one GPU and print ms/step, frames/s, loss, peak memory.
usage: python tools/run_config.py {c1|c2|c3|c4|c5} [batch] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lcasr_amd.models.sconformer_xl import SCConformerXL
from lcasr_amd.train import Trainer, synthetic_batch
base = dict(vocab_size=4095, use_rotary=True, rotary_base_freq=1500000, decoder_norm=True, self_conditioning=True,
            default_norm='layer_norm', bias_in_ff=False)
CFG = {
    'c1': (dict(base, n_layers=6, d_model=256, n_heads=8, head_dim=32, subsampling_conv_channels=256), 1024, 2),
    'c2': (dict(base, n_layers=6, d_model=768, n_heads=6, head_dim=128, subsampling_conv_channels=256), 1024, 64),
    'c3': (dict(base, n_layers=6, d_model=768, n_heads=6, head_dim=128, subsampling_conv_channels=256), 16384, 16),
    # exp/configs/paper_templates/exp_set_seq_rotary_base_9l.yaml: per-layer checkpointing + ff_checkpoint_lvl 2
    'c4': (dict(base, n_layers=9, d_model=768, n_heads=6, head_dim=128, subsampling_conv_channels=256, checkpoint_every_n_layers=1, ff_checkpoint_lvl=2), 16384, 16),
    # exp_set_seq_rotary_base_3l_2048.yaml
    'c5': (dict(base, n_layers=3, d_model=2048, n_heads=16, head_dim=128, subsampling_conv_channels=512, ff_checkpoint_lvl=2), 131072, 2),
}
name = sys.argv[1]
kw, T, B = CFG[name]
if len(sys.argv) > 2: B = int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
torch.manual_seed(12345)
model = SCConformerXL(**kw).cuda().train()
tr = Trainer(model, global_batch=B)
batch = synthetic_batch(B, T, 4095)
losses = []
for i in range(steps + 1):
    if i == 1:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    losses.append(tr.step(*batch))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f'{name}: B={B} T={T} params={sum(p.numel() for p in model.parameters())/1e6:.1f}M  {dt*1e3:.1f} ms/step  {B*T/dt/1e6:.3f} M frames/s  '
      f'loss/frame {[round(float(l)/(B*T)*100, 4) for l in losses]}  peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB')
