"""Trains config 2 (6L/768D, T=1024) on ONE fixed synthetic batch for N steps: the CTC loss per frame must collapse.
A whole-loop sanity check (forward, CTC, backward, clip, MADGRAD) after kernel changes.  usage: python tools/overfit_check.py [steps] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lcasr_amd.models.sconformer_xl import SCConformerXL
from lcasr_amd.train import Trainer, synthetic_batch
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
T = 1024
torch.manual_seed(12345)
model = SCConformerXL(vocab_size=4095, use_rotary=True, rotary_base_freq=1500000, decoder_norm=True, self_conditioning=True, default_norm='layer_norm',
                      bias_in_ff=False, n_layers=6, d_model=768, n_heads=6, head_dim=128, subsampling_conv_channels=256).cuda().train()
tr = Trainer(model, global_batch=B)
batch = synthetic_batch(B, T, 4095)
losses = [float(tr.step(*batch)) / (B * T) * 100 for _ in range(steps)]
print(f'loss/frame: first {losses[0]:.3f}, step 32 {losses[min(32, steps - 1)]:.4f}, last {losses[-1]:.6f}, min {min(losses):.6f}; all finite: {all(l == l and abs(l) != float("inf") for l in losses)}')
assert losses[-1] < 0.01 * losses[0], 'the loop does not train'
