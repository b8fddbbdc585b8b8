"""Runs a few NT / TN GEMMs of the 256-row kernel once each (for rocprofv3 --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lcasr_amd.hip.ops as ops
a = torch.randn(4096, 4096, device='cuda').bfloat16(); b = torch.randn(4096, 4096, device='cuda').bfloat16()
for _ in range(3):
    ops.gemm(a, b, 'nt', out_dtype=torch.float32)
    ops.gemm(a, b, 'tn', out_dtype=torch.float32)
dy = torch.randn(32768, 3072, device='cuda').bfloat16(); x = torch.randn(32768, 768, device='cuda').bfloat16()
for _ in range(3):
    ops.gemm(dy, x, 'tn', out_dtype=torch.float32, split_k=7)
torch.cuda.synchronize()
