"""lcasr_amd — MI355X-native SConformerXL forward/backward hot path (HIP kernels behind a C ABI).

Mirrors the reference package layout for this path only:
  lcasr_amd.models.sconformer_xl.SCConformerXL   <->  lcasr/models/sconformer_xl.py
  lcasr_amd.components.*                          <->  lcasr/components/{attention,convolution,batchrenorm,
                                                       fused_dense,normalisation,wrappers,decoder,subsampling,rotary_emb}.py
  lcasr_amd.losses.CTCLoss                        <->  torch.nn.CTCLoss as used by exp/train.py:104,249
  lcasr_amd.optim.MADGRAD                         <->  lcasr/optim/madgrad.py
  lcasr_amd.hip                                   ->   ctypes binding of libsconf_hip.so (include/sconf.h)
"""
__version__ = '0.1.0'
