"""FusedMLP (lcasr/components/fused_dense.py:425-498): Linear(d->4d) + GELU(tanh) + Linear(4d->d).

Replaces fused_dense_cuda.linear_act_forward / bias_act_linear_dgrad_bgrad / linear_bias_wgrad with the HIP
GEMM epilogues (csrc/gemm.hip).  fc1/fc2 are torch.nn.Linear parameter containers so that initialisation and
state_dict keys are identical to the reference."""
import torch.nn as nn

from .. import functional as Fn


class FusedMLP(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, bias1=True, bias2=True,
                 activation='gelu_approx', return_residual=False, checkpoint_lvl=0, heuristic='auto', device=None, dtype=None):
        assert checkpoint_lvl in [0, 1, 2]
        if activation != 'gelu_approx':
            raise NotImplementedError("only activation='gelu_approx' is used by SConformerXL")
        if return_residual:
            raise NotImplementedError('return_residual is not used by SConformerXL')
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features * 4
        self.activation, self.return_residual, self.checkpoint_lvl, self.heuristic = activation, return_residual, checkpoint_lvl, heuristic
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias1, device=device, dtype=dtype)
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias2, device=device, dtype=dtype)

    def forward_prenorm(self, x, norm, residual, scale=1.0):
        """norm None: x is already normalised (the module-level `forward`)."""
        shape = x.shape
        nw, nb = norm.norm_params() if norm is not None else (None, None)
        mode, eps = (norm.mode, norm.eps) if norm is not None else ('none', 0.0)
        y = Fn.ff_block(x.reshape(-1, shape[-1]), nw, nb, self.fc1.weight, self.fc2.weight, self.fc1.bias, self.fc2.bias,
                        scale, mode, eps, self.checkpoint_lvl, residual)
        return y.view(*shape[:-1], y.shape[-1])

    def forward(self, x, process_group=None):
        """fc2(gelu_tanh(fc1(x))) — the reference's module-level call (fused_dense.py:489-498)."""
        if process_group is not None:
            raise NotImplementedError('tensor-parallel FusedMLP is dead code in the reference (SURVEY §2) and not implemented')
        return self.forward_prenorm(x, None, False)
