"""RotaryPositionalEmbedding (lcasr/components/rotary_emb.py:4-57): same buffers (`inv_freq`,
`rotary_interpolation_factor`) and the same table arithmetic; the rotation itself is applied inside the
qkv de-interleave kernel (csrc/elementwise.hip, rotary_emb.py:61-73)."""
import torch


class RotaryPositionalEmbedding(torch.nn.Module):
    def __init__(self, dim, base=10000, learned_freq=False, rotary_interpolation_factor=1.0, precision=torch.bfloat16):
        super().__init__()
        if learned_freq:
            raise NotImplementedError('learned rotary frequencies are not on the benchmarked hot path')
        inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim))
        self.learned_freq = learned_freq
        self.dim = dim
        self.register_buffer('inv_freq', inv_freq)
        self.seq_len_cached = None
        self.cos_cached = None
        self.sin_cached = None
        self.precision = precision
        self.register_buffer('rotary_interpolation_factor', torch.tensor(rotary_interpolation_factor))

    def reset_if_needed(self):
        if self.learned_freq:
            self.cos_cached = self.sin_cached = self.seq_len_cached = None

    def forward(self, seq_len, device=torch.device('cpu')):
        """Reference-shaped tables (1, seq_len, 1, dim) f32 — rotary_emb.py:44-57."""
        seq_len = int(seq_len)
        if seq_len != self.seq_len_cached or self.cos_cached.device != torch.device(device):
            self.seq_len_cached = seq_len
            t = torch.arange(seq_len, device=device).type_as(self.inv_freq) / self.rotary_interpolation_factor
            freqs = torch.einsum('i,j->ij', t, self.inv_freq.to(device))
            emb = torch.cat((freqs, freqs), dim=-1).to(device)
            self.cos_cached = emb.cos()[None, :, None, :]
            self.sin_cached = emb.sin()[None, :, None, :]
        return self.cos_cached, self.sin_cached

    def tables(self, seq_len, device):
        """Compact (seq_len, dim/2) f32 tables for the HIP kernel (emb = cat(freqs, freqs) ⇒ both halves equal)."""
        cos, sin = self.forward(seq_len, device)
        h = self.dim // 2
        return cos[0, :, 0, :h].contiguous(), sin[0, :, 0, :h].contiguous()
