"""Glue modules of the flow path: frozen 2-D sin-cos tables, timestep embedder, ViT connector (parameter holders
with the reference's state-dict names; modeling_utils.py:24-144).  Arithmetic runs in the HIP kernels (bagel.py)."""
import math

import numpy as np
import torch
from torch import nn


def sincos_table_2d(embed_dim, side):
    """(side*side, embed_dim) fp32 table; first half of the channels encodes the COLUMN (w) coordinate, second half the
    row; each half is [sin | cos] with omega_k = 10000^(-k / (embed_dim/4)) evaluated in fp64
    (get_2d_sincos_pos_embed, modeling_utils.py:24-66)."""
    if embed_dim % 4:
        raise AssertionError("embed_dim must be divisible by 4")
    quarter = embed_dim // 4
    omega = 1.0 / 10000 ** (np.arange(quarter, dtype=np.float64) / quarter)
    coords = np.arange(side, dtype=np.float32)
    ang = coords.astype(np.float64)[:, None] * omega[None, :]            # (side, quarter), fp64
    one_axis = np.concatenate([np.sin(ang), np.cos(ang)], axis=1)        # (side, embed_dim/2)
    w_part = np.tile(one_axis, (side, 1))                                # token (r, c) -> column c
    h_part = np.repeat(one_axis, side, axis=0)                           # token (r, c) -> row r
    return torch.from_numpy(np.concatenate([w_part, h_part], axis=1)).float()


class PositionEmbedding(nn.Module):
    def __init__(self, max_num_patch_per_side, hidden_size):
        super().__init__()
        self.max_num_patch_per_side = max_num_patch_per_side
        self.hidden_size = hidden_size
        table = sincos_table_2d(hidden_size, max_num_patch_per_side)
        # torch.empty honours the ambient device / default dtype (factory.build_bagel allocates straight on the GPU in
        # bf16); the fp32 table is rounded exactly like the reference's model.to(bf16).
        self.pos_embed = nn.Parameter(torch.empty(table.shape), requires_grad=False)
        self.pos_embed.data.copy_(table)


class _Lin(nn.Module):
    def __init__(self, i, o):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(o, i), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(o), requires_grad=False)


class TimestepEmbedder(nn.Module):
    """Linear(256,H) - SiLU - Linear(H,H) over [cos(t f) | sin(t f)], f_k = exp(-ln(1e4) k/128)."""

    def __init__(self, hidden_size, frequency_embedding_size=256):
        super().__init__()
        self.frequency_embedding_size = frequency_embedding_size
        self.mlp = nn.ModuleList([_Lin(frequency_embedding_size, hidden_size), nn.Identity(), _Lin(hidden_size, hidden_size)])
        half = frequency_embedding_size // 2
        # same fp32 op sequence as modeling_utils.py:96-98, evaluated once on the host
        self._freqs_cpu = torch.exp(-math.log(10000) * torch.arange(start=0, end=half, dtype=torch.float32) / half)
        self._freqs_dev = {}

    def freqs(self, device):
        t = self._freqs_dev.get(device)
        if t is None:
            t = self._freqs_cpu.to(device)
            self._freqs_dev[device] = t
        return t


class MLPconnector(nn.Module):
    def __init__(self, in_dim, out_dim, hidden_act):
        super().__init__()
        if hidden_act != "gelu_pytorch_tanh":
            raise NotImplementedError(f"connector activation {hidden_act} (BAGEL uses gelu_pytorch_tanh, app.py:58)")
        self.fc1 = _Lin(in_dim, out_dim)
        self.fc2 = _Lin(out_dim, out_dim)
