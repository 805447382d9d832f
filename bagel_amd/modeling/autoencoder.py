"""FLUX-style KL autoencoder (API of the reference's modeling/autoencoder.py: AutoEncoderParams :20-31, AutoEncoder
:290-322, load_ae :339-360).  Parameter names/shapes equal the reference state dict (``ae.safetensors`` loads
unchanged); the arithmetic runs in the HIP kernels of bagel_amd/csrc/vae.hip (fp32, as the reference's VAE)."""
from dataclasses import dataclass
from typing import List

import torch
from torch import nn

from .packed import PackedWeights


@dataclass
class AutoEncoderParams:
    resolution: int
    in_channels: int
    downsample: int
    ch: int
    out_ch: int
    ch_mult: List[int]
    num_res_blocks: int
    z_channels: int
    scale_factor: float
    shift_factor: float


def _conv(cin, cout, k):
    m = nn.Module()
    m.weight = nn.Parameter(torch.empty(cout, cin, k, k), requires_grad=False)
    m.bias = nn.Parameter(torch.empty(cout), requires_grad=False)
    m.kernel = k
    return m


def _gn(c):
    m = nn.Module()
    m.weight = nn.Parameter(torch.ones(c), requires_grad=False)
    m.bias = nn.Parameter(torch.zeros(c), requires_grad=False)
    return m


def _res(cin, cout):
    m = nn.Module()
    m.in_channels, m.out_channels = cin, cout
    m.norm1, m.conv1, m.norm2, m.conv2 = _gn(cin), _conv(cin, cout, 3), _gn(cout), _conv(cout, cout, 3)
    if cin != cout:
        m.nin_shortcut = _conv(cin, cout, 1)
    return m


def _attn(c):
    m = nn.Module()
    m.norm, m.q, m.k, m.v, m.proj_out = _gn(c), _conv(c, c, 1), _conv(c, c, 1), _conv(c, c, 1), _conv(c, c, 1)
    return m


def _wrap_conv(c):
    m = nn.Module()
    m.conv = c
    return m


class _Encoder(nn.Module):
    def __init__(self, p: AutoEncoderParams):
        super().__init__()
        ch, mult, nb = p.ch, list(p.ch_mult), p.num_res_blocks
        self.conv_in = _conv(p.in_channels, ch, 3)
        in_mult = [1] + mult
        self.down = nn.ModuleList()
        bi = ch
        for lvl in range(len(mult)):
            d = nn.Module()
            d.block = nn.ModuleList()
            d.attn = nn.ModuleList()
            bi, bo = ch * in_mult[lvl], ch * mult[lvl]
            for _ in range(nb):
                d.block.append(_res(bi, bo))
                bi = bo
            if lvl != len(mult) - 1:
                d.downsample = _wrap_conv(_conv(bi, bi, 3))
            self.down.append(d)
        self.mid = nn.Module()
        self.mid.block_1, self.mid.attn_1, self.mid.block_2 = _res(bi, bi), _attn(bi), _res(bi, bi)
        self.norm_out = _gn(bi)
        self.conv_out = _conv(bi, 2 * p.z_channels, 3)


class _Decoder(nn.Module):
    def __init__(self, p: AutoEncoderParams):
        super().__init__()
        ch, mult, nb = p.ch, list(p.ch_mult), p.num_res_blocks
        bi = ch * mult[-1]
        self.conv_in = _conv(p.z_channels, bi, 3)
        self.mid = nn.Module()
        self.mid.block_1, self.mid.attn_1, self.mid.block_2 = _res(bi, bi), _attn(bi), _res(bi, bi)
        ups = []
        for lvl in reversed(range(len(mult))):
            u = nn.Module()
            u.block = nn.ModuleList()
            u.attn = nn.ModuleList()
            bo = ch * mult[lvl]
            for _ in range(nb + 1):
                u.block.append(_res(bi, bo))
                bi = bo
            if lvl != 0:
                u.upsample = _wrap_conv(_conv(bi, bi, 3))
            ups.insert(0, u)
        self.up = nn.ModuleList(ups)
        self.norm_out = _gn(bi)
        self.conv_out = _conv(bi, p.out_ch, 3)


class AutoEncoder(PackedWeights):
    def __init__(self, params: AutoEncoderParams):
        super().__init__()
        self.params = params
        self.encoder = _Encoder(params)
        self.decoder = _Decoder(params)
        self.scale_factor = params.scale_factor
        self.shift_factor = params.shift_factor
        self._engine = None
        self._engine_bf16 = None

    def _drop_packed(self):
        self._engine = None
        self._engine_bf16 = None

    def _eng(self, precision=None):
        """``precision``: "fp32" (the VAE outside any autocast region: app.py:48,138, gen_images_mp.py:93), "bf16" (the VAE as the
        reference's InterleaveInferencer runs it, inside torch.autocast(bf16): inferencer.py:233 -> :174-185) or None = FOLLOW THE CALLER
        like the reference's modules do: bf16 when called inside ``torch.autocast("cuda", dtype=torch.bfloat16)``, fp32 otherwise."""
        if precision is None:
            precision = "bf16" if (torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16) else "fp32"
        if precision not in ("fp32", "bf16"):
            raise ValueError(f"precision={precision!r}: 'fp32', 'bf16' or None (follow the caller's autocast region)")
        self._check_packed()
        from .vae_engine import VaeEngine, VaeEngineBf16
        if precision == "bf16":
            if getattr(self, "_engine_bf16", None) is None:
                self._engine_bf16 = VaeEngineBf16(self)
                self._packed_fresh()
            return self._engine_bf16
        if self._engine is None:
            self._engine = VaeEngine(self)
            self._packed_fresh()
        return self._engine

    @torch.no_grad()
    def encode(self, x, sample_noise=None, precision=None):
        """z = scale * (mean + exp(0.5 logvar) * eps - shift)  (autoencoder.py:275-287,315-318).  ``eps`` is drawn with
        torch.randn on the HOST generator (same stream position as the reference's randn_like) unless given."""
        return self._eng(precision).encode(x, sample_noise)

    @torch.no_grad()
    def decode(self, z, precision=None):
        return self._eng(precision).decode(z)

    def forward(self, x):
        return self.decode(self.encode(x))


def load_ae(local_path: str):
    ae_params = AutoEncoderParams(resolution=256, in_channels=3, downsample=8, ch=128, out_ch=3, ch_mult=[1, 2, 4, 4],
                                  num_res_blocks=2, z_channels=16, scale_factor=0.3611, shift_factor=0.1159)
    ae = AutoEncoder(ae_params)
    if local_path is not None:
        from safetensors.torch import load_file
        sd = load_file(local_path)
        missing, unexpected = ae.load_state_dict(sd, strict=False)
        if missing or unexpected:
            print(f"load_ae: {len(missing)} missing / {len(unexpected)} unexpected keys")
    return ae, ae_params
