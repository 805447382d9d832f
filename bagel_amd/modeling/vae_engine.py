"""Execution plan of the FLUX-style VAE on MI355X: NHWC fp32 activations, every conv an implicit GEMM on the exact-fp32
MFMA, GroupNorm+swish fused, nearest-2x upsample folded into the following conv's loader, ResnetBlock / AttnBlock
skip adds folded into conv epilogues.  Mirrors Encoder.forward / Decoder.forward of the reference
(modeling/autoencoder.py:172-193, 250-272) block for block."""
import torch

from .. import ops

F32 = torch.float32


def _ceil_to(x, m):
    return (x + m - 1) // m * m


def _pack3(w, cin_pad):
    """[Cout, Cin, 3, 3] -> [Cout, 9 * cin_pad] tap-major (dy, dx, cin), zero-padded channels."""
    cout, cin = w.shape[:2]
    t = w.permute(0, 2, 3, 1)
    if cin_pad != cin:
        t = torch.cat([t, t.new_zeros((cout, 3, 3, cin_pad - cin))], 3)
    return t.reshape(cout, 9 * cin_pad).contiguous()


class VaeEngine:
    ATTN_ROWS = 2048       # query rows per score block of the mid-block attention

    def __init__(self, ae):
        p0 = ae.encoder.conv_in.weight
        ops.require_gpu_f32(p0, "VaeEngine")
        self.ae = ae
        self.dev = p0.device
        self.P = ae.params
        self._w = {}
        self._ws = None

    # -- packed weights, cached by module identity
    def _conv_w(self, m):
        w = self._w.get(id(m))
        if w is None:
            wt = m.weight.data
            if wt.shape[-1] == 3:
                cin_pad = _ceil_to(wt.shape[1], 32)
                w = (_pack3(wt, cin_pad), cin_pad)
            else:
                w = (wt.reshape(wt.shape[0], wt.shape[1]).contiguous(), wt.shape[1])
            self._w[id(m)] = w
        return w

    def conv(self, x, m, mode, residual=None, out_hw=None):
        """x: [B,H,W,C] NHWC fp32 -> [B,Ho,Wo,Cout]."""
        B, H, W, C = x.shape
        w, cin = self._conv_w(m)
        if cin != C:
            raise ValueError(f"conv expects {cin} (padded) input channels, got {C}")
        Ho, Wo = (H, W) if mode in (0, 1) else ((H // 2, W // 2) if mode == 2 else (2 * H, 2 * W))
        cout = w.shape[0]
        ldo = _ceil_to(cout, 4)          # rows stay 16-byte aligned (conv_out has 3 channels)
        if residual is not None and (ldo != cout or residual.shape[-1] != cout or not residual.is_contiguous()):
            raise ValueError("residual must be a contiguous NHWC tensor with Cout % 4 == 0 channels")
        out = torch.empty((B, Ho, Wo, ldo), dtype=F32, device=x.device)
        ops.conv_gemm_f32(x, C, w, w.shape[1], m.bias.data, residual, out, ldo, B, H, W, C, Ho, Wo, cout, mode)
        return out if ldo == cout else out[..., :cout]

    def gemm_nt(self, a, b):
        """a:[M,K] b:[N,K] -> a @ b^T  (attention products)."""
        M, K = a.shape
        N = b.shape[0]
        out = torch.empty((M, N), dtype=F32, device=a.device)
        ops.conv_gemm_f32(a, a.stride(0), b, b.stride(0), None, None, out, N, 1, 1, M, K, 1, M, N, 0)
        return out

    def gn(self, x, m, swish):
        B, H, W, C = x.shape
        need = B * 32 * (64 * 2 + 2)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=F32, device=x.device)
        y = torch.empty_like(x)
        ops.groupnorm_f32(x, y, self._ws, m.weight.data, m.bias.data, B, H * W, C, 32, 1e-6, swish)
        return y

    def res(self, x, m):
        h = self.conv(self.gn(x, m.norm1, True), m.conv1, 1)
        skip = self.conv(x, m.nin_shortcut, 0) if hasattr(m, "nin_shortcut") else x
        return self.conv(self.gn(h, m.norm2, True), m.conv2, 1, residual=skip)

    def attn(self, x, m):
        B, H, W, C = x.shape
        h = self.gn(x, m.norm, False)
        q, k, v = self.conv(h, m.q, 0), self.conv(h, m.k, 0), self.conv(h, m.v, 0)
        N = H * W
        o = torch.empty((B, N, C), dtype=F32, device=x.device)
        # fp32 attention like the reference (autoencoder.py:52-60).  The score matrix is produced in blocks of ATTN_ROWS query rows (rows are
        # independent: the results do not depend on the block size): at 1024^2 the mid block has 16 384 tokens and the whole matrix would be
        # a 1.07 GB transient; a 2048-row block is 134 MB and is reused by every block.
        R = min(N, self.ATTN_ROWS)
        sbuf = torch.empty((R, N), dtype=F32, device=x.device)
        for b in range(B):
            qb, kb = q[b].view(N, C), k[b].view(N, C)
            vt = v[b].view(N, C).t().contiguous()          # layout copy only
            for r0 in range(0, N, R):
                n = min(R, N - r0)
                s = sbuf[:n]
                ops.conv_gemm_f32(qb[r0:r0 + n], qb.stride(0), kb, kb.stride(0), None, None, s, N, 1, 1, n, C, 1, n, N, 0)
                ops.softmax_rows_f32(s, s.stride(0), n, N, float(C) ** -0.5)
                ob = o[b, r0:r0 + n]
                ops.conv_gemm_f32(s, s.stride(0), vt, vt.stride(0), None, None, ob, C, 1, 1, n, N, 1, n, C, 0)
        return self.conv(o.view(B, H, W, C), m.proj_out, 0, residual=x)

    @staticmethod
    def to_nhwc(x, cpad):
        B, C, H, W = x.shape
        out = torch.zeros((B, H, W, cpad), dtype=F32, device=x.device)
        out[..., :C] = x.permute(0, 2, 3, 1)
        return out

    # -- public
    def encode(self, x, sample_noise=None):
        P, ae = self.P, self.ae
        x = x.to(device=self.dev, dtype=F32)
        B = x.shape[0]
        e = ae.encoder
        h = self.conv(self.to_nhwc(x, _ceil_to(x.shape[1], 32)), e.conv_in, 1)
        nres = len(P.ch_mult)
        for lvl in range(nres):
            for blk in e.down[lvl].block:
                h = self.res(h, blk)
            if lvl != nres - 1:
                h = self.conv(h, e.down[lvl].downsample.conv, 2)
        h = self.res(h, e.mid.block_1)
        h = self.attn(h, e.mid.attn_1)
        h = self.res(h, e.mid.block_2)
        mom = self.conv(self.gn(h, e.norm_out, True), e.conv_out, 1)          # [B,h,w,2z]
        _, hh, ww, _ = mom.shape
        zc = P.z_channels
        if sample_noise is None:
            sample_noise = torch.randn(B, zc, hh, ww)     # host generator: same draw as the reference's randn_like on CPU
        noise = sample_noise.to(device=self.dev, dtype=F32).permute(0, 2, 3, 1).contiguous()
        z = torch.empty((B, hh, ww, zc), dtype=F32, device=self.dev)
        ops.vae_reparam_f32(mom, noise, z, B * hh * ww, zc, P.scale_factor, P.shift_factor)
        return z.permute(0, 3, 1, 2).contiguous()

    def decode(self, z):
        P, ae = self.P, self.ae
        z = z.to(device=self.dev, dtype=F32).contiguous()
        zz = torch.empty_like(z)
        ops.vae_unscale_f32(z, zz, z.numel(), P.scale_factor, P.shift_factor)
        d = ae.decoder
        h = self.conv(self.to_nhwc(zz, _ceil_to(zz.shape[1], 32)), d.conv_in, 1)
        h = self.res(h, d.mid.block_1)
        h = self.attn(h, d.mid.attn_1)
        h = self.res(h, d.mid.block_2)
        for lvl in reversed(range(len(P.ch_mult))):
            for blk in d.up[lvl].block:
                h = self.res(h, blk)
            if lvl != 0:
                h = self.conv(h, d.up[lvl].upsample.conv, 3)
        h = self.conv(self.gn(h, d.norm_out, True), d.conv_out, 1)            # [B,H,W,3]
        return h.permute(0, 3, 1, 2).contiguous()


BF16 = torch.bfloat16


def _pack3_bf16(w, cin_pad):
    return _pack3(w, cin_pad).to(BF16)


class VaeEngineBf16(VaeEngine):
    """The same plan under the reference inferencer's bf16 autocast (inferencer.py:233 -> decode_image :174-185, and the VAE-encode of an
    edit request inside the same region): NHWC **bf16** activations, every conv / the AttnBlock's two products on the bf16 MFMA
    (``bagel_conv_gemm_bf16``: bf16 weights and bias, fp32 accumulation, bf16 result, residual adds rounded to bf16), GroupNorm + swish in
    fp32 on the bf16 tensor with ONE rounding on the way out (``bagel_groupnorm_bf16`` -- group_norm is on CUDA autocast's fp32 list).
    Cast points = oracle.bagel_oracle VAE_AUTOCAST "cuda"; 16x the fp32 path's matrix rate.  Channel counts are padded to multiples of 8."""

    def _conv_w(self, m):
        w = self._w.get(id(m))
        if w is None:
            wt = m.weight.data
            if wt.shape[-1] == 3:
                cin_pad = _ceil_to(wt.shape[1], 8)
                w = (_pack3_bf16(wt, cin_pad), cin_pad, m.bias.data.to(BF16))
            else:
                w = (wt.reshape(wt.shape[0], wt.shape[1]).to(BF16).contiguous(), wt.shape[1], m.bias.data.to(BF16))
            self._w[id(m)] = w
        return w

    def conv(self, x, m, mode, residual=None, out_hw=None):
        B, H, W, C = x.shape
        w, cin, bias = self._conv_w(m)
        if cin != C:
            raise ValueError(f"conv expects {cin} (padded) input channels, got {C}")
        Ho, Wo = (H, W) if mode in (0, 1) else ((H // 2, W // 2) if mode == 2 else (2 * H, 2 * W))
        cout = w.shape[0]
        ldo = _ceil_to(cout, 8)          # rows stay 16-byte aligned for the next op's chunk loads (conv_out has 3 channels)
        if residual is not None and (ldo != cout or residual.shape[-1] != cout or not residual.is_contiguous()):
            raise ValueError("residual must be a contiguous NHWC tensor with Cout % 8 == 0 channels")
        out = torch.empty((B, Ho, Wo, ldo), dtype=BF16, device=x.device)
        ops.conv_gemm_bf16(x, C, w, w.shape[1], bias, residual, out, ldo, B, H, W, C, Ho, Wo, cout, mode)
        return out if ldo == cout else out[..., :cout]

    def gn(self, x, m, swish):
        B, H, W, C = x.shape
        need = ops.groupnorm_bf16_workspace_floats(B, C, 32)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=F32, device=x.device)
        y = torch.empty_like(x)
        ops.groupnorm_bf16(x, y, self._ws, m.weight.data, m.bias.data, B, H * W, C, 32, 1e-6, swish)
        return y

    def attn(self, x, m):
        B, H, W, C = x.shape
        h = self.gn(x, m.norm, False)
        q, k, v = self.conv(h, m.q, 0), self.conv(h, m.k, 0), self.conv(h, m.v, 0)
        N = H * W
        o = torch.empty((B, N, C), dtype=BF16, device=x.device)
        # scaled_dot_product_attention on bf16 q / k / v: fp32 scores and softmax, bf16 probabilities into the second product (what the
        # library kernels do); the score matrix in blocks of ATTN_ROWS query rows as in the fp32 engine
        R = min(N, self.ATTN_ROWS)
        N8 = _ceil_to(N, 8)                                    # key axis padded to whole 16-byte chunks (zero probabilities x zero V columns)
        sbuf = torch.empty((R, N8), dtype=F32, device=x.device)
        pbuf = torch.zeros((R, N8), dtype=BF16, device=x.device)
        for b in range(B):
            qb, kb = q[b].view(N, C), k[b].view(N, C)
            vt = torch.zeros((C, N8), dtype=BF16, device=x.device)
            vt[:, :N] = v[b].view(N, C).t()                   # layout copy only
            for r0 in range(0, N, R):
                n = min(R, N - r0)
                s, pr = sbuf[:n], pbuf[:n]
                ops.conv_gemm_bf16(qb[r0:r0 + n], qb.stride(0), kb, kb.stride(0), None, None, s, N8, 1, 1, n, C, 1, n, N, 0)
                ops.softmax_rows_bf16(s, pr, n, N, float(C) ** -0.5)
                ob = o[b, r0:r0 + n]
                ops.conv_gemm_bf16(pr, N8, vt, N8, None, None, ob, C, 1, 1, n, N8, 1, n, C, 0)
        return self.conv(o.view(B, H, W, C), m.proj_out, 0, residual=x)

    @staticmethod
    def to_nhwc(x, cpad):
        B, C, H, W = x.shape
        out = torch.zeros((B, H, W, cpad), dtype=BF16, device=x.device)
        out[..., :C] = x.permute(0, 2, 3, 1)                   # the cast autocast puts in front of conv_in
        return out

    def encode(self, x, sample_noise=None):
        P, ae = self.P, self.ae
        x = x.to(device=self.dev, dtype=F32)
        B = x.shape[0]
        e = ae.encoder
        h = self.conv(self.to_nhwc(x, _ceil_to(x.shape[1], 8)), e.conv_in, 1)
        nres = len(P.ch_mult)
        for lvl in range(nres):
            for blk in e.down[lvl].block:
                h = self.res(h, blk)
            if lvl != nres - 1:
                h = self.conv(h, e.down[lvl].downsample.conv, 2)
        h = self.res(h, e.mid.block_1)
        h = self.attn(h, e.mid.attn_1)
        h = self.res(h, e.mid.block_2)
        mom = self.conv(self.gn(h, e.norm_out, True), e.conv_out, 1)          # [B,h,w,2z] bf16
        _, hh, ww, _ = mom.shape
        zc = P.z_channels
        if sample_noise is None:
            sample_noise = torch.randn(B, zc, hh, ww, dtype=BF16)          # randn_like(mean) of the bf16 moments, host generator
        noise = sample_noise.to(device=self.dev, dtype=BF16).permute(0, 2, 3, 1).contiguous()
        z = torch.empty((B, hh, ww, zc), dtype=BF16, device=self.dev)
        ops.vae_reparam_bf16(mom, noise, z, B * hh * ww, zc, P.scale_factor, P.shift_factor)
        return z.permute(0, 3, 1, 2).contiguous()

    def decode(self, z):
        P, ae = self.P, self.ae
        z = z.to(device=self.dev, dtype=F32).contiguous()
        zz = torch.empty_like(z)
        ops.vae_unscale_f32(z, zz, z.numel(), P.scale_factor, P.shift_factor)        # fp32 like the reference (z arrives fp32; conv_in casts)
        d = ae.decoder
        h = self.conv(self.to_nhwc(zz, _ceil_to(zz.shape[1], 8)), d.conv_in, 1)
        h = self.res(h, d.mid.block_1)
        h = self.attn(h, d.mid.attn_1)
        h = self.res(h, d.mid.block_2)
        for lvl in reversed(range(len(P.ch_mult))):
            for blk in d.up[lvl].block:
                h = self.res(h, blk)
            if lvl != 0:
                h = self.conv(h, d.up[lvl].upsample.conv, 3)
        h = self.conv(self.gn(h, d.norm_out, True), d.conv_out, 1)            # [B,H,W,3] bf16 (a view of the 8-channel padded rows)
        return h.permute(0, 3, 1, 2)                                          # CHW VIEW of the NHWC buffer: image_to_u8 reads it strided
