"""GPU parity of the MXFP4 weight-only option of the decode projections against its CPU restatement (oracle/mxfp4.py): the weight
quantiser bit for bit (codes and E8M0 block scales, device scale order), the block-scaled MFMA projection -- FP8 activation quantiser
and optional RMSNorm fused into its prologue -- to fp32-accumulation accuracy for every epilogue and shape of the decode layer."""
import pytest
import torch

from oracle import mxfp4 as MX
from tests.test_ops_gpu import BF16, DEV, close, ops

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,cols", [(5, 128), (37, 640), (300, 3584), (33, 18944)])
def test_quantize_rows_mxfp4_bit_exact(rows, cols):
    g = torch.Generator().manual_seed(rows)
    w = (torch.randn(rows, cols, generator=g) * torch.rand(rows, 1, generator=g) * 0.2).to(BF16)
    w[0, :32] = 0                                                    # an all-zero block: scale byte 0, codes 0
    w[1, 32:64] = torch.tensor([0, .5, 1, 1.5, 2, 3, 4, 6, -.5, -6, .25, .75, 1.25, 1.75, 2.5, 3.5, 5, 7, -7] + [0] * 13).to(BF16) * 0.125
    w[2, :32] = 3.0e38                                               # the largest exponents saturate cleanly
    q, s = ops().quantize_rows_mxfp4(w.to(DEV))
    qr, sr = MX.quantize_mxfp4(w)
    assert torch.equal(s.cpu(), MX.permute_scales(sr)), "block scales differ"
    assert torch.equal(q.cpu(), qr), f"{(q.cpu() != qr).sum().item()} of {qr.numel()} code bytes differ"


@pytest.mark.parametrize("mode", ["plain", "bias", "residual", "swiglu", "norm", "norm_swiglu"])
@pytest.mark.parametrize("M,N,K", [(1, 64, 128), (1, 4608, 3584), (2, 3584, 3584), (1, 3584, 18944), (1, 37888, 3584), (4, 512, 1024)])
def test_gemv_w4_matches_restatement(M, N, K, mode):
    if N == 37888 and mode not in ("swiglu", "norm_swiglu"):
        pytest.skip("the gate+up shape only carries the SwiGLU epilogues in the model (and its CPU restatement takes 10 s)")
    g = torch.Generator().manual_seed(N + K + M)
    x = (torch.randn(M, K, generator=g) * 1.5).to(BF16)
    W = (torch.randn(N, K, generator=g) * K ** -0.5).to(BF16)
    o = ops()
    q, s = o.quantize_rows_mxfp4(W.to(DEV))
    qr, sr = MX.quantize_mxfp4(W)
    assert torch.equal(s.cpu(), MX.permute_scales(sr))
    swiglu = mode.endswith("swiglu")
    bias = (torch.randn(N, generator=g) * 0.1).to(BF16) if mode == "bias" else None
    nw = (1.0 + 0.1 * torch.randn(K, generator=g)).to(BF16) if mode.startswith("norm") else None
    Nout = N // 2 if swiglu else N
    R = torch.randn(M, Nout, generator=g).to(BF16) if mode == "residual" else None
    C = R.to(DEV).clone() if R is not None else torch.full((M, Nout), float("nan"), dtype=BF16, device=DEV)
    o.gemv_w4(x.to(DEV), q, s, C, bias=None if bias is None else bias.to(DEV), residual=C if R is not None else None,
              epilogue=o.EPI_SWIGLU16 if swiglu else o.EPI_NONE, norm_w=None if nw is None else nw.to(DEV), eps=1e-6)
    torch.cuda.synchronize()
    ref = MX.gemv_w4(x, q.cpu(), sr, bias=bias, residual=R, swiglu=swiglu, norm_w=nw, eps=1e-6)
    close(C.cpu(), ref, ulps=2, what=f"gemv_w4 {mode} M={M} N={N} K={K}")
    if mode == "plain":                                              # what 4-bit weights cost against the bf16 product: ~10 %
        full = x.float() @ W.float().t()
        e = ((C.cpu().float() - full).norm() / full.norm()).item()
        assert e < 0.2, e
