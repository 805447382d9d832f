"""CPU restatement of the FP8 (OCP e4m3) option of the gen-expert GEMMs -- TEST INFRASTRUCTURE ONLY.

There is no reference implementation to pin against: the reference's quantised modes (app.py:114-131) are bitsandbytes NF4 / LLM.int8 on
CUDA, un-vendored; what the row in SURVEY.md 8f.4 asks for on MI355X is the chip's own low-precision MFMA path.  This file states the
scheme the kernels implement so the tests can check them bit for bit (quantiser) and to fp32-accumulation accuracy (GEMM):

    scale[r]  = max_k |x[r, k]| / 448                    (448 = largest finite e4m3fn value)
    q[r, k]   = e4m3fn(round-to-nearest-even(x[r, k] * (1 / scale[r])))
    C[m, n]   = epilogue( (sa[m] * sw[n]) * sum_k q_a[m, k] * q_w[n, k] )        fp32 accumulation, epilogues of the bf16 GEMM

DELAYED scaling of the SwiGLU output (round 6; the down projection's input inside a denoise loop -- ``DelayedScales`` below): from the second forward of a stream
on, the gate/up GEMM's epilogue writes the e4m3 bytes itself, so a row's scale has to exist before its values do:

    scale_t[r] = margin * amax_{t-1}[r] / 448     (margin = 2; 1.0 where amax_{t-1}[r] = 0),    amax_t[r] = max_k |x_t[r, k]|
    q_t[r, k]  = e4m3fn(rne(clamp(x_t[r, k] * (1 / scale_t[r]), -448, 448)))

with amax_{t-1} the row maxima of the SAME rows in the previous forward of the same stream; the first forward (no history) uses the exact row-wise scale above and
leaves amax_0 = (max|row| / 448) * 448 (what the product reconstructs from the exact scale).
"""
import torch


def quantize_rows_fp8(x):
    """bf16 / fp32 [rows, K] -> (uint8 view of float8_e4m3fn [rows, K], fp32 scale [rows]) exactly as bagel_quantize_rows_fp8:
    fp32 reciprocal, fp32 product, round-to-nearest-even conversion."""
    xf = x.float()
    amax = xf.abs().amax(dim=1)
    scale = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    inv = 1.0 / scale
    q = (xf * inv[:, None]).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), scale


def dequant(q_u8, scale):
    return q_u8.view(torch.float8_e4m3fn).float() * scale[:, None]


def gemm_fp8(qa, sa, qw, sw, bias=None, residual=None, swiglu=False):
    """fp32 product of the de-quantised operands with the epilogue roundings of the bf16 GEMM (tests/test_ops_gpu.py::ref_gemm)."""
    acc = (qa.view(torch.float8_e4m3fn).float() @ qw.view(torch.float8_e4m3fn).float().t()) * (sa[:, None] * sw[None, :])
    if swiglu:
        N = qw.shape[0]
        a = acc.view(acc.shape[0], N // 32, 2, 16)
        g = a[:, :, 0].reshape(acc.shape[0], -1).to(torch.bfloat16)
        u = a[:, :, 1].reshape(acc.shape[0], -1).to(torch.bfloat16)
        return torch.nn.functional.silu(g) * u
    if bias is not None:
        acc = acc + bias.float()
    c = acc.to(torch.bfloat16)
    if residual is not None:
        c = residual + c
    return c


class DelayedScales:
    """Restates bagel_amd.modeling.bagel.qwen2_navit.Fp8DelayedScales + the fp8-output epilogue of bagel_gemm_fp8_swiglu_q8 + bagel_fp8_delayed_scales for the
    oracle's sequential forwards: the history of a (weight, call index within the Euler step) pair is the history of one forward stream's rows at one layer.
    ``ptrs``: data_ptr()s of the weights whose INPUT is quantised this way (the gen expert's down projections); ``begin_step()`` once per Euler step."""

    def __init__(self, ptrs, margin=2.0):
        self.ptrs, self.margin = set(ptrs), float(margin)
        self.hist, self.calls = {}, {}

    def begin_step(self):
        self.calls = {}

    def quantize(self, key, x):
        i = self.calls.get(key, 0)
        self.calls[key] = i + 1
        prev = self.hist.get((key, i))
        xf = x.float()
        if prev is None:                                   # no history: the exact scheme; the product keeps scale * 448 as the rows' maxima
            q, scale = quantize_rows_fp8(x)
            self.hist[(key, i)] = scale * torch.tensor(448.0)
            return q, scale
        k = torch.tensor(self.margin, dtype=torch.float32) / torch.tensor(448.0, dtype=torch.float32)
        scale = torch.where(prev > 0, prev * k, torch.ones_like(prev))
        inv = 1.0 / scale
        q = (xf * inv[:, None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
        self.hist[(key, i)] = xf.abs().amax(dim=1)
        return q.view(torch.uint8), scale


def down_proj_gen_ptrs(W):
    """The weights whose input the product quantises with DELAYED scales: the gen expert's down projections."""
    return {v.data_ptr() for k, v in W.items() if "language_model.model.layers." in k and k.endswith("mlp_moe_gen.down_proj.weight")}
