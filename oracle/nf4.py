"""CPU restatement of bitsandbytes' NF4 weight quantisation -- the 4-bit load mode the reference ACTUALLY ships
(app.py:114-125: ``BnbQuantizationConfig(load_in_4bit=True, bnb_4bit_compute_dtype=torch.bfloat16, bnb_4bit_use_double_quant=False,
bnb_4bit_quant_type="nf4")`` through accelerate's ``load_and_quantize_model``).  TEST INFRASTRUCTURE ONLY.

bitsandbytes is a third-party CUDA dependency that is NOT under /root/reference (requirements.txt pins no version for it; the
algorithm below is the one of bitsandbytes 0.41-0.45, unchanged across those releases) and cannot be installed here, so this file
restates its PUBLISHED algorithm (Dettmers et al., "QLoRA", 2023, appendix E; ``functional.py::quantize_4bit / dequantize_4bit``,
``csrc/kernels.cu::kQuantizeBlockwise<.., NF4> / dQuantizeNF4 / dDequantizeNF4``) -- parity with the library's own binaries is
therefore **unpinned**; what is pinned are its published constants and the known-answer vectors of tests/test_oracle_golden.py:

    code book    16 fp32 values: the quantiles of N(0, 1) normalised to [-1, 1] with an exact zero (``NF4_CODE`` below)
    blocks       the weight tensor is flattened row-major and cut into blocks of 64 consecutive elements (``blocksize=64``; for a Linear
                 weight [N, K] with K % 64 == 0 a block never crosses a row)
    absmax       per block, fp32:  a = max |w|            (no double quantisation: stored as fp32)
    code         x = w * (1 / a)  in fp32 (a reciprocal and a multiply, as the kernel does), then the nearest code-book entry by the
                 kernel's decision tree: thresholds = the midpoints of adjacent entries, ``x > threshold`` picks the upper one (a value
                 exactly on a midpoint goes DOWN).  An all-zero block has 1 / a = inf, x = NaN, every comparison false -> code 0
                 (de-quantises to -1 * 0 = -0).
    storage      two codes per byte, the EVEN element in the HIGH nibble:  byte = code[2 i] << 4 | code[2 i + 1]
    de-quantise  w' = NF4_CODE[code] * a  in fp32, cast to the compute dtype (bf16)
    compute      y = F.linear(x.to(bf16), w'.to(bf16), bias)  -- the de-quantised bf16 product, fp32 accumulation

accelerate quantises every ``nn.Linear`` except the output head; the product applies the mode where its weight-only decode options
live: the und expert's q/k/v/o/gate/up/down projections of every layer during ``generate_text`` (lm_head stays bf16)."""
import torch

NF4_CODE = torch.tensor([-1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453, -0.28444138169288635,
                         -0.18477343022823334, -0.09105003625154495, 0.0, 0.07958029955625534, 0.16093020141124725,
                         0.24611230194568634, 0.33791524171829224, 0.44070982933044434, 0.5626170039176941, 0.7229568362236023, 1.0],
                        dtype=torch.float32)
# the thresholds of dQuantizeNF4 (csrc/kernels.cu), ascending: THRESH[i] separates code i from code i + 1
NF4_THRESH = torch.tensor([-0.8480964004993439, -0.6106329262256622, -0.4599952697753906, -0.33967943489551544, -0.23460740596055984,
                           -0.13791173323988914, -0.045525018125772476, 0.03979014977812767, 0.1202552504837513, 0.2035212516784668,
                           0.2920137718319893, 0.3893125355243683, 0.5016634166240692, 0.6427869200706482, 0.8614784181118011],
                          dtype=torch.float32)
BLOCK = 64


def quantize_nf4(w):
    """bf16 / fp32 [N, K] (K % 64 == 0) -> (packed uint8 [N, K / 2], absmax fp32 [N, K / 64])."""
    wf = w.float()
    N, K = wf.shape
    assert K % BLOCK == 0, "restated for Linear weights whose rows are whole blocks"
    blk = wf.view(N, K // BLOCK, BLOCK)
    amax = blk.abs().amax(dim=2)
    inv = torch.tensor(1.0, dtype=torch.float32) / amax               # inf for an all-zero block
    x = blk * inv[:, :, None]                                         # NaN there: every comparison below is False -> code 0
    code = (x[..., None] > NF4_THRESH).sum(dim=-1).to(torch.uint8)    # number of thresholds below x == the decision tree's leaf
    code = code.view(N, K)
    packed = (code[:, 0::2] << 4) | code[:, 1::2]
    return packed.contiguous(), amax.contiguous()


def codes_of(packed):
    N, Kh = packed.shape
    return torch.stack([(packed >> 4).long(), (packed & 0xF).long()], dim=2).view(N, 2 * Kh)


def dequantize_nf4(packed, absmax, dtype=torch.bfloat16):
    """-> [N, K] in ``dtype``: NF4_CODE[code] * absmax in fp32, then ONE cast (dequantize_4bit with a bf16 output)."""
    code = codes_of(packed)
    N, K = code.shape
    w = NF4_CODE[code].view(N, K // BLOCK, BLOCK) * absmax[:, :, None]
    return w.view(N, K).to(dtype)


def linear_nf4(x, packed, absmax, bias=None):
    """The reference's Linear4bit forward with bf16 compute: F.linear on the de-quantised bf16 weight."""
    w = dequantize_nf4(packed, absmax, torch.bfloat16)
    return torch.nn.functional.linear(x.to(torch.bfloat16), w, None if bias is None else bias.to(torch.bfloat16))
