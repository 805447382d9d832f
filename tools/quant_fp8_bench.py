"""bagel_quantize_rows_fp8 at the activation shapes of the FP8 denoise path (M = 32 768 gen rows: the attention output, 3584 columns, and the SwiGLU output,
18944 columns): microseconds per call and TB/s of its own 3 bytes per element.  BAGEL_FP8_QUANT_TWO_PASS=1 selects the two-pass kernel (A/B)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bagel_amd import ops  # noqa: E402

for cols in (3584, 18944):
    xs = [torch.randn(32768, cols, device="cuda", dtype=torch.bfloat16) for _ in range(3)]      # cycled: nothing cache-resident
    q = torch.empty(32768, cols, device="cuda", dtype=torch.uint8)
    sc = torch.empty(32768, device="cuda", dtype=torch.float32)
    for x in xs:
        ops.quantize_rows_fp8(x, q, sc)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    n = 30
    for i in range(n):
        ops.quantize_rows_fp8(xs[i % 3], q, sc)
    ev[1].record()
    torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) * 1e3 / n
    print(f"quantize_rows_fp8 32768 x {cols}: {us:8.1f} us = {32768 * cols * 3 / us / 1e6:5.2f} TB/s   (two_pass={os.environ.get('BAGEL_FP8_QUANT_TWO_PASS', '0')})")
