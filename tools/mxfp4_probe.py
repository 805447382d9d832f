"""Hardware probe for the operand layout of v_mfma_scale_f32_16x16x128_f8f6f4 with an FP4 A operand as bagel_gemv_w4_bf16 feeds it:
which weight element k' does the instruction pair with activation element j, and which scale byte applies to it?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bagel_amd import ops
DEV, BF16 = "cuda", torch.bfloat16
K, N = 512, 16
ng = ((K // 128 + 3) // 4) * 16

def run(codes, scales, x):
    C = torch.zeros((1, N), dtype=BF16, device=DEV)
    ops.gemv_w4(x.to(DEV), codes.to(DEV), scales.to(DEV), C)
    torch.cuda.synchronize()
    return C.cpu().float()[0]

# 1. all ones
codes = torch.full((N, K // 2), 0x22, dtype=torch.uint8)
scales = torch.full((N, ng), 127, dtype=torch.uint8)
x = torch.ones(1, K).to(BF16)
print("all-ones (expect 512):", run(codes, scales, x)[:4].tolist())
# 2. mapping: row r holds 1.0 where bit r of k is set (r = 0..8), one-hot activation at j
code_el = torch.zeros((N, K), dtype=torch.uint8)
for r in range(9):
    code_el[r] = ((torch.arange(K) >> r) & 1).to(torch.uint8) * 2          # code 2 = 1.0
packed = code_el[:, 0::2] | (code_el[:, 1::2] << 4)
pairs = []
for j in list(range(0, 40)) + [63, 64, 65, 127, 128, 129, 255, 256, 300, 511]:
    x = torch.zeros(1, K); x[0, j] = 1.0
    y = run(packed, scales, x.to(BF16))
    k = sum((1 << r) for r in range(9) if y[r] > 0.5)
    pairs.append((j, k, round(float(y[:9].max()), 3)))
print("activation j -> weight k (value):", pairs)
# 3. scales: double the scale of device byte i, all-ones operands, see which k-block gains
for i in range(ng):
    sc = scales.clone(); sc[:, i] = 128
    # one-hot blocks: activation ones only in block b
    gains = []
    for b in range(K // 32):
        x = torch.zeros(1, K); x[0, b * 32:(b + 1) * 32] = 1.0
        y = run(codes, sc, x.to(BF16))
        if abs(float(y[0]) - 32.0) > 1: gains.append((b, float(y[0])))
    print(f"scale byte {i} = 2^1 affects blocks:", gains)
