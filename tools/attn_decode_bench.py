"""Stand-alone timing of the paged decode attention (csrc/decode.hip: attn_decode_kernel + combine) at the 7B shapes: B requests on
ctx-token contexts, 28 / 4 heads of 128, over L distinct layers' pools (so no launch finds its pages in the Infinity Cache).
BAGEL_DEC_CPW (chunks per workgroup) is read once per process: run once per value.     python tools/attn_decode_bench.py [B=16] [ctx=4936]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bagel_amd import ops  # noqa: E402
from bagel_amd.modeling.bagel.decode import PagedKVCache  # noqa: E402

DEV, BF16 = "cuda", torch.bfloat16


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 4936
    nq, nkv, D, L = 28, 4, 128, 8
    width = nkv * D
    pg = PagedKVCache(L, B, width, ctx + 8, DEV)
    for li in range(L):
        pg.k[li].normal_(); pg.v[li].normal_()
    pg.kv_len.fill_(ctx)
    qkv = torch.randn(B, (nq + 2 * nkv) * D, device=DEV).to(BF16)
    pos = torch.full((B,), ctx, dtype=torch.long, device=DEV)
    inv = (1.0 / (1e6 ** (torch.arange(0, D, 2).float() / D))).to(DEV)
    cos, sin = ops.rope_table(pos, inv)
    qw = torch.ones(D, device=DEV, dtype=BF16)
    po, pml = ops.attn_decode_workspace(B, nq, D, ctx + 8, DEV)
    out = torch.empty(B, nq * D, device=DEV, dtype=BF16)

    def call(i):
        li = i % L
        ops.attn_decode_fused(qkv, cos, sin, qw, qw, pg.k[li], pg.v[li], pg.block_table, pg.kv_len, ctx + 1, po, pml, out, B, nq, nkv, D, D, 1e-6, True,
                              D ** -0.5)
    for i in range(4):
        call(i)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(24):
            call(i)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 24 * 1e3)
    mb = B * ctx * width * 2 * 2 / 1e6
    print(f"B={B} ctx={ctx} cpw={os.environ.get('BAGEL_DEC_CPW', 'auto')}: {best:.1f} us per layer (attention + combine), {mb:.0f} MB of KV = {mb / best:.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
