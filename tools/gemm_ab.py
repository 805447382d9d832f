"""Timing of the four denoise GEMMs of the stream-batched forward (M = 2 streams x 4 samples x 4098 rows with the model's MoT row lists and
epilogues, variant 4) with whichever library BAGEL_HIP_LIB names -- run it once per A/B build (tools/ab_build.sh), interleaved.
    BAGEL_HIP_LIB=bagel_amd/libbagel_hip_saddr.so python tools/gemm_ab.py [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bagel_amd import ops  # noqa: E402

DEV, BF16 = "cuda", torch.bfloat16


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    S, L = 8, 4096
    rows_t, rows_v = [], []
    for s in range(S):
        rows_t += [s * (L + 2), s * (L + 2) + L + 1]
        rows_v += list(range(s * (L + 2) + 1, s * (L + 2) + L + 1))
    M = S * (L + 2)
    it = lambda x: torch.tensor(x, dtype=torch.int32, device=DEV)  # noqa: E731
    rt, rv = it(rows_t), it(rows_v)
    g = torch.Generator(device=DEV).manual_seed(0)
    out = []
    for name, N, K, kw in [("qkv", 4608, 3584, dict(bias=True)), ("o", 3584, 3584, dict(res=True)), ("gate_up", 37888, 3584, dict(epi=3)),
                           ("down", 3584, 18944, dict(res=True))]:
        A = torch.randn(M, K, device=DEV, generator=g).to(BF16)
        W1 = (torch.randn(N, K, device=DEV, generator=g) * K ** -0.5).to(BF16)
        b1 = (torch.randn(N, device=DEV, generator=g) * 0.1).to(BF16) if kw.get("bias") else None
        Nout = N // 2 if kw.get("epi") else N
        C = torch.randn(M, Nout, device=DEV, generator=g).to(BF16)

        def call(v):
            # the latent rows only (the marker rows take the dense side path in the product): one row group with gather / scatter lists
            ops.gemm(A, W1, C, bias0=b1, a_rows0=rv, c_rows0=rv, M0=len(rows_v), residual=C if kw.get("res") else None,
                     epilogue=kw.get("epi", 0), variant=v)
        best = {4: 1e9, 5: 1e9}
        for _ in range(rounds):
            for v in (4, 5):                      # interleaved in ONE process: 4 = per-lane 64-bit DMA addresses, 5 = SGPR base + 32-bit offsets
                for _ in range(2):
                    call(v)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(8):
                    call(v)
                e1.record()
                torch.cuda.synchronize()
                best[v] = min(best[v], e0.elapsed_time(e1) / 8)
        fl = 2.0 * len(rows_v) * N * K
        out.append(f"{name} v4 {best[4]:.3f} ms {fl / best[4] / 1e9:5.0f} TF, v5 {best[5]:.3f} ms {fl / best[5] / 1e9:5.0f} TF")
        del A, W1, C
    print(os.path.basename(os.environ.get("BAGEL_HIP_LIB", "libbagel_hip.so")), " | ".join(out), flush=True)


if __name__ == "__main__":
    main()
