"""The GEMMs of one decoder layer's backward at the training-step probe's batch (18 256 tokens: 10 064 rows on the und expert, 8 192 on
the gen expert), each timed on its own (HIP events, 5 launches): routed dX products, the gate/up recompute, per-expert dW products on the
transposed images, and the transposes themselves.  Prints ms and TFLOP/s per call."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bagel_amd import ops  # noqa: E402

BF16, DEV = torch.bfloat16, "cuda"


def timeit(fn, iters=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    H, I, QW, QKV = 3584, 18944, 3584, 4608
    n0, n1 = 10064, 8192
    M = n0 + n1
    g = torch.Generator(device=DEV).manual_seed(0)
    rn = lambda *s: (torch.randn(*s, generator=g, device=DEV) * 0.05).to(BF16)  # noqa: E731
    perm = torch.randperm(M, device=DEV)
    r0, r1 = perm[:n0].sort().values.to(torch.int32), perm[n0:].sort().values.to(torch.int32)
    tot = 0.0

    def routed(name, K, N):
        nonlocal tot
        A, C = rn(M, K), torch.empty(M, N, dtype=BF16, device=DEV)
        W0, W1 = rn(N, K), rn(N, K)
        ms = timeit(lambda: ops.gemm(A, W0, C, a_rows0=r0, c_rows0=r0, M0=n0, W1=W1, a_rows1=r1, c_rows1=r1, M1=n1))
        tot += ms
        print(f"  {name:34s} M {M:6d} N {N:6d} K {K:6d}: {ms:7.3f} ms  {2 * M * N * K / ms / 1e9:7.1f} TFLOP/s")

    def wgrad(name, N, K):
        nonlocal tot
        dY, X = rn(M, N), rn(M, K)
        for rows, n in ((r0, n0), (r1, n1)):
            t_ms = timeit(lambda: (ops.transpose(dY, rows=rows, n=n), ops.transpose(X, rows=rows, n=n)))
            dYt, Xt = ops.transpose(dY, rows=rows, n=n), ops.transpose(X, rows=rows, n=n)
            dW = torch.empty(N, K, dtype=BF16, device=DEV)
            ms = timeit(lambda: ops.gemm(dYt, Xt, dW))
            tot += ms + t_ms
            print(f"  {name:34s} M {N:6d} N {K:6d} K {dYt.shape[1]:6d}: {ms:7.3f} ms  {2 * n * N * K / ms / 1e9:7.1f} TFLOP/s   + transposes {t_ms:6.3f} ms "
                  f"({(n * (N + K) * 4) / t_ms / 1e6:6.1f} GB/s)")

    print("dX products (two-expert routing, transposed weight images):")
    routed("d_act   = g @ Wd", H, I)
    routed("gate/up recompute", H, 2 * I)
    routed("d_h2    = d_gu @ Wgu", 2 * I, H)
    routed("d_att   = g @ Wo", H, QW)
    routed("d_h1    = dqkv @ Wqkv", QKV, H)
    print("dW products (per expert, contraction over that expert's rows):")
    wgrad("dWd   = g^T act", H, I)
    wgrad("dWgu  = d_gu^T h", 2 * I, H)
    wgrad("dWo   = g^T att", H, QW)
    wgrad("dWqkv = dqkv^T h", QKV, H)
    Wt = timeit(lambda: [ops.transpose(w) for w in (rn(H, I), rn(2 * I, H), rn(H, QW), rn(QKV, H))])
    print(f"weight transposes of one expert: {Wt:.3f} ms;  total of the above: {tot:.2f} ms per layer")


if __name__ == "__main__":
    main()
