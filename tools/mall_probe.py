"""Does a weight panel that is already on-die (Infinity Cache, 256 MiB; or an XCD's L2) stream faster into the decode GEMV than
one that comes from HBM?  Decides whether warming the next projection's weights during the latency-bound kernels of a decode
layer (qkv epilogue, split attention, combine: ~22 us of idle HBM per layer) can pay.

    python tools/mall_probe.py            # on the GPU box; BAGEL_HIP_LIB selects an A/B build (plain vs non-temporal loads)

For each projection shape of the 7B decode layer:  cold = the weights were evicted by a 1.5 GB sweep,  warm = a read pass over the
weights (torch reduction) immediately before,  hot = the same GEMV back to back (its own previous pass left the lines)."""
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bagel_amd import ops  # noqa: E402

DEV = "cuda"
BF16 = torch.bfloat16


def timed(fn, before=None, reps=12):
    ts = []
    for _ in range(reps):
        if before is not None:
            before()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    torch.manual_seed(0)
    sweep = torch.empty(1536 << 20, dtype=torch.uint8, device=DEV)
    shapes = [("o", 3584, 3584, 0), ("qkv", 4608, 3584, 0), ("down", 3584, 18944, 0), ("100MB", 14336, 3584, 0), ("gate_up", 37888, 3584, 3)]
    print(f"lib: {os.environ.get('BAGEL_HIP_LIB', 'default')}")
    for name, N, K, epi in shapes:
        W = (torch.randn(N, K, device=DEV) * K ** -0.5).to(BF16)
        x = torch.randn(1, K, device=DEV).to(BF16)
        C = torch.empty(1, N // 2 if epi == 3 else N, dtype=BF16, device=DEV)
        mb = W.numel() * 2 / 1e6
        run = lambda: ops.gemv(x, W, C, epilogue=epi)   # noqa: E731
        evict = lambda: sweep.add_(1)                    # noqa: E731
        wi = W.view(torch.int32)

        def warm():
            evict()
            wi.sum()

        half = wi[: N // 2]

        def warm_half():
            evict()
            half.sum()

        run(); torch.cuda.synchronize()
        t_cold = timed(run, evict)
        t_warm = timed(run, warm)
        t_half = timed(run, warm_half)
        t_hot = timed(run, run)
        f = lambda t: f"{t:7.1f} us = {mb / t:5.2f} TB/s"   # noqa: E731
        print(f"{name:8s} {mb:6.1f} MB  cold {f(t_cold)} | warmed by a read pass {f(t_warm)} | half warmed {f(t_half)} | back to back {f(t_hot)}")


if __name__ == "__main__":
    main()
