"""Time bagel_attn_bwd_blockmask_bf16 at the attention shapes of the training-step probe (tools/train_step_probe.py): two understanding
samples [32 causal | 4902 full | 64 causal] and two generation samples [32 causal | 4098 noise], 28 / 4 heads of 128.  Prints ms and
TFLOP/s against the reverse's own 16 D FLOPs per visible (query, key) pair and head (and against the 10 D of a fused flash backward).
BAGEL_ABWD_ONLY=dq|dkv (ablation builds, tools/ab_attn_bwd.sh) times one of the two kernels."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bagel_amd import ops  # noqa: E402
from bagel_amd.modeling.bagel.train_step import AttnBackwardPlan  # noqa: E402

BF16, DEV = torch.bfloat16, "cuda"


def main():
    nq, nkv, D = 28, 4, 128
    samples = [([32, 4902, 64], ["causal", "full", "causal"])] * 2 + [([32, 4098], ["causal", "noise"])] * 2
    lens = [sum(s[0]) for s in samples]
    M = sum(lens)
    pairs = 0
    for sl, modes in samples:
        seen = 0
        for L, mode in zip(sl, modes):
            pairs += L * seen + (L * (L + 1) // 2 if mode == "causal" else L * L)
            seen += 0 if mode == "noise" else L
    g = torch.Generator(device=DEV).manual_seed(0)
    rn = lambda *s: torch.randn(*s, generator=g, device=DEV).to(BF16)  # noqa: E731
    q, k, v, o, do = rn(M, nq * D), rn(M, nkv * D), rn(M, nkv * D), rn(M, nq * D), rn(M, nq * D)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    bp = AttnBackwardPlan(DEV, lens, samples)
    lse = torch.full((nq, M), 8.0, dtype=torch.float32, device=DEV) if os.environ.get("PROBE_LSE") else None     # timing only: any finite value
    fn = lambda: ops.attn_bwd_blockmask(q, k, v, o, do, dq, dk, dv, bp.q_items, bp.k_items, bp.noise_bits, nq, nkv, D, D ** -0.5, lse=lse)  # noqa: E731
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 5
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    fl = pairs * nq * D
    print(f"{os.environ.get('BAGEL_HIP_LIB', 'product').split('_')[-1]:>14} only={os.environ.get('BAGEL_ABWD_ONLY', 'both'):>4}: {ms:8.3f} ms  "
          f"(incl. 3 transposes)  {16 * fl / ms / 1e9:7.1f} TFLOP/s of 16 D   {10 * fl / ms / 1e9:7.1f} of 10 D   pairs {pairs / 1e6:.1f} M")


if __name__ == "__main__":
    main()
