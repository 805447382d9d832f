#!/usr/bin/env python
"""How far the reference's OWN gradients move when its bf16 linears accumulate in another order: the oracle's autograd with
LINEAR_FP32_ACCUM (same bf16 operands and rounding points, fp32 matmul + one rounding = what another CPU backend, or a GPU, does) against
the default run, per parameter tensor (rel-L2).  The parity tolerances of the training backward are set against this floor.
CPU only:  python tools/train_grad_noise_floor.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import bagel_oracle as O  # noqa: E402
from oracle.configs import TINY, TINY_D128, WIDE7B  # noqa: E402
from oracle.shapes import bagel_shapes  # noqa: E402
from oracle.weights import synth_state_dict  # noqa: E402
from tests.util_models import pack_training_batch  # noqa: E402


def weights(cfg):
    W = {k: v.to(torch.bfloat16) for k, v in synth_state_dict(bagel_shapes(cfg), 0).items()}
    H = cfg["llm"]["hidden_size"]
    W["latent_pos_embed.pos_embed"] = O.sincos_2d_table(H, cfg["bagel"]["max_latent_size"]).to(torch.bfloat16)
    W["vit_pos_embed.pos_embed"] = O.sincos_2d_table(H, cfg["bagel"]["vit_max_num_patch_per_side"]).to(torch.bfloat16)
    return W


def main():
    for cfg, samples, seed in ((TINY, None, 0), (TINY_D128, None, 0),
                               (WIDE7B, [[("text", 6, False), ("vit", 28, 42), ("text", 9, True)],
                                         [("text", 5, False), ("vae", 64, 64, False), ("vae", 320, 256, True), ("text", 3, True), ("vae", 64, 48, True)]], 11)):
        W = weights(cfg)
        if samples is None:
            fx = torch.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", f"{cfg['name']}_train.pt"), weights_only=False)
            batch, noise = fx["batch"], fx["noise"]
        else:
            batch, noise, _, _ = pack_training_batch(cfg, samples, seed)
        w = torch.rand(batch["ce_loss_indexes"].numel(), generator=torch.Generator().manual_seed(5)) + 0.5
        names = {k for k in W if not k.startswith(("vit_pos_embed.", "latent_pos_embed.")) and W[k].is_floating_point() and "inv_freq" not in k and "rope." not in k}
        _, g0, _ = O.training_step_grads(W, cfg, batch, noise, w, names=names)
        O.LINEAR_FP32_ACCUM = True
        try:
            _, g1, _ = O.training_step_grads(W, cfg, batch, noise, w, names=names)
        finally:
            O.LINEAR_FP32_ACCUM = False
        rows = []
        for k in g0:
            n0 = float(g0[k].float().norm())
            if n0 == 0 or (k.startswith("vit_model.") and k.endswith("k_proj.bias")):
                continue
            rows.append((float((g1[k].float() - g0[k].float()).norm()) / n0, k))
        rows.sort(reverse=True)
        med = rows[len(rows) // 2][0]
        print(f"[{cfg['name']}] {len(rows)} gradient tensors: worst {rows[0][0]:.3e} ({rows[0][1]}), 2nd {rows[1][0]:.3e}, median {med:.3e}")


if __name__ == "__main__":
    main()
