#!/usr/bin/env python
"""Time one training step -- Bagel.forward with a tape + loss.backward() through the hand-written reverse (bagel_amd/modeling/bagel/
train_step.py) -- at BAGEL-7B-MoT shapes and full depth on one MI355X: the packed batch of tools/train_forward_probe.py (two understanding
samples [prompt | 980^2 ViT image | answer with CE] + two generation samples [prompt | noised 1024^2 latent image with MSE], 18.3k tokens),
random-init bf16 weights, every language-model / connector / head parameter trainable (SigLIP tower frozen; PROBE_VIT=1 trains it too).  No optimizer step (out of scope:
the optimizer is torch's).  Prints one JSON line."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.train_forward_probe import build_batch  # noqa: E402


def main():
    from bagel_amd.factory import BAGEL_7B_MOT, NEW_TOKEN_IDS_QWEN25, build_bagel, init_random_
    layers = int(os.environ.get("PROBE_LAYERS", "0"))
    cfg = BAGEL_7B_MOT if not layers else dict(BAGEL_7B_MOT, llm=dict(BAGEL_7B_MOT["llm"], num_hidden_layers=layers))
    dev = torch.device("cuda", 0)
    model, _ = build_bagel(cfg, device=dev, with_vae=False)
    init_random_(model, seed=0)
    model.llm2vae.weight.data.normal_(0, 3584 ** -0.5)
    frozen = ("vit_pos_embed.", "latent_pos_embed.") if os.environ.get("PROBE_VIT") else ("vit_model.", "vit_pos_embed.", "latent_pos_embed.")
    n_train = 0
    for n, p in model.named_parameters():
        p.requires_grad_(not n.startswith(frozen))
        n_train += p.numel() if p.requires_grad else 0
    scale = int(os.environ.get("PROBE_SCALE", "1"))          # 2: 36.5 k tokens, the reference's max_num_tokens class (pretrain_unified_navit.py)
    batch = build_batch(model, NEW_TOKEN_IDS_QWEN25, n_und=2 * scale, n_gen=2 * scale)
    noise = torch.randn(len(batch["packed_vae_token_indexes"]), 64, generator=torch.Generator().manual_seed(1)).to(dev)

    def step():
        for p in model.parameters():
            p.grad = None
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = model(noise=noise, **batch)
        loss = out["ce"].mean() + out["mse"].mean()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        loss.backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        return float(loss), t1 - t0, t2 - t1

    step()                                   # warm-up (packs the weights, sizes the allocator)
    iters = int(os.environ.get("PROBE_ITERS", "2"))
    rs = [step() for _ in range(iters)]
    fwd, bwd = sum(r[1] for r in rs) / iters, sum(r[2] for r in rs) / iters
    n = batch["sequence_length"]
    L = cfg["llm"]["num_hidden_layers"]
    lin = 13.0506e-3 * n * L / 28            # TFLOP of the decoder's linears in one forward (tools/train_forward_probe.py)
    gn = sum(float(p.grad.float().norm()) ** 2 for p in model.parameters() if p.grad is not None) ** 0.5
    print(json.dumps({"workload": "training step (forward with tape + backward), BAGEL-7B-MoT width, %d layers, packed batch of %d tokens" % (L, n),
                      "tokens": n, "trainable_params": n_train, "ms_forward": fwd * 1e3, "ms_backward": bwd * 1e3,
                      "tokens_per_s": n / (fwd + bwd), "linear_tflops_fwd": lin / fwd, "linear_tflops_bwd_2x_plus_recompute": (2 * lin + lin * 0.68) / bwd,
                      "loss": rs[-1][0], "grad_norm": gn, "finite": bool(gn == gn and gn < float("inf")),
                      "max_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30}))


if __name__ == "__main__":
    main()
