#!/usr/bin/env python
"""Time Bagel.forward (training forward, losses only) at BAGEL-7B-MoT shapes on one MI355X: a packed batch of two
understanding samples [prompt | 980^2 ViT image | answer with CE] and two generation samples [prompt | noised 1024^2 latent
image with MSE], random-init weights.  Prints one JSON line (tokens/s of the forward pass, ms per forward)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_batch(model, ids, n_und=2, n_gen=2, seed=0):
    from bagel_amd.data.data_utils import get_flattened_position_ids_extrapolate as pos_ids, patchify
    g = torch.Generator().manual_seed(seed)
    b = dict(text_ids=[], text_idx=[], pos=[], vit_idx=[], vae_idx=[], ce_idx=[], labels=[], mse_idx=[], t=[], vit_tok=[], vit_pos=[],
             vit_len=[], lat_pos=[], lat_shapes=[], lat=[], split_lens=[], attn_modes=[], sample_lens=[])
    cur = 0

    def text(n, rope, loss):
        nonlocal cur
        toks = torch.randint(0, 151643, (n,), generator=g).tolist()
        sh = [ids["bos_token_id"]] + toks
        b["text_ids"] += sh + [ids["eos_token_id"]]
        b["text_idx"] += list(range(cur, cur + len(sh) + 1))
        if loss:
            b["ce_idx"] += list(range(cur, cur + len(sh)))
            b["labels"] += toks + [ids["eos_token_id"]]
        cur += len(sh) + 1
        b["pos"] += list(range(rope, rope + len(sh) + 1))
        b["split_lens"].append(len(sh) + 1); b["attn_modes"].append("causal")
        return len(sh) + 1, rope + len(sh) + 1

    for _ in range(n_und):
        tot, rope = 0, 0
        n, rope = text(30, rope, False); tot += n
        img = torch.rand(3, 980, 980, generator=g) * 2 - 1
        b["text_ids"].append(ids["start_of_image"]); b["text_idx"].append(cur); cur += 1
        tk = patchify(img, 14)
        b["vit_idx"] += list(range(cur, cur + tk.shape[0])); cur += tk.shape[0]
        b["vit_tok"].append(tk); b["vit_len"].append(tk.shape[0]); b["vit_pos"].append(pos_ids(980, 980, 14, 70))
        b["text_ids"].append(ids["end_of_image"]); b["text_idx"].append(cur); cur += 1
        b["pos"] += [rope] * (tk.shape[0] + 2); rope += 1
        b["split_lens"].append(tk.shape[0] + 2); b["attn_modes"].append("full"); tot += tk.shape[0] + 2
        n, rope = text(62, rope, True); tot += n
        b["sample_lens"].append(tot)
    for _ in range(n_gen):
        tot, rope = 0, 0
        n, rope = text(30, rope, False); tot += n
        b["text_ids"].append(ids["start_of_image"]); b["text_idx"].append(cur); cur += 1
        b["vae_idx"] += list(range(cur, cur + 4096)); b["mse_idx"] += list(range(cur, cur + 4096)); cur += 4096
        b["t"] += [float(torch.randn(1, generator=g))] * 4096
        b["lat_pos"].append(pos_ids(1024, 1024, 16, 64)); b["lat_shapes"].append((64, 64))
        b["lat"].append(torch.randn(16, 128, 128, generator=g))
        b["text_ids"].append(ids["end_of_image"]); b["text_idx"].append(cur); cur += 1
        b["pos"] += [rope] * 4098
        b["split_lens"].append(4098); b["attn_modes"].append("noise"); tot += 4098
        b["sample_lens"].append(tot)
    return dict(sequence_length=cur, packed_text_ids=torch.tensor(b["text_ids"]), packed_text_indexes=torch.tensor(b["text_idx"]),
                sample_lens=b["sample_lens"], packed_position_ids=torch.tensor(b["pos"]), split_lens=b["split_lens"], attn_modes=b["attn_modes"],
                ce_loss_indexes=torch.tensor(b["ce_idx"]), packed_label_ids=torch.tensor(b["labels"]),
                packed_vit_tokens=torch.cat(b["vit_tok"]), packed_vit_token_indexes=torch.tensor(b["vit_idx"]),
                packed_vit_position_ids=torch.cat(b["vit_pos"]), vit_token_seqlens=torch.tensor(b["vit_len"], dtype=torch.int),
                padded_latent=torch.stack(b["lat"]), patchified_vae_latent_shapes=b["lat_shapes"],
                packed_latent_position_ids=torch.cat(b["lat_pos"]), packed_vae_token_indexes=torch.tensor(b["vae_idx"]),
                packed_timesteps=torch.tensor(b["t"]), mse_loss_indexes=torch.tensor(b["mse_idx"]))


def main():
    from bagel_amd.factory import BAGEL_7B_MOT, NEW_TOKEN_IDS_QWEN25, build_bagel, init_random_
    dev = torch.device("cuda", 0)
    model, _ = build_bagel(BAGEL_7B_MOT, device=dev, with_vae=False)
    init_random_(model, seed=0)
    model.llm2vae.weight.data.normal_(0, 3584 ** -0.5)
    batch = build_batch(model, NEW_TOKEN_IDS_QWEN25)
    noise = torch.randn(len(batch["packed_vae_token_indexes"]), 64, generator=torch.Generator().manual_seed(1)).to(dev)
    out = model(noise=noise, **batch)      # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters = 3
    for _ in range(iters):
        out = model(noise=noise, **batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    n = batch["sequence_length"]
    print(json.dumps({"workload": "Bagel.forward (training forward, losses only), BAGEL-7B-MoT, packed batch: 2 x [prompt | 980^2 ViT image | "
                      "answer+CE] + 2 x [prompt | noised 1024^2 latent image + MSE]", "tokens": n, "ms_per_forward": dt * 1e3,
                      "tokens_per_s": n / dt, "linear_tflops": 13.0506e-3 * n / dt,
                      "ce_mean": float(out["ce"].mean()), "mse_mean": float(out["mse"].mean()),
                      "finite": bool(torch.isfinite(out["ce"]).all() and torch.isfinite(out["mse"]).all())}))


if __name__ == "__main__":
    main()
