"""Shared helpers for tests: synthetic weights for the oracle (no reference tree needed)."""
import functools

import torch

from oracle import bagel_oracle as O
from oracle.shapes import bagel_shapes, vae_shapes
from oracle.weights import synth_state_dict

WEIGHT_SEED = 0


@functools.lru_cache(maxsize=4)
def _weights(name):
    from oracle import configs
    cfg = {"tiny": configs.TINY, "tiny_d128": configs.TINY_D128, "tiny_rope": configs.TINY_ROPE, "tiny_dense": configs.TINY_DENSE, "tiny_moe": configs.TINY_MOE}[name]
    W = {k: v.to(torch.bfloat16) for k, v in synth_state_dict(bagel_shapes(cfg), WEIGHT_SEED).items()}
    if cfg["vit"].get("rope", False):
        v = cfg["vit"]
        side = v["image_size"] // v["patch_size"]
        tabs = O.rope2d_tables(v["hidden_size"] // v["num_attention_heads"] // 2, side, side)
        for n, t in zip(("cos_h", "sin_h", "cos_w", "sin_w"), tabs):
            W["vit_model.vision_model.rope." + n] = t.to(torch.bfloat16)
    H = cfg["llm"]["hidden_size"]
    W["latent_pos_embed.pos_embed"] = O.sincos_2d_table(H, cfg["bagel"]["max_latent_size"]).to(torch.bfloat16)
    W["vit_pos_embed.pos_embed"] = O.sincos_2d_table(H, cfg["bagel"]["vit_max_num_patch_per_side"]).to(torch.bfloat16)
    VW = synth_state_dict(vae_shapes(cfg["vae"]), WEIGHT_SEED)
    return W, VW


def oracle_weights(cfg):
    """(W, VW): bf16 Bagel weights and fp32 VAE weights, as the golden generator built them."""
    return _weights(cfg["name"])


@functools.lru_cache(maxsize=2)
def _product(name):
    from oracle import configs
    from bagel_amd.factory import build_bagel
    cfg = {"tiny": configs.TINY, "tiny_d128": configs.TINY_D128, "tiny_rope": configs.TINY_ROPE, "tiny_dense": configs.TINY_DENSE, "tiny_moe": configs.TINY_MOE}[name]
    W, VW = _weights(name)
    model, vae = build_bagel(cfg, device="cuda")
    missing, unexpected = model.load_state_dict(W, strict=True), None
    vae.load_state_dict(VW, strict=True)
    return model, vae


def product_model(cfg):
    """(Bagel, AutoEncoder) on cuda:0 carrying exactly the oracle's synthetic weights."""
    return _product(cfg["name"])
