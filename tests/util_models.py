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
    cfg = {"tiny": configs.TINY, "tiny_d128": configs.TINY_D128, "tiny_rope": configs.TINY_ROPE, "tiny_dense": configs.TINY_DENSE, "tiny_moe": configs.TINY_MOE,
           "wide7b": configs.WIDE7B}[name]
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
    cfg = {"tiny": configs.TINY, "tiny_d128": configs.TINY_D128, "tiny_rope": configs.TINY_ROPE, "tiny_dense": configs.TINY_DENSE, "tiny_moe": configs.TINY_MOE,
           "wide7b": configs.WIDE7B}[name]
    W, VW = _weights(name)
    model, vae = build_bagel(cfg, device="cuda")
    missing, unexpected = model.load_state_dict(W, strict=True), None
    vae.load_state_dict(VW, strict=True)
    return model, vae


def product_model(cfg):
    """(Bagel, AutoEncoder) on cuda:0 carrying exactly the oracle's synthetic weights."""
    return _product(cfg["name"])


def pack_training_batch(cfg, samples, seed):
    """Hand-pack a training batch the way data/dataset_base.py:306-476 does (the recipe of oracle/make_golden.py::scenario_train,
    parameterised): ``samples`` = list of samples, each a list of splits ("text", n_tokens, with_ce_loss) | ("vit", H, W) |
    ("vae", H, W, with_mse_loss).  Attention modes follow the dataset code: text causal, ViT image full, clean VAE image full,
    noised VAE image noise.  Returns (batch dict for Bagel.forward, noise, split_lens, attn_modes)."""
    from oracle import packers as P
    from oracle.configs import NEW_TOKEN_IDS_TINY as ids
    g = torch.Generator().manual_seed(seed)
    V = cfg["llm"]["vocab_size"]
    ps = cfg["vit"]["patch_size"]
    ds = cfg["vae"]["downsample"] * cfg["bagel"]["latent_patch_size"]
    text_ids, text_idx, pos, vit_idx, vae_idx, ce_idx, labels, mse_idx, timesteps = [], [], [], [], [], [], [], [], []
    vit_tokens, vit_pos, vit_lens, lat_pos, lat_shapes, latents = [], [], [], [], [], []
    sample_lens, masks, all_splits, all_modes = [], [], [], []
    curr = 0
    for sample in samples:
        splits, modes, rope = [], [], 0
        for sp in sample:
            if sp[0] == "text":
                _, n_tok, loss = sp
                tokens = torch.randint(8, V, (n_tok,), generator=g).tolist()
                shifted = [ids["bos_token_id"]] + tokens
                text_ids.extend(shifted)
                text_idx.extend(range(curr, curr + len(shifted)))
                if loss:
                    ce_idx.extend(range(curr, curr + len(shifted)))
                    labels.extend(tokens + [ids["eos_token_id"]])
                curr += len(shifted)
                text_ids.append(ids["eos_token_id"]); text_idx.append(curr); curr += 1
                n = len(shifted) + 1
                pos.extend(range(rope, rope + n))
                rope += n
                mode = "causal"
            elif sp[0] == "vit":
                _, H, Wd = sp
                img = torch.rand(3, H, Wd, generator=g) * 2 - 1
                text_ids.append(ids["start_of_image"]); text_idx.append(curr); curr += 1
                toks = P.patchify(img, ps)
                vit_idx.extend(range(curr, curr + toks.shape[0])); curr += toks.shape[0]
                vit_tokens.append(toks); vit_lens.append(toks.shape[0])
                vit_pos.append(P.position_ids_extrapolate(H, Wd, ps, cfg["bagel"]["vit_max_num_patch_per_side"]))
                text_ids.append(ids["end_of_image"]); text_idx.append(curr); curr += 1
                n = toks.shape[0] + 2
                pos.extend([rope] * n)
                rope += 1
                mode = "full"
            else:
                _, H, Wd, loss = sp
                h, w = H // ds, Wd // ds
                text_ids.append(ids["start_of_image"]); text_idx.append(curr); curr += 1
                n_img = h * w
                vae_idx.extend(range(curr, curr + n_img))
                if loss:
                    mse_idx.extend(range(curr, curr + n_img))
                    t = float(torch.randn(1, generator=g))
                else:
                    t = float("-inf")
                timesteps.extend([t] * n_img)
                curr += n_img
                lat_pos.append(P.position_ids_extrapolate(H, Wd, ds, cfg["bagel"]["max_latent_size"]))
                lat_shapes.append((h, w))
                latents.append(torch.randn(cfg["vae"]["z_channels"], H // cfg["vae"]["downsample"], Wd // cfg["vae"]["downsample"], generator=g))
                text_ids.append(ids["end_of_image"]); text_idx.append(curr); curr += 1
                n = n_img + 2
                pos.extend([rope] * n)
                rope = rope if loss else rope + 1
                mode = "noise" if loss else "full"
            splits.append(n); modes.append(mode)
        sample_lens.append(sum(splits))
        masks.append(O.attention_mask_per_sample(splits, modes))
        all_splits += splits; all_modes += modes
    batch = dict(sequence_length=curr, packed_text_ids=torch.tensor(text_ids), packed_text_indexes=torch.tensor(text_idx),
                 sample_lens=sample_lens, packed_position_ids=torch.tensor(pos), nested_attention_masks=masks)
    if ce_idx:
        batch.update(ce_loss_indexes=torch.tensor(ce_idx), packed_label_ids=torch.tensor(labels))
    if vit_tokens:
        batch.update(packed_vit_tokens=torch.cat(vit_tokens, 0), packed_vit_token_indexes=torch.tensor(vit_idx),
                     packed_vit_position_ids=torch.cat(vit_pos, 0), vit_token_seqlens=torch.tensor(vit_lens, dtype=torch.int))
    noise = None
    if latents:
        Hm = max(l.shape[1] for l in latents); Wm = max(l.shape[2] for l in latents)
        padded = torch.zeros(len(latents), cfg["vae"]["z_channels"], Hm, Wm)
        for i, l in enumerate(latents):
            padded[i, :, :l.shape[1], :l.shape[2]] = l
        batch.update(padded_latent=padded, patchified_vae_latent_shapes=lat_shapes, packed_latent_position_ids=torch.cat(lat_pos, 0),
                     packed_vae_token_indexes=torch.tensor(vae_idx), packed_timesteps=torch.tensor(timesteps))
        if mse_idx:
            batch["mse_loss_indexes"] = torch.tensor(mse_idx)
        noise = torch.randn(len(vae_idx), cfg["bagel"]["latent_patch_size"] ** 2 * cfg["vae"]["z_channels"], generator=g)
    return batch, noise, all_splits, all_modes


def text_only_training_grads(W, cfg, batch, ce_loss_weights=None):
    """Gradient oracle of a TEXT-ONLY pack (no ViT image, no latent: the single-expert path of the engines): the oracle's primitives --
    embed_tokens, llm_forward_train with an empty gen row list, lm_head, cross-entropy -- under torch autograd.  (bagel_forward_train
    restates the reference's Bagel.forward, which indexes every modality unconditionally, bagel.py:150-226.)
    -> (loss, {state-dict key: gradient})."""
    import torch.nn.functional as F
    names = [k for k in W if k.startswith("language_model.") and "moe_gen" not in k and "inv_freq" not in k]
    Wg = {k: (v.detach().clone().requires_grad_(True) if k in names else v) for k, v in W.items()}
    O.GRAD_ENABLED = True
    try:
        te = O.embed_tokens(Wg, batch["packed_text_ids"])
        seq = te.new_zeros((batch["sequence_length"], cfg["llm"]["hidden_size"]))
        seq[batch["packed_text_indexes"]] = te
        out = O.llm_forward_train(Wg, cfg["llm"], seq, batch["sample_lens"], batch["nested_attention_masks"], batch["packed_position_ids"],
                                  batch["packed_text_indexes"], torch.zeros(0, dtype=torch.long))
        logits = O.linear(out[batch["ce_loss_indexes"]], Wg["language_model.lm_head.weight"])
        ce = F.cross_entropy(logits.float(), batch["packed_label_ids"], reduction="none")
        loss = O.training_step_loss(dict(ce=ce, mse=None), ce_loss_weights)
        loss.backward()
    finally:
        O.GRAD_ENABLED = False
    return float(loss.detach()), {k: Wg[k].grad for k in names if Wg[k].grad is not None}
