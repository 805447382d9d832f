"""Generate tests/golden/wide7b_edit.pt: ONE IMAGE-EDIT REQUEST (BASELINE.json configs[4]) at BAGEL-7B-MoT WIDTH (oracle.configs.WIDE7B:
hidden 3584, intermediate 18944, 28/4 heads x 128, 2 MoT layers) from the UNMODIFIED reference, and pin the oracle to it bit for bit.

TEST INFRASTRUCTURE ONLY.  Run in the build container (needs /root/reference, ~8 GB of RAM, a few minutes on 8 cores):

    python -m oracle.make_golden_wide_edit

What configs[4] does that the text->image fixtures do not (app.py:224-228, inferencer.py:99-172, bagel.py:491-550,757-907):
  * the source image enters the context TWICE: VAE-encoded latents prefilled in GEN mode at t = 0 (forward_cache_update_vae: the gen expert's
    weights and fp32 QK-norm on the context rows) and SigLIP tokens prefilled in und mode, then the prompt;
  * the sampler runs THREE forwards per step on three different contexts -- cond (VAE + ViT + prompt), cfg-text (VAE + ViT), cfg-img (prompt
    only) -- with cfg_text_scale 4, cfg_img_scale 2 and the ``text_channel`` renorm.
Resolution is reduced to what the reference finishes on a CPU (512 x 384 source and target: 768 latent tokens + 2 markers per stream on
contexts of 866 / 852 / 14 keys; WIDE7B's ViT takes at most 10 x 10 patches: a 140 x 112 view); the width, the head layout and the launch
structure are the benchmark's.  3 Euler steps (num_timesteps 4).  Also recorded: the reference's own accumulation-order noise floor (the
oracle re-run with fp32-accumulating linears), from which tests/test_wide_gpu.py derives its tolerances."""
import copy
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import bagel_oracle as O          # noqa: E402
from oracle import make_golden as MG          # noqa: E402
from oracle import packers as P               # noqa: E402
from oracle.configs import WIDE7B, NEW_TOKEN_IDS_TINY, StubTokenizer  # noqa: E402

PROMPT = "make it blue"
VAE_HW, VIT_HW = (384, 512), (112, 140)
KW = dict(num_timesteps=4, timestep_shift=3.0, cfg_renorm_min=0.0, cfg_renorm_type="text_channel", cfg_interval=[0.0, 1.0],
          cfg_text_scale=4.0, cfg_img_scale=2.0)


def oc(c, d):
    return dict(cache=c, position_ids=d["cfg_packed_position_ids"], query_indexes=d["cfg_packed_query_indexes"],
                key_values_lens=d["cfg_key_values_lens"], key_value_indexes=d["cfg_packed_key_value_indexes"])


def oracle_request(W, VW, cfg, ovi, oti, enc_noise, l2, r2, li, ct, cim, tok):
    """The whole request through the oracle: three contexts, first-step velocity, latents."""
    L = cfg["llm"]["num_hidden_layers"]
    c = O.forward_cache_update_vae(W, cfg, VW, O.OracleCache(L), sample_noise=enc_noise, **ovi)
    c = O.forward_cache_update_vit(W, cfg, c, **oti)
    c_text = c.clone()
    c = O.forward_cache_update_text(W, cfg, c, **P.prepare_prompts(l2, r2, [PROMPT], tok, NEW_TOKEN_IDS_TINY)[0])
    c_img = O.forward_cache_update_text(W, cfg, O.OracleCache(L), **P.prepare_prompts([0], [0], [PROMPT], tok, NEW_TOKEN_IDS_TINY)[0])
    ts, _ = O.flow_schedule(KW["num_timesteps"], KW["timestep_shift"])
    x0 = li["packed_init_noises"]
    v0 = O.forward_flow(W, cfg, x0, torch.tensor([ts[0]] * x0.shape[0]), li, c, oc(c_text, ct), oc(c_img, cim), KW["cfg_text_scale"],
                        KW["cfg_img_scale"], KW["cfg_renorm_min"], KW["cfg_renorm_type"])
    lat = O.generate_image(W, cfg, li, c, cfg_text=oc(c_text, ct), cfg_img=oc(c_img, cim), **KW)
    return c, c_text, c_img, v0, lat


IMG_SEED = 11


def images():
    """The two views of the source image (vae_transform / vit_transform outputs): a seeded draw, re-made by the test (the fixture keeps
    only a checksum of each)."""
    g = torch.Generator().manual_seed(IMG_SEED)
    img_vae = torch.rand(3, *VAE_HW, generator=g) * 2 - 1
    img_vit = torch.rand(3, *VIT_HW, generator=g) * 2 - 1
    return img_vae, img_vit


def compact(d):
    """What the fixture does not need to carry: the images (seeded), the packer outputs that hold them again, and the cfg-text context --
    the first lens[1] rows of the cond context bit for bit (prefill appends; checked here)."""
    L = len(d["key_cache"])
    n = d["key_cache_img"][0].shape[0]
    for i in range(L):
        assert torch.equal(d["key_cache"][i][:n], d["key_cache_img"][i]) and torch.equal(d["value_cache"][i][:n], d["value_cache_img"][i])
    a, b = images()
    assert torch.equal(a, d["img_vae"]) and torch.equal(b, d["img_vit"])
    d["img_seed"], d["img_checksum"], d["n_img_ctx"] = IMG_SEED, [float(a.double().sum()), float(b.double().sum())], n
    for k in ("img_vae", "img_vit", "key_cache_img", "value_cache_img", "vae_inputs", "vit_inputs"):
        d.pop(k, None)
    return d


def main():
    cfg = WIDE7B
    t0 = time.time()
    model, vae, W, VW = MG.build(cfg)
    from modeling.bagel.qwen2_navit import NaiveCache
    L = cfg["llm"]["num_hidden_layers"]
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    ds = cfg["vae"]["downsample"] * cfg["bagel"]["latent_patch_size"]
    img_vae, img_vit = images()                                # vae_transform(image) / vit_transform(image) outputs (14-px patches)
    ident = lambda t: t  # noqa: E731
    fvae = MG._Fp32Vae(vae)
    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())  # noqa: E731
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        vi, l1, r1 = model.prepare_vae_images([0], [0], [img_vae], ident, NEW_TOKEN_IDS_TINY)
        ovi, ol1, or1 = P.prepare_vae_images([0], [0], [img_vae], ident, NEW_TOKEN_IDS_TINY, ds, cfg["bagel"]["max_latent_size"])
        MG.same_dict(vi, ovi, "prepare_vae_images")
        torch.manual_seed(43)
        cache = model.forward_cache_update_vae(fvae, NaiveCache(L), **vi)
        torch.manual_seed(43)
        enc_noise = torch.randn((1, cfg["vae"]["z_channels"], VAE_HW[0] // 8, VAE_HW[1] // 8))
        ti, l2, r2 = model.prepare_vit_images(l1, r1, [img_vit], ident, NEW_TOKEN_IDS_TINY)
        oti, ol2, or2 = P.prepare_vit_images(l1, r1, [img_vit], ident, NEW_TOKEN_IDS_TINY, cfg["vit"]["patch_size"],
                                             cfg["bagel"]["vit_max_num_patch_per_side"])
        MG.same_dict(ti, oti, "prepare_vit_images")
        cache = model.forward_cache_update_vit(cache, **ti)
        cfg_text_cache = copy.deepcopy(cache)
        pi, l3, r3 = model.prepare_prompts(l2, r2, [PROMPT], tok, NEW_TOKEN_IDS_TINY)
        cache = model.forward_cache_update_text(cache, **pi)
        pi2, l4, r4 = model.prepare_prompts([0], [0], [PROMPT], tok, NEW_TOKEN_IDS_TINY)
        cimg_cache = model.forward_cache_update_text(NaiveCache(L), **pi2)
        size = [VAE_HW]
        torch.manual_seed(44)
        li = model.prepare_vae_latent(l3, r3, size, NEW_TOKEN_IDS_TINY)
        ct = model.prepare_vae_latent_cfg(l2, r2, size)
        cim = model.prepare_vae_latent_cfg(l4, r4, size)
        print(f"contexts: cond {l3}, cfg-text {l2}, cfg-img {l4}; query rows {li['packed_seqlens'].tolist()}", flush=True)
        t1 = time.time()
        lat = model.generate_image(
            past_key_values=cache, cfg_text_past_key_values=cfg_text_cache, cfg_img_past_key_values=cimg_cache,
            cfg_text_packed_position_ids=ct["cfg_packed_position_ids"], cfg_text_packed_query_indexes=ct["cfg_packed_query_indexes"],
            cfg_text_key_values_lens=ct["cfg_key_values_lens"], cfg_text_packed_key_value_indexes=ct["cfg_packed_key_value_indexes"],
            cfg_img_packed_position_ids=cim["cfg_packed_position_ids"], cfg_img_packed_query_indexes=cim["cfg_packed_query_indexes"],
            cfg_img_key_values_lens=cim["cfg_key_values_lens"], cfg_img_packed_key_value_indexes=cim["cfg_packed_key_value_indexes"],
            **KW, **li)
        print(f"reference generate_image (3 Euler steps x 3 forwards): {time.time() - t1:.0f} s", flush=True)
        oc_, oc_text, oc_img, v0, olat = oracle_request(W, VW, cfg, ovi, oti, enc_noise, l2, r2, li, ct, cim, tok)
        MG.same(MG.cache_to_lists(cache, L), MG.cache_to_lists(oc_, L), "cond context (VAE + ViT + prompt)")
        MG.same(MG.cache_to_lists(cfg_text_cache, L), MG.cache_to_lists(oc_text, L), "cfg-text context (VAE + ViT)")
        MG.same(MG.cache_to_lists(cimg_cache, L), MG.cache_to_lists(oc_img, L), "cfg-img context (prompt)")
        MG.same(list(lat), list(olat), "edit latents (oracle vs the unmodified reference, 7B width)")
        O.LINEAR_FP32_ACCUM = True
        try:
            c32, c32_text, c32_img, v0_32, olat32 = oracle_request(W, VW, cfg, ovi, oti, enc_noise, l2, r2, li, ct, cim, tok)
        finally:
            O.LINEAR_FP32_ACCUM = False
    x0 = li["packed_init_noises"]
    kc, vc = MG.cache_to_lists(cache, L)
    kc32, vc32 = MG.cache_to_lists(c32, L)
    noise = dict(kv=max(max(rel(a, b) for a, b in zip(kc32, kc)), max(rel(a, b) for a, b in zip(vc32, vc))), v_first_step=rel(v0_32, v0),
                 latents=rel(olat32[0], lat[0]), displacement=rel(olat32[0] - x0, lat[0] - x0))
    print("reference accumulation-order noise floor (fp32-accumulating oracle vs reference, rel-L2):", noise)
    kct, vct = MG.cache_to_lists(cfg_text_cache, L)
    kci, vci = MG.cache_to_lists(cimg_cache, L)
    out = compact(dict(prompt=PROMPT, img_vae=img_vae, img_vit=img_vit, enc_noise=enc_noise, vae_inputs=vi, vit_inputs=ti, lens=[l1, l2, l3, l4],
                       ropes=[r1, r2, r3, r4], image_size=size, key_cache=kc, value_cache=vc, key_cache_img=kct, value_cache_img=vct,
                       key_cache_txt=kci, value_cache_txt=vci, latent_inputs=li, cfg_text_inputs=ct, cfg_img_inputs=cim, gen_kwargs=KW,
                       latents=list(lat), v_first_step=v0, noise_floor=noise, v_first_step_f32acc=v0_32, latents_f32acc=list(olat32),
                       host=dict(torch=torch.__version__)))
    path = os.path.join(MG.GOLD, "wide7b_edit.pt")
    torch.save(out, path)
    print(f"wrote {path} ({os.path.getsize(path) / 1e6:.1f} MB) in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
