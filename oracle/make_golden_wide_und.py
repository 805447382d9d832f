"""Generate tests/golden/wide7b_und.pt and tests/golden/vae_full.pt from the UNMODIFIED reference at the shapes BASELINE.json
configs[1] (image understanding) and the real VAE actually run, and pin the oracle against them bit for bit.

TEST INFRASTRUCTURE ONLY.  Run in the build container (needs /root/reference, ~10 GB of RAM, a few minutes on 8 cores):

    python -m oracle.make_golden_wide_und [--only und|vae]

(1) ``wide7b_und``: SigLIP at so400m WIDTH (hidden 1152, 16 heads x 72, MLP 4304, patch 14, learned position table 70 x 70; 2 layers)
    on a 980 x 980 image = 4900 patches (siglip_navit.py:198-245: head_dim 72, the 588 -> 1152 patch embedding, one 4900-token
    non-causal sequence), the 1152 -> 3584 connector + the 70 x 70 sincos table, then a BAGEL-7B-WIDTH LLM of 2 MoT layers (oracle
    config WIDE7B_UND): ViT prefill -> text prefill -> 8 greedy tokens (bagel.py:362-414, 267-296, 930-1000).  The reference classes
    run with bf16 weights under torch.autocast('cpu', bf16) exactly as in oracle/make_golden.py.
    The image is re-drawn from its seed by the test (torch's CPU generator is host independent); the fixture holds a checksum, ROW
    SAMPLES of the SigLIP features and of the KV caches (every 13th row and the last 48 -- the prompt + decoded rows), the logits of
    every decode step and the token ids, so the fixture stays a few MB.
(2) ``vae_full``: the real ``AutoEncoderParams`` (autoencoder.py:340-351: ch 128, ch_mult [1, 2, 4, 4], 2 res blocks, z 16 --
    512-channel convolutions and the mid-block attention) at 256 x 256: encode (32 x 32 latent grid, 1024-token attention) and
    decode, fp32 like the reference (app.py:48,138).
The noise floors recorded next to the outputs come from the oracle re-run with fp32-accumulating linears (another summation order on
the same operands), as in oracle/make_golden_wide.py."""
import argparse
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
from oracle.configs import WIDE7B_UND, VAE_FULL, NEW_TOKEN_IDS_TINY, StubTokenizer  # noqa: E402

PROMPT = "what is in this picture? answer briefly."
IMAGE_SEED = 17
IMAGE_HW = (980, 980)
MAX_LENGTH = 8


def und_image():
    g = torch.Generator().manual_seed(IMAGE_SEED)
    return torch.rand(3, *IMAGE_HW, generator=g) * 2 - 1


def sample_rows(n):
    """Rows kept in the fixture: every 13th and the last 48."""
    return torch.unique(torch.cat([torch.arange(0, n, 13), torch.arange(max(0, n - 48), n)]))


def rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm())


def main_und():
    cfg = WIDE7B_UND
    t0 = time.time()
    model, vae, W, VW = MG.build(cfg)
    print(f"reference model built in {time.time() - t0:.0f} s", flush=True)
    from modeling.bagel.qwen2_navit import NaiveCache
    import copy
    L = cfg["llm"]["num_hidden_layers"]
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    ident = lambda t: t  # noqa: E731
    img = und_image()
    ps, side = cfg["vit"]["patch_size"], cfg["bagel"]["vit_max_num_patch_per_side"]
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        ti, l1, r1 = model.prepare_vit_images([0], [0], [img], ident, NEW_TOKEN_IDS_TINY)
        oti, ol1, or1 = P.prepare_vit_images([0], [0], [img], ident, NEW_TOKEN_IDS_TINY, ps, side)
        MG.same_dict(ti, oti, "prepare_vit_images")
        n_vit = int(ti["vit_token_seqlens"][0])
        assert n_vit == 4900
        # the encoder alone (what forward_cache_update_vit calls first, bagel.py:379-387)
        cu = torch.nn.functional.pad(torch.cumsum(ti["vit_token_seqlens"], 0), (1, 0)).to(torch.int32)
        t1 = time.time()
        feats = model.vit_model(packed_pixel_values=ti["packed_vit_tokens"], packed_flattened_position_ids=ti["packed_vit_position_ids"],
                                cu_seqlens=cu, max_seqlen=n_vit)
        print(f"reference SigLIP ({n_vit} tokens, width {cfg['vit']['hidden_size']}): {time.time() - t1:.0f} s", flush=True)
        ofeats = O.siglip_forward(W, cfg["vit"], ti["packed_vit_tokens"], ti["packed_vit_position_ids"], cu, n_vit)
        MG.same(feats, ofeats, "siglip features (oracle vs reference, so400m width)")
        t1 = time.time()
        cache = model.forward_cache_update_vit(NaiveCache(L), **ti)
        print(f"reference ViT prefill through the 7B-width LLM: {time.time() - t1:.0f} s", flush=True)
        ocache = O.forward_cache_update_vit(W, cfg, O.OracleCache(L), **oti)
        MG.same(MG.cache_to_lists(cache, L), MG.cache_to_lists(ocache, L), "vit prefill cache")
        pi, l2, r2 = model.prepare_prompts(l1, r1, [PROMPT], tok, NEW_TOKEN_IDS_TINY)
        cache = model.forward_cache_update_text(cache, **pi)
        ocache = O.forward_cache_update_text(W, cfg, ocache, **P.prepare_prompts(l1, r1, [PROMPT], tok, NEW_TOKEN_IDS_TINY)[0])
        MG.same(MG.cache_to_lists(cache, L), MG.cache_to_lists(ocache, L), "vit+text prefill cache")
        si = model.prepare_start_tokens(l2, r2, NEW_TOKEN_IDS_TINY)
        MG.same_dict(si, P.prepare_start_tokens(l2, r2, NEW_TOKEN_IDS_TINY), "prepare_start_tokens")
        toks = model.generate_text(past_key_values=copy.deepcopy(cache), max_length=MAX_LENGTH, do_sample=False, end_token_id=None, **si)
        dcache = ocache.clone()
        otoks, ologits = O.generate_text(W, cfg, dcache, si["packed_key_value_indexes"], si["key_values_lens"], si["packed_start_tokens"],
                                         si["packed_query_position_ids"], MAX_LENGTH, return_logits=True)
        MG.same(toks, otoks, "greedy tokens")
        # accumulation-order noise floor of the reference itself at this width (fp32-accumulating linears, same operands)
        O.LINEAR_FP32_ACCUM = True
        try:
            ofeats32 = O.siglip_forward(W, cfg["vit"], ti["packed_vit_tokens"], ti["packed_vit_position_ids"], cu, n_vit)
            oc32 = O.forward_cache_update_vit(W, cfg, O.OracleCache(L), **oti)
            oc32 = O.forward_cache_update_text(W, cfg, oc32, **pi)
            _, ologits32 = O.generate_text(W, cfg, oc32.clone(), si["packed_key_value_indexes"], si["key_values_lens"],
                                           si["packed_start_tokens"], si["packed_query_position_ids"], MAX_LENGTH, return_logits=True)
        finally:
            O.LINEAR_FP32_ACCUM = False
    kc, vc = MG.cache_to_lists(cache, L)
    kc32, vc32 = MG.cache_to_lists(oc32, L)
    n_ctx = kc[0].shape[0]
    rows = sample_rows(n_ctx)
    frows = sample_rows(n_vit)
    noise = dict(siglip=rel(ofeats32, feats), kv=max(max(rel(a, b) for a, b in zip(kc32, kc)), max(rel(a, b) for a, b in zip(vc32, vc))),
                 logits_step0=rel(ologits32[0], ologits[0]))
    print("reference accumulation-order noise floor (fp32-accumulating oracle vs reference, rel-L2):", noise)
    out = dict(prompt=PROMPT, image_seed=IMAGE_SEED, image_hw=IMAGE_HW, image_checksum=float(img.double().sum()),
               image_probe=img[:, ::97, ::89].clone(), n_vit=n_vit, n_ctx=n_ctx, lens=[l1, l2], ropes=[r1, r2],
               siglip_rows=frows, siglip_out=feats[frows].clone(), kv_rows=rows,
               key_cache=[k[rows].clone() for k in kc], value_cache=[v[rows].clone() for v in vc],
               start_inputs=si, tokens=toks, logits=ologits, max_length=MAX_LENGTH, noise_floor=noise,
               host=dict(torch=torch.__version__))
    path = os.path.join(MG.GOLD, "wide7b_und.pt")
    torch.save(out, path)
    print(f"wrote {path} ({os.path.getsize(path) / 1e6:.1f} MB) in {time.time() - t0:.0f} s")


def main_vae():
    cfg = VAE_FULL
    t0 = time.time()
    MG.ref_env.activate()
    from modeling.autoencoder import AutoEncoder, AutoEncoderParams
    from oracle.weights import load_synth
    vae = AutoEncoder(AutoEncoderParams(**cfg["vae"]))
    load_synth(vae, MG.WEIGHT_SEED)
    vae = vae.eval()
    VW = {k: v for k, v in vae.state_dict().items()}
    g = torch.Generator().manual_seed(23)
    x = torch.rand(1, 3, 256, 256, generator=g) * 2 - 1
    z = torch.randn(1, cfg["vae"]["z_channels"], 32, 32, generator=g)
    with torch.no_grad():
        t1 = time.time()
        dec = vae.decode(z)
        print(f"reference vae.decode 256^2: {time.time() - t1:.0f} s", flush=True)
        MG.same(dec, O.vae_decode(VW, cfg["vae"], z), "vae.decode (full-size VAE)")
        torch.manual_seed(47)
        t1 = time.time()
        enc = vae.encode(x)
        print(f"reference vae.encode 256^2: {time.time() - t1:.0f} s", flush=True)
        torch.manual_seed(47)
        noise = torch.randn(1, cfg["vae"]["z_channels"], 32, 32)
        MG.same(enc, O.vae_encode(VW, cfg["vae"], x, noise), "vae.encode (full-size VAE)")
    out = dict(x=x, z=z, enc_noise=noise, encoded=enc, decoded=dec, host=dict(torch=torch.__version__))
    path = os.path.join(MG.GOLD, "vae_full.pt")
    torch.save(out, path)
    print(f"wrote {path} ({os.path.getsize(path) / 1e6:.1f} MB) in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None, choices=[None, "und", "vae"])
    a = ap.parse_args()
    if a.only in (None, "vae"):
        main_vae()
    if a.only in (None, "und"):
        main_und()
