"""tests/golden/tiny*_t2i_fp32master.pt: the reference's text->image latents when the model keeps FP32 MASTER WEIGHTS under
torch.autocast(bf16) -- what eval/gen/gen_images_mp.py:165-176 does (it never casts the model; the inference loop runs under
autocast, gen_images_mp.py:54) -- next to the bf16-weights run of app.py:105-113 that every other fixture uses.

TEST INFRASTRUCTURE ONLY (needs /root/reference):   python -m oracle.make_golden_fp32master

Under autocast every Linear still multiplies bf16 operands, but everything autocast does not touch -- the residual stream, the
RMSNorm weights and outputs, the embeddings, the timestep embedder's output, llm2vae's input -- stays fp32 instead of being rounded to
bf16 at every step.  The product (like app.py) holds bf16 weights and a bf16 residual stream; loading an fp32 checkpoint casts once
(bagel.py warns).  The fixture records the same scenario as <cfg>_t2i.pt (same weights before the cast, prompts, noise, sampler
arguments) so tests can state how far the two reference precisions are apart and bound the product's distance from the fp32-master run."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import make_golden as MG          # noqa: E402
from oracle import ref_env                    # noqa: E402
from oracle.configs import TINY, TINY_D128, NEW_TOKEN_IDS_TINY, StubTokenizer  # noqa: E402
from oracle.weights import load_synth         # noqa: E402


def build_fp32(cfg):
    ref_env.activate()
    from modeling.bagel import BagelConfig, Bagel, Qwen2Config, Qwen2ForCausalLM, SiglipVisionConfig, SiglipVisionModel
    from modeling.autoencoder import AutoEncoderParams
    llm_config = Qwen2Config(pad_token_id=None, **cfg["llm"])
    vit_config = SiglipVisionConfig(**cfg["vit"])
    bc = BagelConfig(visual_gen=True, visual_und=True, llm_config=llm_config, vit_config=vit_config,
                     vae_config=AutoEncoderParams(**cfg["vae"]), **cfg["bagel"])
    model = Bagel(Qwen2ForCausalLM(llm_config), SiglipVisionModel(vit_config), bc)
    model.vit_model.vision_model.embeddings.convert_conv2d_to_linear(vit_config)
    load_synth(model, MG.WEIGHT_SEED)
    # the synthetic checkpoint is bf16-representable (every other fixture casts it to bf16): round the values, keep fp32 STORAGE, so
    # both precisions start from numerically identical weights and only the activation path differs
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(p.to(torch.bfloat16).float())
    return model.eval()


def main():
    for cfg in (TINY, TINY_D128):
        g = torch.load(os.path.join(MG.GOLD, f"{cfg['name']}_t2i.pt"), weights_only=False)
        model = build_fp32(cfg)
        from modeling.bagel.qwen2_navit import NaiveCache
        L = cfg["llm"]["num_hidden_layers"]
        tok = StubTokenizer(cfg["llm"]["vocab_size"])
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
            gi, newlens, newrope = model.prepare_prompts([0, 0], [0, 0], g["prompts"], tok, NEW_TOKEN_IDS_TINY)
            cache = model.forward_cache_update_text(NaiveCache(L), **gi)
            ci = g["cfg_inputs"]
            out = {}
            for kw_key, key in (("gen_kwargs", "latents"), ("gen_kwargs_channel", "latents_channel")):
                lat = model.generate_image(
                    past_key_values=cache, cfg_text_past_key_values=NaiveCache(L),
                    cfg_text_packed_position_ids=ci["cfg_packed_position_ids"], cfg_text_packed_query_indexes=ci["cfg_packed_query_indexes"],
                    cfg_text_key_values_lens=ci["cfg_key_values_lens"], cfg_text_packed_key_value_indexes=ci["cfg_packed_key_value_indexes"],
                    **g[kw_key], **g["latent_inputs"])
                out[key] = [x.float() for x in lat]
        rel = lambda a, b: float((a - b).norm() / b.norm())  # noqa: E731
        dev = {k: max(rel(a, b) for a, b in zip(out[k], g[k])) for k in out}
        x0 = g["latent_inputs"]["packed_init_noises"]
        print(cfg["name"], "bf16-weights reference vs fp32-master reference, rel-L2 of the final latents:", dev)
        out["deviation_of_bf16_weights_reference"] = dev
        out["kv_dtype"] = str(cache.key_cache[0].dtype)
        path = os.path.join(MG.GOLD, f"{cfg['name']}_t2i_fp32master.pt")
        torch.save(out, path)
        print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
