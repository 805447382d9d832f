"""Generate tests/golden/wide7b_t2i.pt from the UNMODIFIED reference at BAGEL-7B-MoT WIDTH (oracle.configs.WIDE7B: hidden 3584,
intermediate 18944, 28/4 heads x 128, 2 MoT layers) and pin the oracle against it bit for bit.

TEST INFRASTRUCTURE ONLY.  Run in the build container (needs /root/reference, ~6 GB of RAM, about two minutes on 8 cores):

    python -m oracle.make_golden_wide

Scenario = BASELINE.json configs[2] cut down to what a CPU can run: ONE 1024x1024 sample (4096 latent tokens + 2 markers = the
4098-row sequences of the benchmark) on a text context, num_timesteps = 4 (3 Euler steps), CFG-text 4.0 on the whole interval,
global renorm, timestep_shift 3 (gen_images_mp.py:178-182).  The reference classes run with bf16 weights under
torch.autocast('cpu', bf16) exactly as in oracle/make_golden.py.  The fixture holds inputs + reference outputs (prefill KV, first
step velocity, final latents); weights are re-synthesised from (key, shape, seed) by oracle/weights.py.
SURVEY.md section 8c asked for the multi-step trajectory tolerance at 7B shapes "to be confirmed empirically and then frozen":
tests/test_wide_gpu.py is where it is frozen."""
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

PROMPT = "a photo of a small red cube on a wooden table"      # 45 characters -> 45 + 2 context tokens with the stub tokenizer
SIZES = [(1024, 1024)]
KW = dict(num_timesteps=4, timestep_shift=3.0, cfg_renorm_min=0.0, cfg_renorm_type="global", cfg_interval=[0.0, 1.0],
          cfg_text_scale=4.0)


def main():
    cfg = WIDE7B
    t0 = time.time()
    model, vae, W, VW = MG.build(cfg)
    print(f"reference model built in {time.time() - t0:.0f} s", flush=True)
    from modeling.bagel.qwen2_navit import NaiveCache
    L = cfg["llm"]["num_hidden_layers"]
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    ds = cfg["vae"]["downsample"] * cfg["bagel"]["latent_patch_size"]
    pdim = cfg["bagel"]["latent_patch_size"] ** 2 * cfg["vae"]["z_channels"]
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        gi, newlens, newrope = model.prepare_prompts([0], [0], [PROMPT], tok, NEW_TOKEN_IDS_TINY)
        ogi, onl, onr = P.prepare_prompts([0], [0], [PROMPT], tok, NEW_TOKEN_IDS_TINY)
        MG.same_dict(gi, ogi, "prepare_prompts")
        cache = model.forward_cache_update_text(NaiveCache(L), **gi)
        ocache = O.forward_cache_update_text(W, cfg, O.OracleCache(L), **ogi)
        MG.same(MG.cache_to_lists(cache, L), MG.cache_to_lists(ocache, L), "text prefill cache")
        torch.manual_seed(42)
        li = model.prepare_vae_latent(newlens, newrope, SIZES, NEW_TOKEN_IDS_TINY)
        torch.manual_seed(42)
        oli = P.prepare_vae_latent(newlens, newrope, SIZES, NEW_TOKEN_IDS_TINY, ds, cfg["bagel"]["max_latent_size"], pdim)
        MG.same_dict(li, oli, "prepare_vae_latent")
        ci = model.prepare_vae_latent_cfg([0], [0], SIZES)
        t1 = time.time()
        lat = model.generate_image(
            past_key_values=cache, cfg_text_past_key_values=NaiveCache(L),
            cfg_text_packed_position_ids=ci["cfg_packed_position_ids"], cfg_text_packed_query_indexes=ci["cfg_packed_query_indexes"],
            cfg_text_key_values_lens=ci["cfg_key_values_lens"], cfg_text_packed_key_value_indexes=ci["cfg_packed_key_value_indexes"],
            **KW, **li)
        print(f"reference generate_image: {time.time() - t1:.0f} s", flush=True)
        ocfg = dict(cache=O.OracleCache(L), position_ids=ci["cfg_packed_position_ids"], query_indexes=ci["cfg_packed_query_indexes"],
                    key_values_lens=ci["cfg_key_values_lens"], key_value_indexes=ci["cfg_packed_key_value_indexes"])
        t1 = time.time()
        olat = O.generate_image(W, cfg, oli, ocache, cfg_text=ocfg, **KW)
        print(f"oracle generate_image: {time.time() - t1:.0f} s", flush=True)
        MG.same(list(lat), list(olat), "generate_image latents (oracle vs reference, 7B width)")
        ts, _ = O.flow_schedule(KW["num_timesteps"], KW["timestep_shift"])
        timestep = torch.tensor([ts[0]] * li["packed_init_noises"].shape[0])
        v0 = O.forward_flow(W, cfg, oli["packed_init_noises"], timestep, oli, ocache, ocfg, None, 4.0, 1.0, 0.0, "global")
        # the reference's own accumulation-order noise at this width: the oracle again with fp32-accumulating linears (same operands,
        # same rounding points, another summation order) -- prefill, first-step velocity and the whole Euler loop
        O.LINEAR_FP32_ACCUM = True
        try:
            ocache32 = O.forward_cache_update_text(W, cfg, O.OracleCache(L), **ogi)
            ocfg32 = dict(ocfg, cache=O.OracleCache(L))
            v0_32 = O.forward_flow(W, cfg, oli["packed_init_noises"], timestep, oli, ocache32, ocfg32, None, 4.0, 1.0, 0.0, "global")
            olat32 = O.generate_image(W, cfg, oli, ocache32, cfg_text=ocfg32, **KW)
        finally:
            O.LINEAR_FP32_ACCUM = False
    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())  # noqa: E731
    kc32, vc32 = MG.cache_to_lists(ocache32, L)
    kc, vc = MG.cache_to_lists(cache, L)
    x0 = li["packed_init_noises"]
    noise = dict(kv=max(max(rel(a, b) for a, b in zip(kc32, kc)), max(rel(a, b) for a, b in zip(vc32, vc))), v_first_step=rel(v0_32, v0),
                 latents=rel(olat32[0], lat[0]), displacement=rel(olat32[0] - x0, lat[0] - x0))
    print("reference accumulation-order noise floor at 7B width (fp32-accumulating oracle vs reference, rel-L2):", noise)
    print(f"latents: |x0| rms {x0.float().pow(2).mean().sqrt():.3f}, |x_T| rms {lat[0].float().pow(2).mean().sqrt():.3f}, "
          f"|x_T - x0| rms {(lat[0] - x0).float().pow(2).mean().sqrt():.3f}, |v0| rms {v0.float().pow(2).mean().sqrt():.3f}")
    out = dict(prompt=PROMPT, image_sizes=SIZES, prompt_inputs=gi, newlens=newlens, newrope=newrope, key_cache=kc, value_cache=vc,
               latent_inputs=li, cfg_inputs=ci, gen_kwargs=KW, latents=list(lat), v_first_step=v0,
               noise_floor=noise, v_first_step_f32acc=v0_32, latents_f32acc=list(olat32),
               host=dict(torch=torch.__version__, cpu_bf16_backend="mkldnn" if torch.backends.mkldnn.is_available() else "native"))
    path = os.path.join(MG.GOLD, "wide7b_t2i.pt")
    torch.save(out, path)
    print(f"wrote {path} ({os.path.getsize(path) / 1e6:.1f} MB) in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
