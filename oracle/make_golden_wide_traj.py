"""Generate tests/golden/wide7b_traj49.pt: the FULL 49-step Euler trajectory of BASELINE.json configs[2] at BAGEL-7B-MoT WIDTH
(oracle.configs.WIDE7B: hidden 3584, intermediate 18944, 28/4 heads x 128, 2 MoT layers) from the UNMODIFIED reference.

TEST INFRASTRUCTURE ONLY.  Run in the build container (needs /root/reference, ~8 GB of RAM, about an hour on 8 cores):

    python -m oracle.make_golden_wide_traj

SURVEY.md section 8c: "end-to-end latent rel-L2 after 49 steps at 7B shapes, confirmed then frozen".  tests/golden/wide7b_t2i.pt stops
after 3 Euler steps; the drift mode the survey names -- fp32 ``x_t`` integrating bf16 ``v_t`` over the whole schedule
(bagel.py:644-754) -- needs the whole schedule.  Scenario = gen_images_mp.py:178-182 exactly: num_timesteps 50 (49 Euler steps),
timestep_shift 3, CFG-text 4.0 on the whole interval [0, 1], global renorm, renorm_min 0; ONE 1024x1024 sample (4098-row sequences) on
the text context of wide7b_t2i.pt.

Three runs over the same inputs:
  1. the reference's own ``Bagel.generate_image`` (final latents: the golden);
  2. the oracle, step by step (bit-identical to 1. at the end -- asserted -- so its intermediate ``x_t`` ARE the reference's);
  3. the oracle with fp32-accumulating linears (same operands and rounding points, another summation order: what a GPU does) --
     the reference's own accumulation-order noise, per step: the DRIFT CURVE the GPU path is judged against.
The fixture holds the final latents of 1. and 3. in fp32, snapshots at SNAP steps (drift curve of the product), the per-step noise-floor
curve (rel-L2 of x_t, of the displacement x_t - x_0, of v_t) and the inputs.  Snapshots are stored as fp16 DISPLACEMENTS x_n - x_0 (keys
``snap_disp`` / ``snap_disp_f32acc``): x_n itself in fp16 carries a rounding error of ~2e-4 absolute, which is 2-3 % of the first step's
displacement (rms 0.0076) -- larger than the noise floor being measured there.

    python -m oracle.make_golden_wide_traj --refine-early 5

patches a fixture written by the first version of this script (fp16 x_n): it re-runs the two oracle passes for the first 5 steps only (~10 min),
checks that their fp16-rounded x_n reproduce the stored snapshots bit for bit, stores the exact displacements for those steps and converts the
later snapshots (whose displacement is >= 0.09 rms: fp16 error of x_n <= 0.3 % of it)."""
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
from oracle.make_golden_wide import PROMPT, SIZES  # noqa: E402

KW = dict(num_timesteps=50, timestep_shift=3.0, cfg_renorm_min=0.0, cfg_renorm_type="global", cfg_interval=[0.0, 1.0],
          cfg_text_scale=4.0)
SNAP = (1, 2, 3, 5, 10, 20, 30, 40)          # x_t AFTER this many Euler steps (fp16 snapshots); the final one is kept in fp32


def euler(W, cfg, gi, cache, cfg_text, tag, nsteps=None):
    """The loop of oracle.generate_image / bagel.py:691-752 with every intermediate kept."""
    x_t = gi["packed_init_noises"]
    ts, dts = O.flow_schedule(KW["num_timesteps"], KW["timestep_shift"])
    xs, vs = [], []
    t0 = time.time()
    for i, t in enumerate(ts):
        if nsteps is not None and i >= nsteps:
            break
        timestep = torch.tensor([t] * x_t.shape[0])
        use = t > KW["cfg_interval"][0] and t <= KW["cfg_interval"][1]
        v_t = O.forward_flow(W, cfg, x_t, timestep, gi, cache, cfg_text, None, KW["cfg_text_scale"] if use else 1.0, 1.0,
                             KW["cfg_renorm_min"], KW["cfg_renorm_type"])
        x_t = x_t - v_t.to(x_t.device) * dts[i]
        xs.append(x_t.clone())
        vs.append(v_t.clone())
        if i % 5 == 0:
            print(f"  {tag}: step {i + 1}/{len(ts)}  {time.time() - t0:.0f} s", flush=True)
    return xs, vs


def refine_early(nsteps):
    """See the module docstring: exact early displacements for a fixture that stored fp16 x_n."""
    cfg = WIDE7B
    path = os.path.join(MG.GOLD, "wide7b_traj49.pt")
    g = torch.load(path, weights_only=False)
    model, vae, W, VW = MG.build(cfg)
    L = cfg["llm"]["num_hidden_layers"]
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        ogi, _, _ = P.prepare_prompts([0], [0], [PROMPT], tok, NEW_TOKEN_IDS_TINY)
        oli, ci = g["latent_inputs"], g["cfg_inputs"]
        mk = lambda: dict(cache=O.OracleCache(L), position_ids=ci["cfg_packed_position_ids"], query_indexes=ci["cfg_packed_query_indexes"],  # noqa: E731
                          key_values_lens=ci["cfg_key_values_lens"], key_value_indexes=ci["cfg_packed_key_value_indexes"])
        ocache = O.forward_cache_update_text(W, cfg, O.OracleCache(L), **ogi)
        xs, _ = euler(W, cfg, oli, ocache, mk(), "oracle", nsteps)
        O.LINEAR_FP32_ACCUM = True
        try:
            ocache32 = O.forward_cache_update_text(W, cfg, O.OracleCache(L), **ogi)
            xs32, _ = euler(W, cfg, oli, ocache32, mk(), "oracle, fp32-accumulating linears", nsteps)
        finally:
            O.LINEAR_FP32_ACCUM = False
    x0 = oli["packed_init_noises"].float()
    disp, disp32 = [], []
    for n, sx, sx32 in zip(g["snap_steps"], g["snap_x"], g["snap_x_f32acc"]):
        if n <= nsteps:
            assert torch.equal(xs[n - 1].to(torch.float16), sx) and torch.equal(xs32[n - 1].to(torch.float16), sx32), f"step {n}: the partial re-run differs from the stored trajectory"
            rel = float(((xs32[n - 1] - x0) - (xs[n - 1] - x0)).norm() / (xs[n - 1] - x0).norm())
            assert abs(rel - g["noise_floor_curve"]["displacement"][n - 1]) < 1e-6, (n, rel)
            disp.append((xs[n - 1].float() - x0).to(torch.float16))
            disp32.append((xs32[n - 1].float() - x0).to(torch.float16))
            print(f"  step {n}: re-run reproduces the stored snapshots bit for bit; floor {rel:.3e}; exact displacement stored")
        else:
            disp.append((sx.float() - x0).to(torch.float16))
            disp32.append((sx32.float() - x0).to(torch.float16))
    g["snap_disp"], g["snap_disp_f32acc"] = disp, disp32
    del g["snap_x"], g["snap_x_f32acc"]
    g["snap_note"] = f"displacements x_n - x_0 in fp16; exact for n <= {nsteps}, from fp16 x_n snapshots beyond"
    torch.save(g, path)
    print(f"patched {path} ({os.path.getsize(path) / 1e6:.1f} MB)")


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--refine-early":
        return refine_early(int(sys.argv[2]))
    cfg = WIDE7B
    t0 = time.time()
    model, vae, W, VW = MG.build(cfg)
    from modeling.bagel.qwen2_navit import NaiveCache
    L = cfg["llm"]["num_hidden_layers"]
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    ds = cfg["vae"]["downsample"] * cfg["bagel"]["latent_patch_size"]
    pdim = cfg["bagel"]["latent_patch_size"] ** 2 * cfg["vae"]["z_channels"]
    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())  # noqa: E731
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        gi, newlens, newrope = model.prepare_prompts([0], [0], [PROMPT], tok, NEW_TOKEN_IDS_TINY)
        ogi, _, _ = P.prepare_prompts([0], [0], [PROMPT], tok, NEW_TOKEN_IDS_TINY)
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
        print(f"reference generate_image (49 Euler steps): {time.time() - t1:.0f} s", flush=True)
        ocfg = dict(cache=O.OracleCache(L), position_ids=ci["cfg_packed_position_ids"], query_indexes=ci["cfg_packed_query_indexes"],
                    key_values_lens=ci["cfg_key_values_lens"], key_value_indexes=ci["cfg_packed_key_value_indexes"])
        xs, vs = euler(W, cfg, oli, ocache, ocfg, "oracle")
        MG.same([lat[0]], [xs[-1]], "final latents after 49 Euler steps (oracle vs the unmodified reference, 7B width)")
        O.LINEAR_FP32_ACCUM = True
        try:
            ocache32 = O.forward_cache_update_text(W, cfg, O.OracleCache(L), **ogi)
            xs32, vs32 = euler(W, cfg, oli, ocache32, dict(ocfg, cache=O.OracleCache(L)), "oracle, fp32-accumulating linears")
        finally:
            O.LINEAR_FP32_ACCUM = False
    x0 = li["packed_init_noises"].float()
    curve = dict(x=[rel(a, b) for a, b in zip(xs32, xs)], displacement=[rel(a - x0, b - x0) for a, b in zip(xs32, xs)],
                 v=[rel(a, b) for a, b in zip(vs32, vs)],
                 x_rms=[float(b.float().pow(2).mean().sqrt()) for b in xs], v_rms=[float(b.float().pow(2).mean().sqrt()) for b in vs])
    print("noise floor of the reference's own arithmetic (fp32-accumulating restatement vs reference), rel-L2 after n Euler steps:")
    for n in (1, 3, 5, 10, 20, 30, 40, 49):
        print(f"  n = {n:2d}: x_t {curve['x'][n - 1]:.3e}  displacement {curve['displacement'][n - 1]:.3e}  v_t {curve['v'][n - 1]:.3e}  "
              f"|x_t| rms {curve['x_rms'][n - 1]:.3f}  |v_t| rms {curve['v_rms'][n - 1]:.3f}")
    kc, vc = MG.cache_to_lists(cache, L)
    out = dict(prompt=PROMPT, image_sizes=SIZES, prompt_inputs=gi, newlens=newlens, newrope=newrope, key_cache=kc, value_cache=vc,
               latent_inputs=li, cfg_inputs=ci, gen_kwargs=KW, latents=[lat[0].clone()], latents_f32acc=[xs32[-1].clone()],
               snap_steps=list(SNAP), snap_disp=[(xs[n - 1].float() - x0).to(torch.float16) for n in SNAP],
               snap_disp_f32acc=[(xs32[n - 1].float() - x0).to(torch.float16) for n in SNAP], noise_floor_curve=curve,
               snap_note="displacements x_n - x_0 in fp16 (exact to 2^-11 of the displacement)",
               host=dict(torch=torch.__version__, cpu_bf16_backend="mkldnn" if torch.backends.mkldnn.is_available() else "native"))
    path = os.path.join(MG.GOLD, "wide7b_traj49.pt")
    torch.save(out, path)
    print(f"wrote {path} ({os.path.getsize(path) / 1e6:.1f} MB) in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
