"""How far does the REFERENCE'S OWN arithmetic move at full 7B DEPTH when only the summation order of its bf16 linears changes?

CPU only, ~30 GB of RAM, a few minutes on 8 AMX cores:   python tools/full_depth_noise_floor.py [--layers 28] [--threads 8]

One Euler step of BASELINE.json configs[2] for ONE 1024^2 sample -- text prefill of the prompt, then the cond and the CFG-text
forward of all 28 MoT layers over 4098 tokens, CFG 4.0, global renorm (bagel.py:757-907) -- through the oracle (the CPU restatement
that is pinned bit-for-bit to the unmodified reference, oracle/README.md) twice: with the reference's bf16 F.linear, and with
oracle.LINEAR_FP32_ACCUM (same bf16 operands, same rounding points, an fp32 matmul: another summation order -- what a GPU, or another
CPU backend, does).  The rel-L2 distance between the two CFG-combined velocities is the accumulation-order noise floor of the
reference at this depth; bench.py's ``parity_at_full_depth`` tolerance (bench.FULL_DEPTH_TOL) is 1.5 x that floor, the rule
tests/test_wide_gpu.py froze at 2 layers.  Weights: the benchmark's random-init distribution (bagel_amd.factory.init_random_) drawn on
the CPU.  TEST / MEASUREMENT INFRASTRUCTURE: imports oracle/."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=28)
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    ap.add_argument("--resolution", type=int, default=1024)
    ap.add_argument("--prompt-tokens", type=int, default=30)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    from bagel_amd.factory import BAGEL_7B_MOT, build_bagel, init_random_
    from oracle import bagel_oracle as O
    from oracle import packers as P
    cfg = dict(BAGEL_7B_MOT, llm=dict(BAGEL_7B_MOT["llm"], num_hidden_layers=a.layers, vocab_size=512))
    t0 = time.time()
    model, _ = build_bagel(cfg, device="cpu", with_vae=False)
    init_random_(model, seed=0)
    H = cfg["llm"]["hidden_size"]
    model.llm2vae.weight.data.normal_(0, H ** -0.5, generator=torch.Generator().manual_seed(1))
    W = {k: v for k, v in model.state_dict().items() if not k.startswith(("vit_model.", "connector."))}
    print(f"{a.layers}-layer 7B-width weights drawn in {time.time() - t0:.0f} s", flush=True)
    L, R = a.layers, a.resolution
    ids = dict(bos_token_id=1, eos_token_id=2, start_of_image=3, end_of_image=4)

    class Tok:
        def encode(self, s):
            return torch.randint(8, 500, (a.prompt_tokens,), generator=torch.Generator().manual_seed(1)).tolist()
    ds = cfg["vae"]["downsample"] * cfg["bagel"]["latent_patch_size"]
    pdim = cfg["bagel"]["latent_patch_size"] ** 2 * cfg["vae"]["z_channels"]
    gi, lens, ropes = P.prepare_prompts([0], [0], ["p"], Tok(), ids)
    torch.manual_seed(42)
    li = P.prepare_vae_latent(lens, ropes, [(R, R)], ids, ds, cfg["bagel"]["max_latent_size"], pdim)
    ci = P.prepare_vae_latent_cfg([0], [0], [(R, R)], ds)
    x = li["packed_init_noises"]
    ts = torch.tensor([1.0] * x.shape[0])
    out = {}
    for tag, flag in (("bf16_linear", False), ("fp32_accum_linear", True)):
        O.LINEAR_FP32_ACCUM = flag
        try:
            t1 = time.time()
            cache = O.forward_cache_update_text(W, cfg, O.OracleCache(L), **gi)
            ocfg = dict(cache=O.OracleCache(L), position_ids=ci["cfg_packed_position_ids"], query_indexes=ci["cfg_packed_query_indexes"],
                        key_values_lens=ci["cfg_key_values_lens"], key_value_indexes=ci["cfg_packed_key_value_indexes"])
            v_cfg = O.forward_flow(W, cfg, x, ts, li, cache, ocfg, None, 4.0, 1.0, 0.0, "global")
            v_cond = O.forward_flow(W, cfg, x, ts, li, cache, None, None, 1.0, 1.0, 0.0, "global")
            out[tag] = (v_cfg.float(), v_cond.float(), time.time() - t1)
            print(f"{tag}: 3 forwards of {L} layers in {time.time() - t1:.0f} s; |v_cfg| rms {v_cfg.float().pow(2).mean().sqrt():.3f}, "
                  f"|v_cond| rms {v_cond.float().pow(2).mean().sqrt():.3f}", flush=True)
        finally:
            O.LINEAR_FP32_ACCUM = False
    rel = lambda p, q: float((p - q).norm() / q.norm())  # noqa: E731
    res = dict(layers=L, tokens=int(x.shape[0]) + 2, context=int(lens[0]), threads=a.threads,
               noise_floor_cfg_combined_velocity=rel(out["fp32_accum_linear"][0], out["bf16_linear"][0]),
               noise_floor_cond_velocity=rel(out["fp32_accum_linear"][1], out["bf16_linear"][1]),
               seconds_bf16=out["bf16_linear"][2], seconds_fp32_accum=out["fp32_accum_linear"][2])
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
