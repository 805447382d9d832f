"""GPU check of the opt-in stream-batched CFG forward (model.cfg_batched) on the tiny golden configs:
  * batched, marker rows inside the tile GEMM (und_side_path=False): latents must equal the sequential path BIT FOR BIT
    (same kernels, every row's arithmetic unchanged -- only its tile position differs);
  * batched + marker-row side path: latents vs the REFERENCE golden within the written tolerance (2e-2; the marker rows go
    through the skinny GEMM, i.e. another accumulation order on 2 rows per sample) and the distance to the sequential path;
  * optional --bench: one denoise step at 7B shapes, sequential vs batched (ms / step).
Run on the GPU box: python tools/check_stream_batch.py [--bench]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.configs import TINY, TINY_D128, NEW_TOKEN_IDS_TINY, StubTokenizer  # noqa: E402
from tests.util_models import product_model  # noqa: E402
from tests.test_model_gpu import cfg_kwargs, new_cache, rel_l2  # noqa: E402


def golden(name):
    return torch.load(os.path.join(ROOT, "tests", "golden", name + ".pt"), weights_only=False)


def tiny():
    ok = True
    for cfg in (TINY, TINY_D128):
        g = golden(f"{cfg['name']}_t2i")
        model, _ = product_model(cfg)
        tok = StubTokenizer(cfg["llm"]["vocab_size"])
        gi, _, _ = model.prepare_prompts([0, 0], [0, 0], g["prompts"], tok, NEW_TOKEN_IDS_TINY)
        cache = model.forward_cache_update_text(new_cache(cfg), **gi)

        def run():
            return model.generate_image(past_key_values=cache, **cfg_kwargs("cfg_text", new_cache(cfg), g["cfg_inputs"]),
                                        **g["gen_kwargs"], **g["latent_inputs"])
        model.cfg_batched = False
        seq = run()
        model.cfg_batched, model.und_side_path = True, False
        bat = run()
        same = all(torch.equal(a, b) for a, b in zip(seq, bat))
        model.und_side_path = True
        side = run()
        e_gold = max(rel_l2(a, b) for a, b in zip(side, g["latents"]))
        e_seq = max(rel_l2(a, b) for a, b in zip(side, seq))
        e_seq_gold = max(rel_l2(a, b) for a, b in zip(seq, g["latents"]))
        print(f"{cfg['name']}: batched == sequential bit-for-bit: {same};  side path vs golden {e_gold:.3e} (sequential vs golden "
              f"{e_seq_gold:.3e}), side path vs sequential {e_seq:.3e}", flush=True)
        ok = ok and same and e_gold <= 2e-2
        model.cfg_batched = False
    return ok


def bench():
    from bagel_amd.factory import BAGEL_7B_MOT, NEW_TOKEN_IDS_QWEN25, build_bagel, init_random_
    from bagel_amd.modeling.bagel.qwen2_navit import NaiveCache
    cfg, dev = BAGEL_7B_MOT, torch.device("cuda", 0)
    model, _ = build_bagel(cfg, device=dev, with_vae=False)
    init_random_(model, seed=0)
    model.llm2vae.weight.data.normal_(0, cfg["llm"]["hidden_size"] ** -0.5, generator=torch.Generator(device=dev).manual_seed(1))
    L, B, R = cfg["llm"]["num_hidden_layers"], 4, 1024

    class Tok:
        def encode(self, s):
            return torch.randint(0, 151643, (30,), generator=torch.Generator().manual_seed(1)).tolist()
    ids = NEW_TOKEN_IDS_QWEN25
    gi, lens, ropes = model.prepare_prompts([0] * B, [0] * B, ["p"] * B, Tok(), ids)
    cache = model.forward_cache_update_text(NaiveCache(L), **gi)
    torch.manual_seed(42)
    li = model.prepare_vae_latent(lens, ropes, [(R, R)] * B, ids)
    ci = model.prepare_vae_latent_cfg([0] * B, [0] * B, [(R, R)] * B)
    res = {}
    modes = [("sequential", False, False), ("batched", True, False), ("batched+side", True, True)]
    for tag, batched, side in modes:
        model.cfg_batched, model.und_side_path = batched, side

        def run(T):
            return model.generate_image(past_key_values=cache, num_timesteps=T, cfg_text_scale=4.0, cfg_interval=[0, 1.0],
                                        cfg_renorm_min=0.0, cfg_renorm_type="global", timestep_shift=3.0,
                                        cfg_text_past_key_values=NaiveCache(L), cfg_text_packed_position_ids=ci["cfg_packed_position_ids"],
                                        cfg_text_packed_query_indexes=ci["cfg_packed_query_indexes"],
                                        cfg_text_key_values_lens=ci["cfg_key_values_lens"],
                                        cfg_text_packed_key_value_indexes=ci["cfg_packed_key_value_indexes"], **li)
        run(2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lat = run(6)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        res[tag] = (ms, lat)
        print(f"7B text->image B=4 1024^2, {tag}: {ms:.1f} ms per Euler step (2 forwards)  -> {4 / (49 * ms * 1e-3):.4f} images/s (denoise only)",
              flush=True)
    e = max(rel_l2(a, b) for a, b in zip(res["batched"][1], res["sequential"][1]))
    e2 = max(rel_l2(a, b) for a, b in zip(res["batched+side"][1], res["sequential"][1]))
    print(f"latents after 5 steps: batched vs sequential rel-L2 {e:.3e} (expected 0), batched+side vs sequential {e2:.3e}", flush=True)


if __name__ == "__main__":
    ok = tiny()
    if "--bench" in sys.argv:
        bench()
    sys.exit(0 if ok else 1)
