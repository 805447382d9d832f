"""Where the wall time of ONE batched generate_text call goes (BAGEL-7B-MoT, B requests on 4936-token contexts): session set-up (paged cache
allocation + adoption of the NaiveCache), step 0 (eager), graph capture, the replayed steps, the write-back into the caller's cache.
    python tools/decode_phase_probe.py [B=16] [new_tokens=160]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bagel_amd.factory import BAGEL_7B_MOT, NEW_TOKEN_IDS_QWEN25, build_bagel, init_random_  # noqa: E402
from bagel_amd.modeling.bagel.decode import DecodeSession  # noqa: E402
from bagel_amd.modeling.bagel.qwen2_navit import NaiveCache  # noqa: E402
import bench  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 160
    dev = torch.device("cuda", 0)
    cfg = BAGEL_7B_MOT
    model, _ = build_bagel(cfg, device=dev, with_vae=False)
    init_random_(model, seed=0)
    ids = NEW_TOKEN_IDS_QWEN25
    L = model.config.llm_config.num_hidden_layers
    image = torch.rand(3, 980, 980, generator=torch.Generator().manual_seed(2)) * 2 - 1
    tok = bench.FixedTokenizer(torch.randint(0, 151643, (32,), generator=torch.Generator().manual_seed(1)).tolist())

    def prefill():
        cache = NaiveCache(L)
        gi, lens, ropes = model.prepare_vit_images([0] * B, [0] * B, [image] * B, lambda t: t, ids)
        cache = model.forward_cache_update_vit(cache, **gi)
        gi, lens, ropes = model.prepare_prompts(lens, ropes, ["p"] * B, tok, ids)
        cache = model.forward_cache_update_text(cache, **gi)
        return cache, lens, ropes

    sync = torch.cuda.synchronize
    for rep in range(2):
        cache, lens, ropes = prefill()
        st = model.prepare_start_tokens(lens, ropes, ids)
        lm = model.language_model
        sync(); t0 = time.perf_counter()
        kv_lens = [int(x) for x in st["key_values_lens"].tolist()]
        sess = DecodeSession(lm.engine(check=True), lm.model.embed_tokens.weight.data, lm.lm_head.weight.data, cache, kv_lens,
                             st["packed_start_tokens"], st["packed_query_position_ids"], n)
        sync(); t1 = time.perf_counter()
        sess.step(None)
        sync(); t2 = time.perf_counter()
        sess.capture(include_advance=True)
        sync(); t3 = time.perf_counter()
        for _ in range(n - 1):
            sess.step(None)
        sync(); t4 = time.perf_counter()
        sess.write_back(cache)
        sync(); t5 = time.perf_counter()
        toks = sess.tokens_so_far()
        sync(); t6 = time.perf_counter()
        print(f"rep {rep}: B={B} n={n} ctx={kv_lens[0]} | session init {1e3 * (t1 - t0):.1f} ms | step 0 (eager) {1e3 * (t2 - t1):.1f} | capture {1e3 * (t3 - t2):.1f} | "
              f"{n - 1} graph steps {1e3 * (t4 - t3):.1f} = {1e3 * (t4 - t3) / (n - 1):.3f} ms/step | write_back {1e3 * (t5 - t4):.1f} | tokens {1e3 * (t6 - t5):.1f} | "
              f"total {1e3 * (t6 - t0):.1f} = {1e3 * (t6 - t0) / n:.3f} ms/step as the bench counts it; graph={sess.graph is not None}", flush=True)
        del sess, cache


if __name__ == "__main__":
    main()
