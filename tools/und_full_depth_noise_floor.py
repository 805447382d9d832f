"""How far does the REFERENCE'S OWN arithmetic move on the image-understanding path at full 7B DEPTH when only the summation order of its bf16 linears changes?

CPU only, ~40 GB of RAM, ~15 minutes on 8 cores:   python tools/und_full_depth_noise_floor.py [--image 980] [--threads 8]

BASELINE.json configs[1] for one request: 26-layer SigLIP (so400m width) on a 980^2 image (4900 patches) + connector + the 28-layer non-causal prefill of the
4902-token ViT block + the causal prefill of a 34-token prompt + ONE decode step, through the oracle (pinned bit for bit to the unmodified reference, oracle/README.md)
twice: with the reference's bf16 F.linear, and with oracle.LINEAR_FP32_ACCUM (same bf16 operands, same rounding points, an fp32 matmul: another summation order --
what a GPU does).  The rel-L2 distances between the two runs' per-layer K / V and first-step logits are the accumulation-order noise floor of the reference at this
depth; bench.py's ``understanding.parity_at_full_depth`` bounds (bench.UND_DEPTH_NOISE) are 1.5 x these.  Weights: the benchmark's random-init distribution
(bagel_amd.factory.init_random_) drawn on the CPU.  TEST / MEASUREMENT INFRASTRUCTURE: imports oracle/."""
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
    ap.add_argument("--image", type=int, default=980)
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    ap.add_argument("--layers", type=int, default=28)
    ap.add_argument("--vocab", type=int, default=8192, help="lm_head rows (the noise of a logit does not depend on how many there are)")
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    from bagel_amd.factory import BAGEL_7B_MOT, build_bagel, init_random_
    from oracle import bagel_oracle as O
    from oracle import packers as P
    cfg = dict(BAGEL_7B_MOT, llm=dict(BAGEL_7B_MOT["llm"], num_hidden_layers=a.layers, vocab_size=a.vocab))
    t0 = time.time()
    model, _ = build_bagel(cfg, device="cpu", with_vae=False)
    init_random_(model, seed=0)
    W = {k: v for k, v in model.state_dict().items() if k.startswith(("language_model.", "vit_model.", "connector.", "vit_pos_embed."))}
    print(f"{a.layers}-layer 7B-width weights + SigLIP drawn in {time.time() - t0:.0f} s", flush=True)
    L = a.layers
    ids = dict(bos_token_id=1, eos_token_id=2, start_of_image=3, end_of_image=4)
    image = torch.rand(3, a.image, a.image, generator=torch.Generator().manual_seed(2)) * 2 - 1

    class Tok:
        def encode(self, s):
            return torch.randint(8, a.vocab - 8, (32,), generator=torch.Generator().manual_seed(1)).tolist()
    ident = lambda t: t  # noqa: E731
    ti, l1, r1 = model.prepare_vit_images([0], [0], [image], ident, ids)
    pi, l2, r2 = model.prepare_prompts(l1, r1, ["p"], Tok(), ids)
    st = model.prepare_start_tokens(l2, r2, ids)
    runs = {}
    for tag, flag in (("fp32_accum_linear", True), ("bf16_linear", False)):
        O.LINEAR_FP32_ACCUM = flag
        try:
            t1 = time.time()
            cache = O.forward_cache_update_vit(W, cfg, O.OracleCache(L), **ti)
            cache = O.forward_cache_update_text(W, cfg, cache, **pi)
            kv = [(cache.key_cache[i].float().clone(), cache.value_cache[i].float().clone()) for i in range(L)]
            toks, logits = O.generate_text(W, cfg, cache, st["packed_key_value_indexes"], st["key_values_lens"], st["packed_start_tokens"],
                                           st["packed_query_position_ids"], 1, return_logits=True)
            runs[tag] = (kv, logits[0].float(), time.time() - t1)
            print(f"{tag}: SigLIP + {L}-layer prefill of {int(l2[0])} tokens + 1 decode step in {time.time() - t1:.0f} s", flush=True)
        finally:
            O.LINEAR_FP32_ACCUM = False
    rel = lambda p, q: float((p - q).norm() / q.norm())  # noqa: E731
    a32, b16 = runs["fp32_accum_linear"], runs["bf16_linear"]
    ek = [rel(a32[0][i][0], b16[0][i][0]) for i in range(L)]
    ev = [rel(a32[0][i][1], b16[0][i][1]) for i in range(L)]
    res = dict(layers=L, context=int(l2[0]), threads=a.threads, vocab=a.vocab, noise_floor_kv_max=max(ek + ev), noise_floor_k_by_layer=[round(e, 5) for e in ek],
               noise_floor_v_by_layer=[round(e, 5) for e in ev], noise_floor_first_step_logits=rel(a32[1], b16[1]),
               seconds_bf16=b16[2], seconds_fp32_accum=a32[2])
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
