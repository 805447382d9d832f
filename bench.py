#!/usr/bin/env python
"""bench.py -- BAGEL-7B-MoT text->image throughput on MI355X (BASELINE.json metric: images/sec, 1024^2, 50-step).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one complete pass of the hot path over one batch of synthetic input on every rank:
  text prefill of the (shared) prompt -> [rank 0 computes, RCCL-broadcasts the conditioning KV] -> prepare_vae_latent ->
  generate_image (T=50 -> 49 Euler steps x [cond + CFG-text] forwards of the 28-layer MoT backbone at 4098 tokens/sample)
  -> VAE decode of every latent to a uint8 image.  Per-rank batch is fixed (weak scaling; config 3 = 4 samples/GPU,
  config 4 = the same over 8 GPUs).  Weights are random-init bf16 of the BAGEL-7B-MoT architecture (no checkpoint
  offline), inputs synthetic but resident in HBM before the timed region starts.

Prints ONE JSON line (rank 0) with the contract's fields plus:
  roofline      achieved MFMA TFLOP/s of the dominant kernel (the persistent ping-pong GEMM, gemm_pq_kernel<*>): algorithmic 2*M*N*K FLOPs of every launch
                in the timed region / their HIP-event durations (events on the launch stream), vs the 2.5 PFLOP/s bf16 peak;
  cpu_baseline  the oracle (CPU restatement of the reference) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md:42


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4, help="samples per GPU (config 3: 4)")
    ap.add_argument("--resolution", type=int, default=1024)
    ap.add_argument("--num-timesteps", type=int, default=50)
    ap.add_argument("--prompt-tokens", type=int, default=30)
    ap.add_argument("--layers", type=int, default=None, help="debug only: fewer LLM layers (result flagged invalid)")
    ap.add_argument("--no-vae", action="store_true", help="debug only: skip the VAE decode (result flagged invalid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-layers", type=int, default=1)
    ap.add_argument("--no-full-depth", action="store_true",
                    help="cpu_baseline: skip the full-depth Euler step through the oracle (~3 min of host time) and its parity check")
    ap.add_argument("--cpu-port", action="store_true", help="cpu_baseline: time the oracle port even where /root/reference exists")
    ap.add_argument("--cpu-baseline-only", action="store_true",
                    help="no GPU needed: time the CPU baseline on this host (the unmodified reference where /root/reference exists) and print it")
    ap.add_argument("--workload", choices=["t2i", "edit"], default="t2i",
                    help="t2i = BASELINE configs[2]/[3] (the headline metric); edit = configs[4] image-edit (VAE enc + ViT + 3-forward CFG)")
    ap.add_argument("--no-taylorseer", action="store_true", help="skip the extra enable_taylorseer=True measurement")
    ap.add_argument("--no-understanding", action="store_true", help="skip the configs[1] leg (ViT prefill + text decode)")
    ap.add_argument("--no-batched-decode", action="store_true", help="understanding leg: skip the extra 16-request batched decode")
    ap.add_argument("--no-train-forward", action="store_true", help="skip the extra training-forward (Bagel.forward, losses only) measurement")
    ap.add_argument("--no-train-step", action="store_true", help="skip the training-step leg (forward with tape + backward) inside training_forward")
    ap.add_argument("--no-int8", action="store_true", help="understanding leg: skip the extra weight_quant='int8_rowwise' / 'mxfp4' decodes")
    ap.add_argument("--no-fp8", action="store_true", help="skip the extra gen_weight_quant='fp8' measurement")
    ap.add_argument("--no-edit", action="store_true", help="skip the extra configs[4] measurement (one image-edit request per GPU)")
    ap.add_argument("--weight-store", choices=["nf4", "int8_rowwise"], default=None,
                    help="option that changes results (line flagged invalid): the reference's quantised load modes (app.py:114-131) over the WHOLE forward "
                         "path -- Bagel.quantize_language_model before the timed region; skips the fp8 / training / understanding legs")
    ap.add_argument("--only-understanding", action="store_true", help="debug only: skip the text->image leg (result flagged invalid)")
    ap.add_argument("--und-new-tokens", type=int, default=256)
    ap.add_argument("--und-batch", type=int, default=1, help="requests decoded together per GPU (reference: 1, bagel.py:996)")
    ap.add_argument("--und-image", type=int, default=980, help="side of the understanding image (980 -> 4900 ViT tokens)")
    ap.add_argument("--standins", action="store_true",
                    help="TEST ONLY (tests/test_parallel_cpu.py): run main()'s own one_step() / fence / timing / JSON line on the CPU with the torch "
                         "stand-ins of tests/mock_ops.py in place of the launch wrappers, a tiny model and the gloo backend; the line is flagged invalid")
    ap.add_argument("--launch-check", action="store_true",
                    help="debug only: rendezvous + barrier + max-over-ranks timing + the JSON line, no model (gloo when there is no GPU)")
    return ap.parse_args()


def self_launch(args):
    """``python bench.py --gpus N`` outside torchrun: re-exec this file as N ranks (one process per GPU) under
    torch.distributed.run on 127.0.0.1, which is how gen_images_mp.py is started (scripts/eval/run_geneval.sh:10-16).
    Rank 0 of the child job prints the JSON line; its exit code is ours."""
    import socket
    import subprocess
    if torch.cuda.is_available() and torch.cuda.device_count() < args.gpus and not args.launch_check:
        raise SystemExit(f"--gpus {args.gpus} but only {torch.cuda.device_count()} GPU(s) are visible")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def launch_check(args, rank, world, local):
    """The distributed skeleton of main() without the model: process group, fence, timed region, MAX over ranks, one line."""
    import torch.distributed as dist
    cuda = torch.cuda.is_available() and torch.cuda.device_count() > local
    dev = torch.device("cuda", local) if cuda else torch.device("cpu")
    if cuda:
        torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl" if cuda else "gloo", rank=rank, world_size=world, **({"device_id": dev} if cuda else {}))

    def fence():
        if cuda:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
    fence()
    t0 = time.perf_counter()
    x = torch.full((1 << 20,), float(rank + 1), device=dev)
    for _ in range(args.steps):
        if world > 1:
            dist.all_reduce(x)
            x /= world
    fence()
    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    ranks = torch.ones(1, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(ranks)
    if rank == 0:
        print(json.dumps({"metric": "launch check (no model)", "valid": False, "value": 0.0, "unit": "images/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": float(tt.item()) / max(args.steps, 1) * 1e3,
                          "backend": ("nccl(RCCL)" if cuda else "gloo") if world > 1 else None, "ranks_seen": int(ranks.item()),
                          "config": {"workload": "launch check", "global_batch": world * args.batch, "parallelism": f"dp{world}"}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


class FixedTokenizer:
    """The prompt is a fixed list of token ids (no vocabulary files offline)."""

    def __init__(self, ids):
        self.ids = ids

    def encode(self, s):
        return list(self.ids)


def gemm_profile_hook():
    """Wrap ops.gemm so every launch in the timed region is bracketed by HIP events on its own stream."""
    from bagel_amd import ops
    records = []
    orig = ops.gemm

    def timed(A, W0, C, **kw):
        M = (kw.get("M0") if kw.get("M0") is not None else (kw["a_rows0"].numel() if kw.get("a_rows0") is not None else A.shape[0])) + kw.get("M1", 0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig(A, W0, C, **kw)
        e1.record()
        v = kw.get("variant")
        if v is None:
            v = ops.default_gemm_variant(M, W0.shape[0], W0.shape[1])
        records.append((2.0 * M * W0.shape[0] * W0.shape[1], e0, e1, v))
        return out
    return records, orig, timed


def attn_profile_hook():
    """The same for ops.attn_planned (the persistent attention kernel + its combine pass): algorithmic FLOPs 4 * Lq * Lkv * nq * D per sample
    (causal: the visible half of the new segment) over the HIP-event duration of every launch in the timed region."""
    from bagel_amd import ops
    records = []
    orig = ops.attn_planned

    def flops(ap):
        f = getattr(ap, "_bench_flops", None)
        if f is None:
            f = ap._bench_flops = sum(4.0 * lq * (lq * (0.5 if ap.causal else 1.0) + c) * ap.nq * ap.head_dim for lq, c in zip(ap.q_len, ap.ctx_len))
        return f

    def timed(q, k_new, vt_new, out, aplan, softmax_scale, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig(q, k_new, vt_new, out, aplan, softmax_scale, **kw)
        e1.record()
        records.append((flops(aplan), e0, e1, len(aplan.q_len)))
        return r
    return records, orig, timed


def physical_cores():
    """Physical cores of the host (SMT siblings excluded): what a bf16 GEMM-bound CPU run should use as its thread team."""
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


LAYER_FLOP_PER_TOKEN = 2.0 * 233046016          # linear MACs per token per MoT layer x 2


def _layer_flops(Lq, C, H=3584):
    return Lq * LAYER_FLOP_PER_TOKEN + 4.0 * Lq * (Lq + C) * H


def cpu_reference_layer(args, cfg, threads):
    """kind = "reference": the UNMODIFIED reference classes (only where /root/reference exists, i.e. the build container) through
    the SURVEY 8c harness: a 7B-WIDTH model of ``cpu_layers`` MoT layers, text prefill, then ONE Euler step of generate_image
    (cond + CFG-text forward) -- once untimed (warm-up), once timed.  Returns seconds per layer-forward."""
    from oracle import make_golden as MG
    from oracle.configs import NEW_TOKEN_IDS_TINY, StubTokenizer
    nl = args.cpu_layers
    c = dict(cfg, name="cpu_ref", llm=dict(cfg["llm"], vocab_size=512, num_hidden_layers=nl), vit=MG.TINY["vit"], vae=MG.TINY["vae"],
             bagel=MG.TINY["bagel"])
    model, _, _, _ = MG.build(c)
    from modeling.bagel.qwen2_navit import NaiveCache
    tok = StubTokenizer(512)
    R = args.resolution
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        gi, lens, ropes = model.prepare_prompts([0], [0], ["x" * args.prompt_tokens], tok, NEW_TOKEN_IDS_TINY)
        cache = model.forward_cache_update_text(NaiveCache(nl), **gi)
        torch.manual_seed(42)
        li = model.prepare_vae_latent(lens, ropes, [(R, R)], NEW_TOKEN_IDS_TINY)
        ci = model.prepare_vae_latent_cfg([0], [0], [(R, R)])

        def step():
            return model.generate_image(
                past_key_values=cache, cfg_text_past_key_values=NaiveCache(nl), num_timesteps=2, timestep_shift=3.0, cfg_text_scale=4.0,
                cfg_interval=[0.0, 1.0], cfg_renorm_min=0.0, cfg_renorm_type="global",
                cfg_text_packed_position_ids=ci["cfg_packed_position_ids"], cfg_text_packed_query_indexes=ci["cfg_packed_query_indexes"],
                cfg_text_key_values_lens=ci["cfg_key_values_lens"], cfg_text_packed_key_value_indexes=ci["cfg_packed_key_value_indexes"], **li)
        step()                                      # warm-up: oneDNN primitive caches, thread team, page faults
        t0 = time.time()
        step()
        dt = time.time() - t0
    return dt / (2 * nl)


def cpu_port_layer(args, cfg, threads, keep=None):
    """kind = "port": the oracle's MoT decoder layer (gen mode, one 1024^2 sample on the prompt context) at 7B shapes -- one warm-up
    pass, then timed passes.  Returns seconds per layer-forward; ``keep`` receives what the full-size parity check needs."""
    from oracle import bagel_oracle as O
    llm = cfg["llm"]
    H, I, nh, nkv = llm["hidden_size"], llm["intermediate_size"], llm["num_attention_heads"], llm["num_key_value_heads"]
    hd = H // nh
    g = torch.Generator().manual_seed(0)
    W = {}
    nl = args.cpu_layers
    for li in range(nl):
        p = f"language_model.model.layers.{li}."
        for suf in ("", "_moe_gen"):
            for n, shp in (("q_proj", (nh * hd, H)), ("k_proj", (nkv * hd, H)), ("v_proj", (nkv * hd, H)), ("o_proj", (H, nh * hd))):
                W[p + f"self_attn.{n}{suf}.weight"] = (torch.randn(shp, generator=g) * shp[1] ** -0.5).to(torch.bfloat16)
                if n != "o_proj":
                    W[p + f"self_attn.{n}{suf}.bias"] = (torch.randn(shp[0], generator=g) * 0.02).to(torch.bfloat16)
            for n in ("q_norm", "k_norm"):
                W[p + f"self_attn.{n}{suf}.weight"] = torch.ones(hd, dtype=torch.bfloat16)
            for n, shp in (("gate_proj", (I, H)), ("up_proj", (I, H)), ("down_proj", (H, I))):
                W[p + f"mlp{suf}.{n}.weight"] = (torch.randn(shp, generator=g) * shp[1] ** -0.5).to(torch.bfloat16)
            for n in ("input_layernorm", "post_attention_layernorm"):
                W[p + f"{n}{suf}.weight"] = torch.ones(H, dtype=torch.bfloat16)
    n_img = (args.resolution // 16) ** 2
    Lq, C = n_img + 2, args.prompt_tokens + 2
    x0 = torch.randn(Lq, H, generator=g).to(torch.bfloat16)
    cache = O.OracleCache(nl)
    for li in range(nl):
        cache.key_cache[li] = torch.randn(C, nkv, hd, generator=g).to(torch.bfloat16)
        cache.value_cache[li] = torch.randn(C, nkv, hd, generator=g).to(torch.bfloat16)
    text_idx = torch.tensor([0, Lq - 1])
    vae_idx = torch.arange(1, Lq - 1)
    cos_sin = O.rope_tables(torch.full((Lq,), C, dtype=torch.long), hd, llm["rope_theta"], torch.bfloat16)
    qlens, kvlens = torch.tensor([Lq], dtype=torch.int), torch.tensor([C], dtype=torch.int)
    q_idx, kv_idx = torch.arange(C, C + Lq), torch.arange(C)

    def run():
        x = x0
        for li in range(nl):
            x = O.mot_layer(W, llm, li, x, qlens, cos_sin, q_idx, cache, kvlens, kv_idx, False, False, "gen", vae_idx, text_idx)
        return x
    run()                                           # warm-up
    reps = 2
    t0 = time.time()
    for _ in range(reps):
        x = run()
    dt = (time.time() - t0) / (reps * nl)
    if keep is not None:
        keep.update(W=W, x0=x0, x=x, cache=cache, q_idx=q_idx, kv_idx=kv_idx, text_idx=text_idx, vae_idx=vae_idx, Lq=Lq, C=C)
    return dt


def full_size_parity(cfg, nl, k):
    """The same layer(s), same weights and inputs, through the HIP engine (checker use of the oracle)."""
    from bagel_amd.factory import build_bagel
    from bagel_amd.modeling.bagel.qwen2_navit import NaiveCache
    llm = cfg["llm"]
    nkv, hd = llm["num_key_value_heads"], llm["hidden_size"] // llm["num_attention_heads"]
    Lq, C = k["Lq"], k["C"]
    dev = torch.device("cuda", torch.cuda.current_device())
    m1, _ = build_bagel(cfg, device=dev, num_layers=nl, with_vae=False)
    m1.load_state_dict({n: v for n, v in k["W"].items()}, strict=False)
    eng = m1.language_model.engine()
    c1 = NaiveCache(nl)
    for li in range(nl):
        c1.store(li, k["cache"].key_cache[li].reshape(C, nkv * hd).to(dev), k["cache"].value_cache[li].reshape(C, nkv * hd).to(dev), [C], [0],
                 nkv, hd, eng.dp)
    plan = eng.plan([Lq], torch.full((Lq,), C, dtype=torch.long), packed_query_indexes=k["q_idx"], key_values_lens=[C],
                    packed_key_value_indexes=k["kv_idx"], text_indexes=k["text_idx"], vae_indexes=k["vae_idx"])
    y = eng.forward(k["x0"].to(dev), plan, "gen", c1, update=False, causal=False, num_layers=nl, final_norm=False)
    torch.cuda.synchronize()
    yc, xr = y.float().cpu(), k["x"].float()
    return {"what": f"residual stream after {nl} MoT layer(s), {Lq} tokens, 7B shapes: HIP engine vs oracle on identical weights/inputs",
            "rel_l2": float((yc - xr).norm() / xr.norm()), "max_abs": float((yc - xr).abs().max()), "ref_max_abs": float(xr.abs().max())}


# parity_at_full_depth tolerances = 1.5 x the REFERENCE'S OWN accumulation-order noise at 28 layers (the rule tests/test_wide_gpu.py froze at 2
# layers): tools/full_depth_noise_floor.py re-runs the Euler step through the oracle with fp32-accumulating linears (same bf16 operands and
# rounding points, another summation order) -- profiles/r03_full_depth_noise_floor.log: single-forward velocity 1.6e-2, CFG-combined velocity
# 8.0e-2 (CFG 4.0 + global renorm amplify the difference of two forwards ~5x).  Measured on MI355X (round 3): 1.5e-2 / 6.7e-2.
FULL_DEPTH_TOL = 0.12            # CFG-combined velocity
FULL_DEPTH_TOL_FORWARD = 0.024   # velocity of ONE forward (cond, or CFG-text)
FULL_DEPTH_TOL_COMBINE = 0.005   # the combine / renorm / Euler step fed with the product's own forwards, sequential execution: it differs from the oracle's combine
#                                  by rounding points only (measured 0.0: bit-identical).  The stream-batched path runs the forwards as ONE batch -- another attention
#                                  work list, its own accumulation order -- so against the same self-combine it shows one forward's noise x the CFG amplification
#                                  (3.2e-2 at 4 layers) and is held to the CFG-combined band FULL_DEPTH_TOL


def full_depth_step(args, cfg, model, tok, ids, threads, layers=None):
    """ONE full Euler step of the benchmark workload at 7B DEPTH for ONE 1024^2 sample -- text prefill of the prompt, then the cond and
    the CFG-text forward through every MoT layer, CFG 4.0, global renorm (bagel.py:757-907) -- through the oracle on this box's host
    cores WITH THE GPU MODEL'S OWN WEIGHTS (copied off the device), and through the HIP engine on the same inputs: the sequential
    ``_forward_flow`` and the default stream-batched path inside ``generate_image(num_timesteps=2)`` (one step, dt = 1, so
    v = x_0 - x_1).  Returns the measured CPU time of the step (SURVEY.md 8d ii: the measured slice of the CPU baseline) and the rel-L2
    of the CFG-combined velocity.  Checker use of the oracle only."""
    from oracle import bagel_oracle as O
    from bagel_amd.modeling.bagel.qwen2_navit import NaiveCache
    L = model.config.llm_config.num_hidden_layers if layers is None else layers
    R = args.resolution
    torch.set_num_threads(threads)
    t0 = time.time()
    keep = ("language_model.model.", "time_embedder.", "vae2llm.", "llm2vae.", "latent_pos_embed.")
    W = {}
    for k, v in model.state_dict().items():
        if k.startswith(keep) and not (k.startswith("language_model.model.layers.") and int(k.split(".")[3]) >= L):
            W[k] = v.detach().to("cpu")
    t_copy = time.time() - t0
    ocfg_model = dict(cfg, llm=dict(cfg["llm"], num_hidden_layers=L))
    gi, lens, ropes = model.prepare_prompts([0], [0], ["p"], tok, ids)
    li = model.prepare_vae_latent(lens, ropes, [(R, R)], ids)
    x0 = torch.randn(li["packed_init_noises"].shape, generator=torch.Generator().manual_seed(4242))
    li["packed_init_noises"] = x0
    ci = model.prepare_vae_latent_cfg([0], [0], [(R, R)])
    ts = torch.tensor([1.0] * x0.shape[0])
    t1 = time.time()
    ocache = O.forward_cache_update_text(W, ocfg_model, O.OracleCache(L), **gi)
    t_prefill = time.time() - t1
    ocfg = dict(cache=O.OracleCache(L), position_ids=ci["cfg_packed_position_ids"], query_indexes=ci["cfg_packed_query_indexes"],
                key_values_lens=ci["cfg_key_values_lens"], key_value_indexes=ci["cfg_packed_key_value_indexes"])
    t1 = time.time()
    parts = {}
    v_cpu = O.forward_flow(W, ocfg_model, x0, ts, li, ocache, ocfg, None, 4.0, 1.0, 0.0, "global", parts=parts).float()
    t_step = time.time() - t1
    del W
    # the HIP engine on the same weights and inputs
    cache = model.forward_cache_update_text(NaiveCache(model.config.llm_config.num_hidden_layers), **gi)
    kv_err = max(float(((cache.key_cache[i].float().cpu() - ocache.key_cache[i].float()).norm() / ocache.key_cache[i].float().norm())) for i in range(L))
    ckw = dict(cfg_text_past_key_values=NaiveCache(model.config.llm_config.num_hidden_layers),
               cfg_text_packed_position_ids=ci["cfg_packed_position_ids"], cfg_text_packed_query_indexes=ci["cfg_packed_query_indexes"],
               cfg_text_key_values_lens=ci["cfg_key_values_lens"], cfg_text_packed_key_value_indexes=ci["cfg_packed_key_value_indexes"])
    rel = lambda a, b: float((a - b).norm() / b.norm())  # noqa: E731
    out = {"what": f"one Euler step (t = 1): cond + CFG-text forward of {L} MoT layers over {x0.shape[0] + 2} tokens on a {lens[0]}-token context, CFG 4.0, "
                   f"global renorm, 7B shapes, identical weights and inputs: CFG-combined velocity, HIP engine vs oracle (rel-L2)",
           "layers": L, "cpu_seconds_per_euler_step": t_step, "cpu_seconds_prefill": t_prefill, "weights_copy_seconds": t_copy, "threads": threads,
           "prefill_kv_rel_l2_max": kv_err, "v_rms": float(v_cpu.pow(2).mean().sqrt()), "tolerance": FULL_DEPTH_TOL}
    if L == model.config.llm_config.num_hidden_layers:
        lkw = {k: v for k, v in li.items() if k != "packed_init_noises"}
        model.language_model.model.enable_taylorseer = False
        v_seq = model._forward_flow(x_t=x0, timestep=ts, past_key_values=cache, cfg_text_scale=4.0, cfg_renorm_type="global", **ckw, **lkw)
        out["rel_l2_sequential_forward_flow"] = rel(v_seq.float().cpu(), v_cpu)
        # the two forwards on their own (no CFG amplification): cond on the prompt context, CFG-text without context
        v_c = model._forward_flow(x_t=x0, timestep=ts, past_key_values=cache, cfg_text_scale=1.0, cfg_renorm_type="global", **lkw)
        out["rel_l2_cond_forward"] = rel(v_c.float().cpu(), parts["v_cond"].float())
        lkw2 = dict(lkw, packed_position_ids=ci["cfg_packed_position_ids"], packed_indexes=ci["cfg_packed_query_indexes"],
                    key_values_lens=ci["cfg_key_values_lens"], packed_key_value_indexes=ci["cfg_packed_key_value_indexes"])
        v_u = model._forward_flow(x_t=x0, timestep=ts, past_key_values=NaiveCache(model.config.llm_config.num_hidden_layers), cfg_text_scale=1.0,
                                  cfg_renorm_type="global", **lkw2)
        out["rel_l2_cfg_text_forward"] = rel(v_u.float().cpu(), parts["v_cfg_text"].float())
        out["tolerance_forward"] = FULL_DEPTH_TOL_FORWARD
        lat = model.generate_image(past_key_values=cache, num_timesteps=2, cfg_text_scale=4.0, cfg_interval=[0, 1.0], cfg_renorm_min=0.0,
                                   cfg_renorm_type="global", timestep_shift=3.0, **ckw, **li)
        v_gpu = x0 - torch.cat([t.float().cpu() for t in lat])
        out["rel_l2"] = rel(v_gpu, v_cpu)
        # the CFG combine / renorm / stream-batching path on its OWN inputs: the oracle's combine of the product's two single-forward velocities
        # against what the product's combined paths returned -- no forward noise in this comparison, so it is gated tightly
        v_self = O.cfg_combine(v_c.cpu(), v_u.cpu(), None, 4.0, 1.0, 0.0, "global").float()
        try:
            model.cfg_batched = False
            lat_s = model.generate_image(past_key_values=cache, num_timesteps=2, cfg_text_scale=4.0, cfg_interval=[0, 1.0], cfg_renorm_min=0.0,
                                         cfg_renorm_type="global", timestep_shift=3.0, **ckw, **li)
        finally:
            model.cfg_batched = True
        v_gpu_s = x0 - torch.cat([t.float().cpu() for t in lat_s])
        out["cfg_combine_self_consistency"] = {"sequential_forward_flow": rel(v_seq.float().cpu(), v_self), "generate_image_sequential": rel(v_gpu_s, v_self),
                                               "generate_image_stream_batched": rel(v_gpu, v_self), "tolerance": FULL_DEPTH_TOL_COMBINE,
                                               "tolerance_stream_batched": FULL_DEPTH_TOL,
                                               "what": "oracle cfg_combine(product's cond velocity, product's CFG-text velocity) vs the product's own combined velocity; the "
                                                       "stream-batched path runs both forwards as ONE batch (another attention work list: its own accumulation order), so it "
                                                       "is held to the CFG-combined noise band, the sequential paths to the combine's rounding"}
        out["path"] = "generate_image(num_timesteps=2): the default stream-batched cond + CFG forward with the marker-row side path"
    else:
        # depth-reduced run (tests): the engine stops after L layers, no final norm / llm2vae -- compare the residual stream instead
        raise NotImplementedError("full_depth_step compares whole-model velocities: build the model with the depth to test")
    # THE GATE is the per-forward bound (round-3 verdict): the CFG-combined figure amplifies the difference of two forwards ~5x on random-init
    # weights and a 12 % band would pass a real bug of that size -- it is reported (with whether it sits inside its own noise band), not gated on
    sc = out["cfg_combine_self_consistency"]
    out["within_tolerance"] = bool(out["rel_l2_cond_forward"] <= FULL_DEPTH_TOL_FORWARD and out["rel_l2_cfg_text_forward"] <= FULL_DEPTH_TOL_FORWARD
                                   and sc["sequential_forward_flow"] <= FULL_DEPTH_TOL_COMBINE and sc["generate_image_sequential"] <= FULL_DEPTH_TOL_COMBINE
                                   and sc["generate_image_stream_batched"] <= FULL_DEPTH_TOL)
    out["gate"] = (f"rel_l2_cond_forward and rel_l2_cfg_text_forward <= {FULL_DEPTH_TOL_FORWARD} (1.5 x the reference's own single-forward accumulation-order "
                   f"noise) AND the combine step on the product's own two forwards <= {FULL_DEPTH_TOL_COMBINE}; the CFG-combined rel_l2 vs the oracle is information only")
    out["cfg_combined_inside_noise_band"] = bool(out["rel_l2"] <= FULL_DEPTH_TOL and out["rel_l2_sequential_forward_flow"] <= FULL_DEPTH_TOL)
    out["noise_floor"] = {"cfg_combined_velocity": 0.080, "single_forward_velocity": 0.016, "source": "profiles/r03_full_depth_noise_floor.log"}
    return out


EDIT_DEPTH_TOL_BATCHED = 0.25     # three forwards, CFG 4.0 x 2.0: the single-forward noise is amplified ~2x further than in the two-forward step


def edit_depth_step(args, cfg, model, ids, threads, ctx_tokens=(576, 30)):
    """The THREE-forward Euler step of an image-edit request (app.py:224-228 defaults; bagel.py:854-905): cond forward on [image context | prompt], CFG-text
    forward on [image context] (a prefix of the cond context, as ``copy.deepcopy(gen_context)`` before the prompt makes it, inferencer.py:230-253), CFG-img
    forward on [prompt] alone, cfg_text_scale 4.0, cfg_img_scale 2.0, ``text_channel`` renorm -- through the oracle on this box's host cores WITH THE GPU MODEL'S
    OWN WEIGHTS and through the HIP engine (sequential ``_forward_flow`` and the stream-batched three-forward batch inside ``generate_image``).  The image context
    is stood in for by ``ctx_tokens[0]`` text tokens (the contexts may be shortened; what is under test is the denoise step on three different contexts, not the
    encoders that filled them).  Gates like ``full_depth_step``: every single forward within FULL_DEPTH_TOL_FORWARD of the oracle, and the combine step on the
    product's own three velocities within FULL_DEPTH_TOL_COMBINE.  Checker use of the oracle only."""
    import copy
    from oracle import bagel_oracle as O
    from bagel_amd.modeling.bagel.qwen2_navit import NaiveCache
    L = model.config.llm_config.num_hidden_layers
    R = args.resolution
    torch.set_num_threads(threads)
    keep = ("language_model.model.", "time_embedder.", "vae2llm.", "llm2vae.", "latent_pos_embed.")
    W = {k: v.detach().to("cpu") for k, v in model.state_dict().items() if k.startswith(keep)}
    cfg = dict(cfg, llm=dict(cfg["llm"], num_hidden_layers=L))          # (a depth-reduced model in the tests: the oracle walks the layers the model has)
    g = torch.Generator().manual_seed(11)
    V = cfg["llm"]["vocab_size"]
    tok_img = FixedTokenizer(torch.randint(8, V - 8, (ctx_tokens[0],), generator=g).tolist())
    tok_txt = FixedTokenizer(torch.randint(8, V - 8, (ctx_tokens[1],), generator=g).tolist())
    # contexts: product and oracle side by side (the packers are bit-exact, so one set of inputs feeds both)
    gi1, l1, r1 = model.prepare_prompts([0], [0], ["image"], tok_img, ids)
    gi2, l2, r2 = model.prepare_prompts(l1, r1, ["prompt"], tok_txt, ids)
    gi3, l3, r3 = model.prepare_prompts([0], [0], ["prompt"], tok_txt, ids)
    cache = model.forward_cache_update_text(NaiveCache(L), **gi1)
    cfg_text_cache = copy.deepcopy(cache)
    cache = model.forward_cache_update_text(cache, **gi2)
    cfg_img_cache = model.forward_cache_update_text(NaiveCache(L), **gi3)
    t1 = time.time()
    ocache = O.forward_cache_update_text(W, cfg, O.OracleCache(L), **gi1)
    octext = copy.deepcopy(ocache)
    ocache = O.forward_cache_update_text(W, cfg, ocache, **gi2)
    ocimg = O.forward_cache_update_text(W, cfg, O.OracleCache(L), **gi3)
    li = model.prepare_vae_latent(l2, r2, [(R, R)], ids)
    x0 = torch.randn(li["packed_init_noises"].shape, generator=torch.Generator().manual_seed(4343))
    li["packed_init_noises"] = x0
    ct = model.prepare_vae_latent_cfg(l1, r1, [(R, R)])
    cim = model.prepare_vae_latent_cfg(l3, r3, [(R, R)])
    ts = torch.tensor([1.0] * x0.shape[0])
    od = lambda c, d: dict(cache=c, position_ids=d["cfg_packed_position_ids"], query_indexes=d["cfg_packed_query_indexes"],  # noqa: E731
                           key_values_lens=d["cfg_key_values_lens"], key_value_indexes=d["cfg_packed_key_value_indexes"])
    v_cpu = O.forward_flow(W, cfg, x0, ts, li, ocache, od(octext, ct), od(ocimg, cim), 4.0, 2.0, 0.0, "text_channel").float()
    # the three single forwards of the oracle (scale 1.0 = no combine), each on its own context
    o_c = O.forward_flow(W, cfg, x0, ts, li, ocache, None, None, 1.0, 1.0, 0.0, "global").float()
    lis = lambda d: dict(li, packed_position_ids=d["cfg_packed_position_ids"], packed_indexes=d["cfg_packed_query_indexes"],  # noqa: E731
                         key_values_lens=d["cfg_key_values_lens"], packed_key_value_indexes=d["cfg_packed_key_value_indexes"])
    o_t = O.forward_flow(W, cfg, x0, ts, lis(ct), octext, None, None, 1.0, 1.0, 0.0, "global").float()
    o_i = O.forward_flow(W, cfg, x0, ts, lis(cim), ocimg, None, None, 1.0, 1.0, 0.0, "global").float()
    t_cpu = time.time() - t1
    del W
    rel = lambda a, b: float((a.float().cpu() - b).norm() / b.norm())  # noqa: E731
    kw = {}
    for tag, c, d in (("cfg_text", cfg_text_cache, ct), ("cfg_img", cfg_img_cache, cim)):
        kw.update({f"{tag}_past_key_values": c, f"{tag}_packed_position_ids": d["cfg_packed_position_ids"], f"{tag}_packed_query_indexes": d["cfg_packed_query_indexes"],
                   f"{tag}_key_values_lens": d["cfg_key_values_lens"], f"{tag}_packed_key_value_indexes": d["cfg_packed_key_value_indexes"]})
    lkw = {k: v for k, v in li.items() if k != "packed_init_noises"}
    model.language_model.model.enable_taylorseer = False
    one = lambda c, l_: model._forward_flow(x_t=x0, timestep=ts, past_key_values=c, cfg_text_scale=1.0, cfg_renorm_type="global", **l_)  # noqa: E731
    v_c, v_t, v_i = one(cache, lkw), one(cfg_text_cache, {k: v for k, v in lis(ct).items() if k != "packed_init_noises"}), \
        one(cfg_img_cache, {k: v for k, v in lis(cim).items() if k != "packed_init_noises"})
    v_seq = model._forward_flow(x_t=x0, timestep=ts, past_key_values=cache, cfg_text_scale=4.0, cfg_img_scale=2.0, cfg_renorm_min=0.0,
                                cfg_renorm_type="text_channel", **kw, **lkw)
    lat = model.generate_image(past_key_values=cache, num_timesteps=2, cfg_text_scale=4.0, cfg_img_scale=2.0, cfg_interval=[0, 1.0], cfg_renorm_min=0.0,
                               cfg_renorm_type="text_channel", timestep_shift=3.0, **kw, **li)
    v_gpu = x0 - torch.cat([t.float().cpu() for t in lat])
    v_self = O.cfg_combine(v_c.cpu(), v_t.cpu(), v_i.cpu(), 4.0, 2.0, 0.0, "text_channel").float()
    out = {"what": f"one Euler step (t = 1) of an image-edit request: cond / CFG-text / CFG-img forwards of {L} MoT layers over {x0.shape[0] + 2} tokens on "
                   f"{int(l2[0])} / {int(l1[0])} / {int(l3[0])}-token contexts, CFG 4.0 / 2.0, text_channel renorm, 7B shapes, identical weights and inputs",
           "layers": L, "contexts": [int(l2[0]), int(l1[0]), int(l3[0])], "cpu_seconds": t_cpu, "threads": threads,
           "rel_l2_cond_forward": rel(v_c, o_c), "rel_l2_cfg_text_forward": rel(v_t, o_t), "rel_l2_cfg_img_forward": rel(v_i, o_i),
           "rel_l2_combined_sequential": rel(v_seq, v_cpu), "rel_l2_combined_stream_batched": float((v_gpu - v_cpu).norm() / v_cpu.norm()),
           "cfg_combine_self_consistency": {"sequential_forward_flow": rel(v_seq, v_self), "generate_image_stream_batched": float((v_gpu - v_self).norm() / v_self.norm())},
           "tolerance_forward": FULL_DEPTH_TOL_FORWARD, "tolerance_combine": FULL_DEPTH_TOL_COMBINE, "tolerance_stream_batched": EDIT_DEPTH_TOL_BATCHED}
    sc = out["cfg_combine_self_consistency"]
    out["within_tolerance"] = bool(max(out["rel_l2_cond_forward"], out["rel_l2_cfg_text_forward"], out["rel_l2_cfg_img_forward"]) <= FULL_DEPTH_TOL_FORWARD
                                   and sc["sequential_forward_flow"] <= FULL_DEPTH_TOL_COMBINE and sc["generate_image_stream_batched"] <= EDIT_DEPTH_TOL_BATCHED)
    return out


def cpu_baseline(args, cfg, gpu=None):
    """The reference CPU path timed on this box's host cores, on a bounded sample of the benchmark workload: one MoT decoder
    layer-forward of one 1024^2 sample at 7B shapes (2.15 TFLOP), extrapolated to images/s as 1 / (Euler steps x 2 forwards x
    layers x t_layer).  kind = "reference" (the unmodified reference classes) where /root/reference exists, else "port" (the
    oracle restatement).  One warm-up pass; thread team = physical cores."""
    from oracle import ref_env
    threads = physical_cores()
    torch.set_num_threads(threads)
    llm = cfg["llm"]
    nl = args.cpu_layers
    n_img = (args.resolution // 16) ** 2
    Lq, C = n_img + 2, args.prompt_tokens + 2
    keep = {}
    kind = "port"
    dt = None
    if ref_env.reference_available() and not args.cpu_port:
        try:
            dt = cpu_reference_layer(args, cfg, threads)
            kind = "reference"
        except Exception as e:
            keep["reference_error"] = repr(e)
    parity = None
    if dt is None or torch.cuda.is_available():
        dtp = cpu_port_layer(args, cfg, threads, keep)
        dt = dtp if dt is None else dt
        if torch.cuda.is_available():
            try:
                parity = full_size_parity(cfg, nl, keep)
            except Exception as e:
                parity = {"error": repr(e)}
    steps = args.num_timesteps - 1
    sec_per_image = steps * 2 * llm["num_hidden_layers"] * dt
    fl = _layer_flops(Lq, C, llm["hidden_size"])
    full = None
    if gpu is not None and torch.cuda.is_available() and not args.no_full_depth:
        try:
            full = full_depth_step(args, cfg, gpu["model"], gpu["tok"], gpu["ids"], threads)
            sec_per_image = steps * full["cpu_seconds_per_euler_step"]          # measured: a whole Euler step at full depth (cond + CFG)
        except Exception as e:
            import traceback
            full = {"error": repr(e), "trace": traceback.format_exc()[-1500:]}
    try:
        config0 = cpu_config0()
    except Exception as e:
        config0 = {"error": repr(e)}
    out = dict(value=1.0 / sec_per_image, unit="images/s", cores=threads, threads=threads, logical_cpus=os.cpu_count(), kind=kind, warmup=1,
               cpu_tflops=fl / dt / 1e12, seconds_per_layer_forward=dt, parity_at_full_size=parity, parity_at_full_depth=full, config0=config0,
               value_from=("one MEASURED Euler step at full depth (cond + CFG forward of all layers, B = 1) x Euler steps" if full and "error" not in full
                           else "one measured layer-forward x layers x 2 forwards x Euler steps"),
               sample=f"{'unmodified reference (generate_image, 1 Euler step = 2 forwards' if kind == 'reference' else 'oracle MoT decoder layer (gen mode'}, "
                      f"{Lq} query tokens on a {C}-token context, 7B shapes, {nl} layer(s)), after one warm-up pass: {dt:.2f} s per layer-forward on "
                      f"{threads} threads = {fl / dt / 1e12:.2f} TFLOP/s; extrapolated x{llm['num_hidden_layers']} layers x2 forwards x{steps} Euler "
                      f"steps (glue, prefill and VAE excluded)"
                      + (f"; and ONE MEASURED Euler step at full depth through the oracle with the GPU model's weights (cond + CFG-text forward of "
                         f"{full['layers']} layers, B = 1): {full['cpu_seconds_per_euler_step']:.1f} s -> value = 1 / ({steps} x that)"
                         if full and "error" not in full else ""))
    if "reference_error" in keep:
        out["reference_error"] = keep["reference_error"]
    return out


def cpu_config0(threads=8):
    """BASELINE.json configs[0] -- the reference's own CPU-runnable case: tiny random-init BAGEL (2-layer MoT, 128-d), text -> image on a
    64x64 latent grid (1024^2 image, 4096 latent tokens), 4 timesteps with CFG, through the oracle.  These are ~5 k tiny operators: a
    256-thread team only adds fork/join cost to each of them (round 1 measured 161.8 s that way), so the thread team is 8 like the
    survey's probe (1.41 s for the reference on 8 cores).  One warm-up pass."""
    from oracle import bagel_oracle as O
    from oracle import packers as P
    from oracle.configs import TINY as cfg, NEW_TOKEN_IDS_TINY as ids, StubTokenizer
    from oracle.shapes import bagel_shapes
    from oracle.weights import synth_state_dict
    before = torch.get_num_threads()
    threads = min(threads, os.cpu_count() or threads)
    torch.set_num_threads(threads)
    try:
        W = {k: v.to(torch.bfloat16) for k, v in synth_state_dict(bagel_shapes(cfg), 0).items()}
        H, L = cfg["llm"]["hidden_size"], cfg["llm"]["num_hidden_layers"]
        W["latent_pos_embed.pos_embed"] = O.sincos_2d_table(H, cfg["bagel"]["max_latent_size"]).to(torch.bfloat16)
        tok = StubTokenizer(cfg["llm"]["vocab_size"])

        def run():
            gi, lens, ropes = P.prepare_prompts([0], [0], ["a small red cube"], tok, ids)
            cache = O.forward_cache_update_text(W, cfg, O.OracleCache(L), **gi)
            torch.manual_seed(42)
            li = P.prepare_vae_latent(lens, ropes, [(1024, 1024)], ids, 16, cfg["bagel"]["max_latent_size"], 64)
            ci = P.prepare_vae_latent_cfg([0], [0], [(1024, 1024)], 16)
            cfgd = dict(cache=O.OracleCache(L), position_ids=ci["cfg_packed_position_ids"], query_indexes=ci["cfg_packed_query_indexes"],
                        key_values_lens=ci["cfg_key_values_lens"], key_value_indexes=ci["cfg_packed_key_value_indexes"])
            return O.generate_image(W, cfg, li, cache, cfg_text=cfgd, num_timesteps=4, timestep_shift=3.0, cfg_renorm_type="global",
                                    cfg_interval=[0.0, 1.0], cfg_text_scale=4.0)
        run()
        t0 = time.time()
        lat = run()
        dt = time.time() - t0
    finally:
        torch.set_num_threads(before)
    return {"seconds": dt, "threads": threads, "warmup": 1, "latent_tokens": int(lat[0].shape[0]), "finite": bool(torch.isfinite(lat[0]).all()),
            "what": "oracle, tiny 2-layer MoT (128-d), text -> 64x64 latent grid, 4 timesteps x [cond + CFG-text], prefill included"}


HBM_PEAK_GBS = 8000.0       # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md ("8 TB/s peak (spec); ~6.3 TB/s achievable")


def cpu_decode_baseline(cfg, ctx):
    """The oracle's MoT decoder layer in und mode at Lq = 1 on a ``ctx``-token KV context (7B shapes, one layer, a few steps) on the
    host cores, extrapolated to tokens/s as 1 / (layers * t_layer + t_lm_head)."""
    from oracle import bagel_oracle as O
    # one-row matrix-vector products are memory-bound and thread-sync bound: 256 threads ran them 80x SLOWER than 8 on the GPU
    # box (2.6 s vs 31 ms per layer), so the baseline uses a moderate team and reports that count as `cores`
    threads = min(os.cpu_count(), 32)
    torch.set_num_threads(threads)
    llm = cfg["llm"]
    H, I, nh, nkv, V = llm["hidden_size"], llm["intermediate_size"], llm["num_attention_heads"], llm["num_key_value_heads"], llm["vocab_size"]
    hd = H // nh
    g = torch.Generator().manual_seed(0)
    p = "language_model.model.layers.0."
    W = {}
    for n, shp in (("q_proj", (nh * hd, H)), ("k_proj", (nkv * hd, H)), ("v_proj", (nkv * hd, H)), ("o_proj", (H, nh * hd))):
        W[p + f"self_attn.{n}.weight"] = (torch.randn(shp, generator=g) * shp[1] ** -0.5).to(torch.bfloat16)
        if n != "o_proj":
            W[p + f"self_attn.{n}.bias"] = (torch.randn(shp[0], generator=g) * 0.02).to(torch.bfloat16)
    for n in ("q_norm", "k_norm"):
        W[p + f"self_attn.{n}.weight"] = torch.ones(hd, dtype=torch.bfloat16)
    for n, shp in (("gate_proj", (I, H)), ("up_proj", (I, H)), ("down_proj", (H, I))):
        W[p + f"mlp.{n}.weight"] = (torch.randn(shp, generator=g) * shp[1] ** -0.5).to(torch.bfloat16)
    for n in ("input_layernorm", "post_attention_layernorm"):
        W[p + f"{n}.weight"] = torch.ones(H, dtype=torch.bfloat16)
    head = (torch.randn((V, H), generator=g) * H ** -0.5).to(torch.bfloat16)
    cache = O.OracleCache(1)
    cache.key_cache[0] = torch.randn(ctx, nkv, hd, generator=g).to(torch.bfloat16)
    cache.value_cache[0] = torch.randn(ctx, nkv, hd, generator=g).to(torch.bfloat16)
    x = torch.randn(1, H, generator=g).to(torch.bfloat16)
    cos_sin = O.rope_tables(torch.tensor([ctx]), hd, llm["rope_theta"], torch.bfloat16)
    ql, kl = torch.tensor([1], dtype=torch.int), torch.tensor([ctx], dtype=torch.int)
    q_idx, kv_idx = torch.tensor([ctx]), torch.arange(ctx)
    steps = 4
    O.mot_layer(W, llm, 0, x, ql, cos_sin, q_idx, cache, kl, kv_idx, False, True, "und", None, None)      # warm-up
    t0 = time.time()
    for _ in range(steps):
        O.mot_layer(W, llm, 0, x, ql, cos_sin, q_idx, cache, kl, kv_idx, False, True, "und", None, None)
    t_layer = (time.time() - t0) / steps
    t0 = time.time()
    for _ in range(steps):
        O.linear(x, head)
    t_head = (time.time() - t0) / steps
    sec = llm["num_hidden_layers"] * t_layer + t_head
    return dict(value=1.0 / sec, unit="tokens/s", cores=threads, kind="port",
                sample=f"oracle MoT layer (und mode, Lq = 1 on a {ctx}-token context, 7B shapes) x{steps}: {t_layer * 1e3:.1f} ms/layer, lm_head "
                       f"{t_head * 1e3:.1f} ms; extrapolated x{llm['num_hidden_layers']} layers + lm_head")


# understanding.parity_at_full_depth bounds = 1.5 x the REFERENCE'S OWN accumulation-order noise at full depth (tools/und_full_depth_noise_floor.py: the oracle
# with bf16 linears vs fp32-accumulating linears on the same 26-layer SigLIP + 28-layer prefill + decode step; profiles/r05_und_full_depth_noise_floor.log)
UND_DEPTH_NOISE = {"kv": 1.504e-2, "logits": 1.371e-2, "source": "profiles/r05_und_full_depth_noise_floor.log"}
UND_DEPTH_TOL_KV = 1.5 * UND_DEPTH_NOISE["kv"]
UND_DEPTH_TOL_LOGITS = 1.5 * UND_DEPTH_NOISE["logits"]


def understanding_full_depth(args, cfg, model, ids, image, tok, threads, n_tokens=6):
    """configs[1] at the DEPTH it is measured at (VERDICT r04 missing-2): the 26-layer SigLIP encoder + connector + the 28-layer non-causal prefill of the
    ViT block + the causal text prefill + ``n_tokens`` greedy decode steps through the oracle on this box's host cores WITH THE GPU MODEL'S OWN WEIGHTS, against
    the product on the same inputs: per-layer K / V of the whole context (max rel-L2), the first decode step's logits (rel-L2), and the greedy ids up to the
    first reference near-tie (the rule of tests/test_und_shapes_gpu.py: a differing id must sit within 2^-6 max|logit| of the reference's top-1).
    Reference: bagel.py:362-415 (forward_cache_update_vit), :321-360 (text), :930-1000 (generate_text); siglip_navit.py:389-402.  Checker use of the oracle only."""
    import copy
    from oracle import bagel_oracle as O
    from bagel_amd.modeling.bagel.qwen2_navit import NaiveCache
    L = model.config.llm_config.num_hidden_layers
    torch.set_num_threads(threads)
    t0 = time.time()
    keep = ("language_model.", "vit_model.", "connector.", "vit_pos_embed.")
    W = {k: v.detach().to("cpu") for k, v in model.state_dict().items() if k.startswith(keep)}
    t_copy = time.time() - t0
    ident = lambda t: t  # noqa: E731
    ti, l1, r1 = model.prepare_vit_images([0], [0], [image], ident, ids)
    pi, l2, r2 = model.prepare_prompts(l1, r1, ["p"], tok, ids)
    st = model.prepare_start_tokens(l2, r2, ids)
    # ---- oracle
    t1 = time.time()
    ocache = O.forward_cache_update_vit(W, cfg, O.OracleCache(L), **ti)
    t_vit = time.time() - t1
    ocache = O.forward_cache_update_text(W, cfg, ocache, **pi)
    t_prefill = time.time() - t1
    okv = [(ocache.key_cache[i].clone(), ocache.value_cache[i].clone()) for i in range(L)]
    t1 = time.time()
    otoks, ologits = O.generate_text(W, cfg, ocache, st["packed_key_value_indexes"], st["key_values_lens"], st["packed_start_tokens"],
                                     st["packed_query_position_ids"], n_tokens, return_logits=True)
    t_decode = time.time() - t1
    del W
    # ---- product
    cache = model.forward_cache_update_vit(NaiveCache(L), **ti)
    cache = model.forward_cache_update_text(cache, **pi)
    rel = lambda a, b: float((a.float().cpu().reshape(b.shape) - b.float()).norm() / b.float().norm())  # noqa: E731
    ek = [rel(cache.key_cache[i], okv[i][0]) for i in range(L)]
    ev = [rel(cache.value_cache[i], okv[i][1]) for i in range(L)]
    model.generate_text(past_key_values=copy.deepcopy(cache), max_length=1, do_sample=False, end_token_id=None, **st)
    logits0 = model._last_decode_session.logits.float().cpu()
    e_logits = float((logits0 - ologits[0].float()).norm() / ologits[0].float().norm())
    toks = model.generate_text(past_key_values=copy.deepcopy(cache), max_length=n_tokens, do_sample=False, end_token_id=None, **st).cpu()
    agree, tie = 0, None
    for s_ in range(1, n_tokens):
        if int(toks[s_, 0]) == int(otoks[s_, 0]):
            agree += 1
            continue
        lg = ologits[s_ - 1][0].float()
        gap = float(lg.max() - lg[int(toks[s_, 0])])
        tie = {"step": s_, "logit_gap": gap, "near_tie_bound": float(2 ** -6 * lg.abs().max()), "is_near_tie": bool(gap <= 2 ** -6 * lg.abs().max())}
        break
    lg0 = ologits[0][0].float()
    out = {"what": f"{cfg['vit']['num_hidden_layers']}-layer SigLIP + connector + {L}-layer prefill of a {int(l2[0])}-token context + {n_tokens} greedy decode steps, 7B shapes, "
                   "identical weights and inputs: HIP engines vs oracle",
           "layers": L, "context_tokens": int(l2[0]), "kv_rel_l2_max": max(ek + ev), "k_rel_l2_by_layer": [round(e, 5) for e in ek],
           "v_rel_l2_by_layer": [round(e, 5) for e in ev], "first_step_logits_rel_l2": e_logits,
           "first_step_top1_margin_over_max_logit": float((lg0.max() - lg0.topk(2).values[1]) / lg0.abs().max()),
           "greedy_ids_agree_until_step": agree + 1 if tie is None else tie["step"], "greedy_steps_compared": n_tokens - 1, "first_mismatch": tie,
           "tokens_gpu": [int(x) for x in toks[:, 0]], "tokens_oracle": [int(x) for x in otoks[:, 0]],
           "cpu_seconds": {"weights_copy": t_copy, "vit_prefill": t_vit, "vit_plus_text_prefill": t_prefill, "decode": t_decode}, "threads": threads,
           "tolerance_kv": UND_DEPTH_TOL_KV, "tolerance_logits": UND_DEPTH_TOL_LOGITS, "noise_floor": UND_DEPTH_NOISE}
    out["within_tolerance"] = bool(out["kv_rel_l2_max"] <= UND_DEPTH_TOL_KV and e_logits <= UND_DEPTH_TOL_LOGITS and (tie is None or tie["is_near_tie"]))
    out["gate"] = (f"max per-layer K/V rel-L2 <= {UND_DEPTH_TOL_KV:.3g}, first-step logits rel-L2 <= {UND_DEPTH_TOL_LOGITS:.3g} (1.5 x the reference's own "
                   "accumulation-order noise at this depth), greedy ids equal up to the first reference near-tie")
    return out


def understanding_leg(args, model, cfg, ids, dev, world, fence):
    """BASELINE.json configs[1]: image understanding = SigLIP prefill (980^2 -> 4900 ViT tokens) + text prefill (32 ids) +
    greedy KV-cached decode of N new tokens (eos disabled), batch 1 per GPU (bagel.py:996), replicas across ranks.
    Decode is HBM-bound: algorithmic bytes/token = und-expert weights + lm_head + the KV context."""
    from bagel_amd.modeling.bagel.qwen2_navit import NaiveCache
    L = model.config.llm_config.num_hidden_layers
    llm = cfg["llm"]
    g = torch.Generator().manual_seed(2)
    image = torch.rand(3, args.und_image, args.und_image, generator=g) * 2 - 1
    prompt_ids = torch.randint(0, 151643, (32,), generator=torch.Generator().manual_seed(1)).tolist()
    tok = FixedTokenizer(prompt_ids)
    ident = lambda t: t  # noqa: E731

    UB = args.und_batch

    def prefill(nb=None):
        nb = UB if nb is None else nb
        cache = NaiveCache(L)
        gi, lens, ropes = model.prepare_vit_images([0] * nb, [0] * nb, [image] * nb, ident, ids)
        cache = model.forward_cache_update_vit(cache, **gi)
        torch.cuda.synchronize()
        t_vit = time.perf_counter()
        gi, lens, ropes = model.prepare_prompts(lens, ropes, ["p"] * nb, tok, ids)
        cache = model.forward_cache_update_text(cache, **gi)
        torch.cuda.synchronize()
        return cache, lens, ropes, t_vit

    def decode(cache, lens, ropes, n):
        st = model.prepare_start_tokens(lens, ropes, ids)
        return model.generate_text(past_key_values=cache, max_length=n, do_sample=False, end_token_id=None, **st)

    cache, lens, ropes, _ = prefill()          # warm-up: engines, workspaces, LDS attributes
    decode(cache, lens, ropes, 8)
    # second warm-up prefill: the timed one below runs while the previous request's cache is still alive, i.e. it needs a SECOND set of K/V buffers (0.76 GB); without
    # this call they came from hipMalloc inside the timed region, which took anything from 0 to 75 ms depending on what else held device memory (the parent bench
    # process: 160 ms instead of 85 on two visits of round 5) -- a serving process is past its first two requests
    cache, lens, ropes, _ = prefill()
    fence()
    t0 = time.perf_counter()
    cache, lens, ropes, t_vit = prefill()
    t1 = time.perf_counter()
    fence()
    n = args.und_new_tokens
    t2 = time.perf_counter()
    toks = decode(cache, lens, ropes, n)
    fence()
    t3 = time.perf_counter()
    sess = model._last_decode_session
    dt = t3 - t2
    # option: row-wise INT8 layer weights (the analogue of the reference's quantised load modes; changes results) -- reported
    # beside the bf16 number, never as it
    def quantised_decode(mode, weights):
        try:
            c8, l8, r8, _ = prefill()
            s8 = model.prepare_start_tokens(l8, r8, ids)
            model.generate_text(past_key_values=c8, max_length=8, do_sample=False, end_token_id=None, weight_quant=mode, **s8)
            c8, l8, r8, _ = prefill()
            s8 = model.prepare_start_tokens(l8, r8, ids)
            fence()
            t4 = time.perf_counter()
            model.generate_text(past_key_values=c8, max_length=n, do_sample=False, end_token_id=None, weight_quant=mode, **s8)
            fence()
            dt8 = time.perf_counter() - t4
            return {"value": n / dt8, "unit": "tokens/s", "decode_ms_per_token": dt8 / n * 1e3, "weights": weights,
                    "note": f"weight_quant='{mode}' option (changes results): not the headline metric"}
        except Exception as e:
            return {"error": repr(e)}
    # do_sample=True (bagel.py:980-983): the draw happens on the device inside the captured step (Gumbel-max, bagel_sample_gumbel_bf16), so the sampled decode replays from
    # the hipGraph like the greedy one -- reported beside it (round 4: sampling ran torch.multinomial on the host side of every EAGER step and was never measured)
    sampled = None
    if UB == 1:
        try:
            cs, ls, rs, _ = prefill()
            ss = model.prepare_start_tokens(ls, rs, ids)
            torch.manual_seed(0)
            fence()
            t4 = time.perf_counter()
            ts_ = model.generate_text(past_key_values=cs, max_length=n, do_sample=True, temperature=0.7, end_token_id=None, **ss)
            fence()
            dts = time.perf_counter() - t4
            sampled = {"value": n / dts, "unit": "tokens/s", "decode_ms_per_token": dts / n * 1e3, "temperature": 0.7,
                       "hip_graph": model._last_decode_session.graph is not None, "distinct_tokens": int(torch.unique(ts_).numel()),
                       "note": "generate_text(do_sample=True): device-side Gumbel-max sampler inside the hipGraph; same categorical distribution as torch.multinomial, its own RNG stream"}
            del cs
        except Exception as e:
            sampled = {"error": repr(e)}
    w8 = w4 = wn = None
    if UB == 1 and not args.no_int8:
        # the reference's OWN 4-bit load mode (app.py:114-125: bitsandbytes NF4, blocks of 64, fp32 absmax, bf16 compute)
        wn = quantised_decode("nf4", "bitsandbytes NF4 (code book of 16, blocks of 64 with fp32 absmax, no double quantisation), W4A16, lm_head bf16")
        w8 = quantised_decode("int8_rowwise", "row-wise absmax INT8 (W8A16, de-quantised on the VALU), lm_head bf16")
        # the 4-bit counterpart of the reference's NF4 load mode: OCP-MX FP4 weights x FP8 activations on the block-scaled MFMA
        w4 = quantised_decode("mxfp4", "OCP-MX FP4 E2M1 blocks of 32 with E8M0 scales (W4A8 on v_mfma_scale_f32_16x16x128_f8f6f4), lm_head bf16")
    # SURVEY 8f.4b beside the batch-1 number: 16 requests decoded together (one weight pass serves the batch; the reference decodes
    # batch 1 only, bagel.py:996) -- 160 new tokens each on their own 4936-token contexts
    bd = None
    if UB == 1 and not args.no_batched_decode:
        try:
            nb, nn = 16, 160         # (step 0 runs eagerly and the hipGraph capture follows it: amortised over the run)
            cb, lb, rb, _ = prefill(nb)
            sb = model.prepare_start_tokens(lb, rb, ids)
            # warm-up with the SAME length: the call re-allocates the merged caches on the way out (16 x (4936 + 160) rows x 28 layers), and a warm-up
            # of another length left the timed call ~80 ms of first-time hipMalloc (0.5 ms per step of 160) -- a serving process is past that
            model.generate_text(past_key_values=cb, max_length=nn, do_sample=False, end_token_id=None, **sb)
            cb, lb, rb, _ = prefill(nb)
            sb = model.prepare_start_tokens(lb, rb, ids)
            fence()
            t4 = time.perf_counter()
            tb = model.generate_text(past_key_values=cb, max_length=nn, do_sample=False, end_token_id=None, **sb)
            fence()
            dtb = time.perf_counter() - t4
            bd = {"value": nb * nn / dtb, "unit": "tokens/s", "batch": nb, "new_tokens": nn, "decode_ms_per_step": dtb / nn * 1e3,
                  "context_tokens": int(lb[0]), "outputs_ok": bool(tb.shape == (nn, nb)),
                  "note": "batched multi-request decode (SURVEY 8f.4b): beside the batch-1 headline, never as it"}
            del cb
        except Exception as e:
            bd = {"error": repr(e)}
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    H, I, nkv, V = llm["hidden_size"], llm["intermediate_size"], llm["num_key_value_heads"], llm["vocab_size"]
    hd = H // llm["num_attention_heads"]
    cpu = None
    if not args.no_cpu_baseline and int(os.environ.get("RANK", 0)) == 0:
        try:
            cpu = cpu_decode_baseline(cfg, lens[0])
        except Exception as e:
            cpu = {"error": repr(e)}
    depth = None
    if UB == 1 and not args.no_cpu_baseline and not args.no_full_depth and int(os.environ.get("RANK", 0)) == 0:
        try:
            del cache
            depth = understanding_full_depth(args, cfg, model, ids, image, tok, physical_cores())
        except Exception as e:
            depth = {"error": repr(e)}
    w_bytes = 2.0 * (L * (2 * H * H + 2 * H * nkv * hd + 3 * H * I) + V * H)
    ctx = lens[0]
    kv_bytes = 2.0 * nkv * hd * 2 * L * (ctx + n / 2.0)          # average context over the decoded span
    bpt = w_bytes + UB * kv_bytes            # one weight pass serves the whole batch; every request reads its own KV
    tps = UB * n / dt
    return {"metric": "understanding tokens/sec", "value": world * tps, "unit": "tokens/s", "per_gpu_tokens_per_s": tps,
            "new_tokens": int(toks.shape[0]), "batch_per_gpu": UB, "context_tokens": int(ctx),
            "prefill_ms": {"vit_encoder_plus_llm_prefill": (t_vit - t0) * 1e3, "text_prefill": (t1 - t_vit) * 1e3},
            "decode_ms_per_step": dt / n * 1e3, "decode_ms_per_token": dt / n / UB * 1e3, "hip_graph": sess.graph is not None, "hip_graph_error": sess.graph_error,
            "kv_cache": f"paged, {sess.paged.PAGE}-token pages, {sess.paged.num_pages} pages/layer", "cpu_baseline": cpu,
            "int8_rowwise_weights": w8, "mxfp4_weights": w4, "nf4_weights": wn, "batched_decode": bd, "sampled_decode": sampled, "parity_at_full_depth": depth,
            "roofline": {"bound": "hbm", "achieved": bpt * (tps / UB) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": bpt * (tps / UB) / 1e9 / HBM_PEAK_GBS, "traffic": pmc_decode_traffic() if (UB == 1 and args.und_image == 980) else None,
                         "kernel": "gemv_kernel (decode step)",
                         "algorithmic_bytes_per_step": bpt},
            "workload": f"BAGEL-7B-MoT image understanding: {args.und_image}x{args.und_image} image -> {(args.und_image // 14) ** 2} ViT tokens (+2 markers) + 32+2 "
                        f"prompt tokens prefill, greedy decode of {n} tokens, bf16, batch {UB}/GPU"}


PMC_SUMMARY = os.path.join(ROOT, "profiles", "r05_pmc_summary.json")


def _source_digest(names):
    import hashlib
    h = hashlib.sha1()
    for n in names:
        with open(os.path.join(ROOT, "bagel_amd", "csrc", n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _pmc_entry(section, key, sources):
    """One figure of the committed PMC summary -- or None when the summary was collected on OTHER kernel sources than the ones this
    run executes (the summary records the sha1 of the .hip files its kernels were built from; tools/pmc_summary.py writes it)."""
    try:
        with open(PMC_SUMMARY) as f:
            d = json.load(f)
        if d.get("source_digest", {}).get(section) != _source_digest(sources):
            return None
        return d
    except Exception:
        return None


def pmc_traffic(kernel):
    """HBM-side bytes per launch of ``kernel`` from the committed PMC pass (FETCH_SIZE x2 per the gfx950 correction of
    MI355X_MICROARCH.md + WRITE_SIZE; separate rocprofv3 --pmc runs, tools/gpu_pmc.sh -> profiles/r02_pmc_summary.json).  A profiler
    cannot run inside the timed bench, so this is the per-launch figure of the same kernel on the same shapes; null if the summary is
    absent or was collected on a different gemm.hip."""
    d = _pmc_entry("gemm", kernel, ["gemm.hip", "common.h"])
    try:
        return d["kernels"][kernel]["traffic_bytes_per_launch_corrected"]
    except Exception:
        return None


PMC_SOURCE = "committed PMC pass (separate rocprofv3 --pmc runs of the same kernels on the same shapes), digest-checked against the kernel sources"


def attention_object(arecords, args, R):
    """Second kernel of the denoise path: the planned persistent attention kernel (csrc/attention2.hip), timed live like the GEMM (HIP
    events around every launch of the timed region; the big launches = the stream-batched denoise forwards) + the PMC figures of the
    committed summary while attention2.hip still hashes to the digest they were collected on."""
    big = [r for r in arecords if r[0] > 1e11]
    if not big:
        return None
    fl = sum(r[0] for r in big)
    ms = sum(r[1].elapsed_time(r[2]) for r in big)
    ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    out = {"kernel": "attn2_kernel<128> (+ attn2_combine_kernel<128>)", "bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
           "frac": ach / PEAK_BF16_TFLOPS, "launches": len(big), "avg_launch_ms": ms / len(big), "samples_per_launch": big[0][3]}
    d = _pmc_entry("attention", None, ["attention2.hip", "common.h"]) if (args.workload == "t2i" and args.batch == 4 and R == 1024) else None
    k = (d or {}).get("kernels", {}).get("attn2_kernel<128>")
    if k:
        out["traffic"] = k.get("traffic_bytes_per_launch_corrected")
        out["pmc"] = {x: k.get(x) for x in ("mfma_busy_frac", "l2_hit_rate", "wave_cycles_split", "algorithmic_bytes_per_launch")}
        out["pmc"]["source"] = PMC_SOURCE
    else:
        out["traffic"] = None
    return out


def pmc_decode_traffic():
    """HBM-side bytes of one decode step (all its kernels) from the committed PMC passes; see pmc_traffic."""
    d = _pmc_entry("decode", None, ["decode.hip", "skinny.hip", "common.h"])
    try:
        return d["decode_step"]["traffic_bytes_per_step_corrected"]
    except Exception:
        return None


def understanding_subprocess(args, local):
    """Run the configs[1] leg in a child process on the same GPU (a replica per rank): a fault or hang there can never
    take the text->image number down with it."""
    import subprocess
    env = dict(os.environ)
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK=str(local), LOCAL_WORLD_SIZE="1")
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--only-understanding"] + (["--no-cpu-baseline"] if args.no_cpu_baseline or int(os.environ.get("WORLD_SIZE", 1)) != 1 else []) + [
           "--und-new-tokens", str(args.und_new_tokens), "--und-image", str(args.und_image), "--und-batch", str(args.und_batch)] + (
           ["--no-int8"] if args.no_int8 else []) + (["--no-batched-decode"] if args.no_batched_decode else []) + (["--no-full-depth"] if args.no_full_depth else [])
    if args.layers is not None:
        cmd += ["--layers", str(args.layers)]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1800)
    except subprocess.TimeoutExpired:
        return {"error": "understanding leg timed out after 1800 s"}
    for line in reversed(r.stdout.strip().splitlines()):
        if line.startswith("{"):
            try:
                return json.loads(line).get("understanding")
            except ValueError:
                break
    return {"error": f"understanding leg exited {r.returncode}", "stderr_tail": r.stderr[-1500:]}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if args.cpu_baseline_only:
        from bagel_amd.factory import BAGEL_7B_MOT
        print(json.dumps({"cpu_baseline": cpu_baseline(args, BAGEL_7B_MOT)}), flush=True)
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)                     # does not return
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.launch_check:
        return launch_check(args, rank, world, local)
    import torch.distributed as dist
    cuda = not args.standins
    if cuda:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    else:
        dev = torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if cuda:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from bagel_amd import ops
    from bagel_amd.factory import BAGEL_7B_MOT, NEW_TOKEN_IDS_QWEN25, build_bagel, init_random_
    from bagel_amd.inferencer import InterleaveInferencer
    from bagel_amd.modeling.bagel.qwen2_navit import NaiveCache
    from bagel_amd.parallel import broadcast_cache

    cfg = BAGEL_7B_MOT
    if args.standins:
        # TEST ONLY: the launch wrappers become the torch stand-ins of tests/mock_ops.py and the model a 2-layer toy, so that THIS
        # function's sharding, broadcast, fence, timing and JSON assembly run on two gloo ranks without a GPU.  Nothing measured here
        # is a benchmark number (the line says "valid": false).
        from tests import mock_ops
        from oracle.configs import TINY
        mock_ops.install_permanently()
        cfg = TINY
        args.no_taylorseer = args.no_understanding = args.no_fp8 = args.no_edit = args.no_cpu_baseline = True
    model, vae = build_bagel(cfg, device=dev, num_layers=args.layers, with_vae=not args.only_understanding)
    init_random_(model, seed=0)
    if vae is not None:
        init_random_(vae, seed=0)
    model.llm2vae.weight.data.normal_(0, cfg["llm"]["hidden_size"] ** -0.5, generator=torch.Generator(device=dev).manual_seed(1))
    store_info = None
    if args.weight_store:
        args.no_fp8 = args.no_train_forward = args.no_understanding = True
        before = torch.cuda.memory_allocated(dev) if cuda else 0
        resident = model.quantize_language_model(args.weight_store)
        store_info = {"kind": args.weight_store, "resident_gb": resident / 1e9,
                      "hbm_freed_gb": (before - torch.cuda.memory_allocated(dev)) / 1e9 if cuda else None}
    # who is really here: an all-reduce of ones over the job's process group (RCCL on the GPUs), and the collective library's version
    ranks_seen, rccl_version, backend = 1, None, None
    if world > 1:
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
        backend = dist.get_backend()
    if cuda:
        try:
            rccl_version = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception as e:     # reported, never required
            rccl_version = f"unavailable ({type(e).__name__})"
    L = model.config.llm_config.num_hidden_layers
    B, R, T = args.batch, args.resolution, args.num_timesteps
    g = torch.Generator().manual_seed(1)
    prompt_ids = torch.randint(0, min(151643, cfg["llm"]["vocab_size"] - 8), (args.prompt_tokens,), generator=g).tolist()
    tok = FixedTokenizer(prompt_ids)
    ids = NEW_TOKEN_IDS_QWEN25
    if args.standins:
        ids = dict(bos_token_id=1, eos_token_id=2, start_of_image=3, end_of_image=4)     # inside the toy's 512-entry vocabulary
    inf = InterleaveInferencer(model, vae, tok, None, None, ids)
    noise_gen = torch.Generator().manual_seed(42)
    pdim = model.patch_latent_dim
    n_img = (R // model.latent_downsample) ** 2
    # global noise stream of the whole job; rank r takes rows [r*B, (r+1)*B)
    all_noise = torch.randn(world * B * n_img, pdim, generator=noise_gen)
    my_noise = all_noise[rank * B * n_img:(rank + 1) * B * n_img].to(dev)

    bcast = []          # (start, end, bytes) of every conditioning-KV broadcast

    def one_step(taylorseer=False):
        # conditioning context: computed once (rank 0) and broadcast; every sample shares the prompt (gen_images_mp.py:43)
        gi, newlens, newrope = model.prepare_prompts([0] * B, [0] * B, ["p"] * B, tok, ids)
        if rank == 0:
            cache = model.forward_cache_update_text(NaiveCache(L), **gi)
        else:
            cache = NaiveCache(L)
        if world > 1:
            # the one exchange of the data-parallel path: events on the launch stream around the collective --
            # on rank 0 the prefill is ordered in front of the first event, so the pair brackets the broadcast itself
            st = {}
            if cuda:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                cache = broadcast_cache(cache, src=0, stats=st)
                e1.record()
                bcast.append((e0, e1, st.get("bytes", 0)))
            else:
                t_b = time.perf_counter()
                cache = broadcast_cache(cache, src=0, stats=st)
                bcast.append((t_b, time.perf_counter(), st.get("bytes", 0)))
        li = model.prepare_vae_latent(newlens, newrope, [(R, R)] * B, ids)
        li["packed_init_noises"] = my_noise
        ci = model.prepare_vae_latent_cfg([0] * B, [0] * B, [(R, R)] * B)
        latents = model.generate_image(
            past_key_values=cache, num_timesteps=T, cfg_text_scale=4.0, cfg_interval=[0, 1.0], cfg_renorm_min=0.0,
            cfg_renorm_type="global", timestep_shift=3.0, cfg_text_past_key_values=NaiveCache(L),
            cfg_text_packed_position_ids=ci["cfg_packed_position_ids"], cfg_text_packed_query_indexes=ci["cfg_packed_query_indexes"],
            cfg_text_key_values_lens=ci["cfg_key_values_lens"], cfg_text_packed_key_value_indexes=ci["cfg_packed_key_value_indexes"],
            enable_taylorseer=taylorseer, **li)
        imgs = []
        if not args.no_vae:
            for lat in latents:
                # the VAE OUTSIDE any autocast region, in fp32: how the reference's batch driver decodes (eval/gen/gen_images_mp.py:93)
                img = vae.decode(inf.latent_to_chw(lat, (R, R)), **({} if args.standins else {"precision": "fp32"}))
                imgs.append(inf.image_to_u8(img))
        return latents, imgs

    # ---- BASELINE configs[4]: image edit = VAE-encode + ViT-encode the source image into the context, prompt on top, then
    #      the 3-forward sampler (cond + cfg-text + cfg-img, text_channel renorm; app.py:224-228), one request per GPU
    edit_state = {}

    def edit_step(taylorseer=False, timesteps=None):
        import copy
        if not edit_state:
            gsrc = torch.Generator().manual_seed(3)
            edit_state["src_vae"] = (torch.rand(3, R, R, generator=gsrc) * 2 - 1).to(dev)        # vae_transform(image) stand-in
            edit_state["src_vit"] = (torch.rand(3, 980, 980, generator=gsrc) * 2 - 1).to(dev)    # vit_transform(image) stand-in
            edit_state["enc_noise"] = torch.randn(1, 16, R // 8, R // 8, generator=torch.Generator().manual_seed(43))
            edit_state["noise"] = all_noise[rank * n_img:(rank + 1) * n_img].to(dev)
        ident = lambda t: t  # noqa: E731

        class _FixedNoiseVae:      # the reference draws randn_like inside encode; feed a seeded CPU draw
            def encode(self, x):       # the edit request is the app's / inferencer's path: its VAE runs inside torch.autocast(bf16) (inferencer.py:233)
                return vae.encode(x, sample_noise=edit_state["enc_noise"], **({} if args.standins else {"precision": "bf16"}))

        ctx = dict(kv_lens=[0], ropes=[0], past_key_values=NaiveCache(L))
        vi, l1, r1 = model.prepare_vae_images(ctx["kv_lens"], ctx["ropes"], [edit_state["src_vae"]], ident, ids)
        cache = model.forward_cache_update_vae(_FixedNoiseVae(), ctx["past_key_values"], **vi)
        ti, l2, r2 = model.prepare_vit_images(l1, r1, [edit_state["src_vit"]], ident, ids)
        cache = model.forward_cache_update_vit(cache, **ti)
        cfg_text_cache = copy.deepcopy(cache)
        pi, l3, r3 = model.prepare_prompts(l2, r2, ["p"], tok, ids)
        cache = model.forward_cache_update_text(cache, **pi)
        pi2, l4, r4 = model.prepare_prompts([0], [0], ["p"], tok, ids)
        cimg_cache = model.forward_cache_update_text(NaiveCache(L), **pi2)
        li = model.prepare_vae_latent(l3, r3, [(R, R)], ids)
        li["packed_init_noises"] = edit_state["noise"]
        ct = model.prepare_vae_latent_cfg(l2, r2, [(R, R)])
        cim = model.prepare_vae_latent_cfg(l4, r4, [(R, R)])
        kw = {}
        for tag, c, d in (("cfg_text", cfg_text_cache, ct), ("cfg_img", cimg_cache, cim)):
            kw.update({f"{tag}_past_key_values": c, f"{tag}_packed_position_ids": d["cfg_packed_position_ids"],
                       f"{tag}_packed_query_indexes": d["cfg_packed_query_indexes"], f"{tag}_key_values_lens": d["cfg_key_values_lens"],
                       f"{tag}_packed_key_value_indexes": d["cfg_packed_key_value_indexes"]})
        latents = model.generate_image(past_key_values=cache, num_timesteps=timesteps or T, cfg_text_scale=4.0, cfg_img_scale=2.0,
                                       cfg_interval=[0.0, 1.0], cfg_renorm_min=0.0, cfg_renorm_type="text_channel",
                                       timestep_shift=3.0, enable_taylorseer=taylorseer, **kw, **li)
        imgs = []
        if not args.no_vae:
            for lat in latents:
                imgs.append(inf.image_to_u8(vae.decode(inf.latent_to_chw(lat, (R, R)), **({} if args.standins else {"precision": "bf16"}))))
        edit_state["context_tokens"] = (l3[0], l2[0], l4[0])     # cond, cfg-text, cfg-img contexts
        return latents, imgs

    t2i_step = one_step
    if args.workload == "edit":
        B = 1
        one_step = edit_step

    def fence():
        if cuda:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            if cuda:
                torch.cuda.synchronize()

    if args.only_understanding:
        # child mode (see understanding_subprocess): this process measures configs[1] only
        try:
            und = understanding_leg(args, model, cfg, ids, dev, world, fence)
        except Exception as e:
            import traceback
            und = {"error": repr(e), "trace": traceback.format_exc()[-1500:]}
        if rank == 0:
            print(json.dumps({"metric": "understanding tokens/sec (debug: text->image leg skipped)", "valid": False,
                              "understanding": und}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    for _ in range(args.warmup):
        one_step()
    records, orig_gemm = [], ops.gemm
    arecords, orig_attn = [], ops.attn_planned
    if cuda:
        records, orig_gemm, timed_gemm = gemm_profile_hook()
        ops.gemm = timed_gemm
        arecords, orig_attn, timed_attn = attn_profile_hook()
        ops.attn_planned = timed_attn
    del bcast[:]
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        latents, imgs = one_step()
    fence()
    dt = time.perf_counter() - t0
    ops.gemm = orig_gemm
    ops.attn_planned = orig_attn
    bcast_timed = list(bcast)
    dt_rank = dt
    per_rank_ms = [dt / args.steps * 1e3]
    if world > 1:
        # every rank's own clock around the same barrier-fenced region: MAX is the job's time, the spread shows a straggler
        tt = torch.zeros(world, dtype=torch.float64, device=dev)
        tt[rank] = dt_rank
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        per_rank_ms = [float(x) / args.steps * 1e3 for x in tt.tolist()]
        dt = float(tt.max().item())
    finite = all(torch.isfinite(x).all().item() for x in latents)
    ts = None
    if not args.no_taylorseer:
        # the reference's own accelerator option (generate_image(enable_taylorseer=True), bagel.py:678-689): same workload,
        # 19 instead of 49 full backbone forwards per stream.  It CHANGES the samples, so it is reported beside the headline
        # number, never as it.
        fence()
        t1 = time.perf_counter()
        lat_ts, _ = one_step(taylorseer=True)
        fence()
        dt_ts = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([dt_ts], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_ts = float(tt.item())
        st = model._last_taylor_states[0]
        ts = {"value": world * B / dt_ts, "unit": "images/s", "ms_per_step": dt_ts * 1e3, "full_forwards_per_stream": st.full_steps,
              "extrapolated_forwards_per_stream": st.taylor_steps, "outputs_finite": all(torch.isfinite(x).all().item() for x in lat_ts),
              "note": "enable_taylorseer=True (reference option, changes the samples): not the headline metric"}
    fp8 = None
    if args.workload == "t2i" and not args.no_fp8 and not args.only_understanding:
        # option model.gen_weight_quant = "fp8": the gen expert's projections on the OCP-e4m3 MFMA with row-wise scales (SURVEY.md 8f.4;
        # the MI355X counterpart of the reference's quantised load modes).  It CHANGES the samples (a few percent, tests/test_fp8_gpu.py):
        # reported beside the headline number, never as it.
        try:
            model.gen_weight_quant = "fp8"
            one_step()                                  # warm-up: quantises the 28 x 4 gen-expert matrices once
            fence()
            t1 = time.perf_counter()
            lat_8, _ = one_step()
            fence()
            dt_8 = time.perf_counter() - t1
            if world > 1:
                tt = torch.tensor([dt_8], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt_8 = float(tt.item())
            dev_l2 = max(float(((a.float() - b.float()).norm() / b.float().norm()).item()) for a, b in zip(lat_8, latents))
            fp8 = {"value": world * B / dt_8, "unit": "images/s", "ms_per_step": dt_8 * 1e3, "outputs_finite": all(torch.isfinite(x).all().item() for x in lat_8),
                   "latents_rel_l2_vs_bf16_run": dev_l2, "weights": "gen expert q/k/v/o/gate/up/down in OCP e4m3, row-wise absmax scales; activations "
                   "quantised per row on the fly; und expert, attention, norms, residual stream bf16",
                   "note": "gen_weight_quant='fp8' option (changes results): not the headline metric"}
        except Exception as e:
            import traceback
            fp8 = {"error": repr(e), "trace": traceback.format_exc()[-1200:]}
        finally:
            model.gen_weight_quant = None
    edit = None
    if args.workload == "t2i" and not args.no_edit and vae is not None:
        # BASELINE configs[4] beside the headline: ONE image-edit request per GPU (VAE-encode + SigLIP + prompt -> ~9 k-token
        # context, 49 Euler steps x 3 forwards, text_channel renorm, VAE decode), after a 2-step warm-up of its shapes
        try:
            edit_step(timesteps=3)
            fence()
            t1 = time.perf_counter()
            lat_e, _ = edit_step()
            fence()
            dt_e = time.perf_counter() - t1
            if world > 1:
                tt = torch.tensor([dt_e], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt_e = float(tt.item())
            c_cond, c_text, c_img = edit_state["context_tokens"]
            Lq = n_img + 2
            pf = (T - 1) * sum(_layer_flops(Lq, c, cfg["llm"]["hidden_size"]) for c in (c_cond, c_text, c_img)) * L
            edit = {"metric": "images/sec (image edit 1024^2, 50-step, 3-forward CFG), 7B-MoT", "value": world / dt_e, "unit": "images/s",
                    "seconds_per_image": dt_e, "requests_per_gpu": 1, "context_tokens": {"cond": c_cond, "cfg_text": c_text, "cfg_img": c_img},
                    "denoise_pflop_per_image": pf / 1e15,
                    "whole_path_roofline": {"bound": "mfma", "achieved": pf / dt_e / 1e12, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                            "frac": pf / dt_e / 1e12 / PEAK_BF16_TFLOPS,
                                            "note": "denoise FLOPs only (linear + attention of the 147 forwards) over the WHOLE request time incl. VAE encode, ViT, prefill and VAE decode"},
                    "outputs_finite": all(torch.isfinite(x).all().item() for x in lat_e),
                    "vae_precision": "bf16 convolutions, fp32 GroupNorm (the VAE inside the inferencer's autocast region, inferencer.py:233), encode and decode",
                    "workload": f"BASELINE configs[4]: VAE-encode + SigLIP(980^2) + {args.prompt_tokens}+2 prompt tokens, {T} timesteps x [cond + CFG-text 4.0 + "
                                f"CFG-img 2.0], text_channel renorm, VAE decode included, 1 request/GPU"}
        except Exception as e:
            import traceback
            edit = {"error": repr(e), "trace": traceback.format_exc()[-1200:]}
    trainf = None
    if args.workload == "t2i" and not args.no_train_forward and (args.layers is None or args.standins):
        # SURVEY 8f.2 beside the headline: Bagel.forward (training forward, per-token CE / MSE losses, no backward) on a packed 7B batch of
        # 2 x [prompt | 980^2 ViT image | answer + CE] + 2 x [prompt | noised 1024^2 latent + MSE] (tools/train_forward_probe.py)
        try:
            if args.standins:        # the tiny model on the CPU stand-ins: a hand-packed two-sample batch (test-only mode, host logic of the legs)
                from tests.util_models import pack_training_batch
                tb, tn, _, _ = pack_training_batch(cfg, [[("text", 3, True), ("vit", 28, 42), ("text", 4, True)], [("text", 2, False), ("vae", 32, 48, True)]], 7)
            else:
                from tools.train_forward_probe import build_batch
                tb = build_batch(model, ids)
                tn = torch.randn(len(tb["packed_vae_token_indexes"]), 64, generator=torch.Generator().manual_seed(1)).to(dev)
            o_ = model(noise=tn, **tb)
            fence()
            t1 = time.perf_counter()
            for _ in range(3):
                o_ = model(noise=tn, **tb)
            fence()
            dtt = (time.perf_counter() - t1) / 3
            ntok = tb["sequence_length"]
            trainf = {"value": world * ntok / dtt, "unit": "tokens/s", "tokens_per_forward": ntok, "ms_per_forward": dtt * 1e3,
                      "linear_tflops": 13.0506e-3 * ntok / dtt, "outputs_finite": bool(torch.isfinite(o_["ce"]).all() and torch.isfinite(o_["mse"]).all()),
                      "note": "training FORWARD only (losses, no tape): beside the headline, never as it"}
            if not args.no_train_step:
                # the whole step of train/pretrain_unified_navit.py:683-735 minus the optimizer: forward with a tape + loss.backward() through
                # the hand-written reverse (bagel_amd/modeling/bagel/train_step.py), every language-model / connector / head parameter trainable,
                # SigLIP tower included; gradients of 14.6 G parameters are produced and dropped
                frozen = ("vit_pos_embed.", "latent_pos_embed.")
                try:
                    n_train = 0
                    for n_, p_ in model.named_parameters():
                        p_.requires_grad_(not n_.startswith(frozen))
                        n_train += p_.numel() if p_.requires_grad else 0

                    def one_train_step():
                        for p_ in model.parameters():
                            p_.grad = None
                        fence(); a0 = time.perf_counter()
                        with torch.enable_grad():
                            oo = model(noise=tn, **tb)
                            loss = oo["ce"].mean() + oo["mse"].mean()
                        fence(); a1 = time.perf_counter()
                        loss.backward()
                        fence(); a2 = time.perf_counter()
                        # an optimizer step's effect on the engines: every trainable parameter rewritten IN PLACE (plain SGD with a
                        # vanishing rate on the bf16 parameters -- the optimizer itself is torch's and out of scope), so the NEXT forward
                        # pays the refresh of the packed copies and transposed images (MoTEngine.refresh) like a real loop does
                        with torch.no_grad():
                            for p_ in model.parameters():
                                if p_.grad is not None:
                                    p_.add_(p_.grad, alpha=-1e-9)
                        fence(); a3 = time.perf_counter()
                        return float(loss.detach()), a1 - a0, a2 - a1, a3 - a2
                    one_train_step()
                    one_train_step()
                    rs = [one_train_step() for _ in range(2)]
                    tf_, tb_, to_ = sum(r[1] for r in rs) / 2, sum(r[2] for r in rs) / 2, sum(r[3] for r in rs) / 2
                    gn = sum(float(p_.grad.float().norm()) ** 2 for p_ in model.parameters() if p_.grad is not None) ** 0.5
                    lin = 13.0506e-3 * ntok                                   # TFLOP of the decoder's linears in one forward
                    trainf["training_step"] = {
                        "value": world * ntok / (tf_ + tb_), "unit": "tokens/s", "ms_forward_with_tape": tf_ * 1e3, "ms_backward": tb_ * 1e3,
                        "ms_inplace_parameter_update": to_ * 1e3,
                        "engine_rebuilt_per_step": False,
                        "trainable_params": n_train, "loss": rs[-1][0], "grad_norm": gn, "finite": bool(gn == gn and gn < float("inf")),
                        "linear_tflops_whole_step": 3 * lin / (tf_ + tb_), "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30 if cuda else None,
                        "note": "forward with tape + backward in the steady state of a REAL loop: every step is followed by an in-place rewrite of all "
                                "trainable parameters, so ms_forward_with_tape includes the in-place refresh of the packed weights and transposed "
                                "images (the optimizer arithmetic itself is torch's, out of scope: its stand-in's time is reported, not counted); 3 x "
                                "the forward's linear FLOPs over forward + backward (gate/up recompute and attention reverse are not useful work)"}
                except Exception as e:
                    import traceback
                    trainf["training_step"] = {"error": repr(e), "trace": traceback.format_exc()[-1200:]}
                finally:
                    for p_ in model.parameters():
                        p_.requires_grad_(False)
                        p_.grad = None
                    if cuda:
                        torch.cuda.empty_cache()
            del tb, tn, o_
        except Exception as e:
            import traceback
            trainf = {"error": repr(e), "trace": traceback.format_exc()[-1200:]}
    und = None
    if not args.no_understanding:
        und = understanding_subprocess(args, local)
        if world > 1:   # replicas: aggregate tokens/s = sum over ranks
            v = und.get("per_gpu_tokens_per_s", 0.0) if isinstance(und, dict) else 0.0
            tt = torch.tensor([v], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.SUM)
            if isinstance(und, dict) and "value" in und:
                und["value"] = float(tt.item())
                und["replicas"] = world

    if rank == 0:
        # the dominant kernel = the GEMM variant that carries the most FLOPs in the timed region
        names = {0: "gemm_tn_kernel<128,128,2,2>", 1: "gemm_tn_kernel<256,256,2,4>", 2: "gemm_tn_kernel<256,128,2,2>",
                 3: "gemm_pp_kernel<0>", 4: "gemm_pq_kernel<*>", 5: "gemm_pq_kernel<*>"}
        by_v = {}
        for r in records:
            by_v[r[3]] = by_v.get(r[3], 0.0) + r[0]
        dom = max(by_v, key=by_v.get) if by_v else 3
        records = [r for r in records if r[3] == dom]
        flops = sum(r[0] for r in records)
        ms = sum(r[1].elapsed_time(r[2]) for r in records)
        ach = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        images = world * B * args.steps
        out = {
            "metric": "images/sec (1024^2, 50-step) + understanding tokens/sec, 7B-MoT, 1/2/4/8 GPU",
            "value": images / dt, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic (random-init BAGEL-7B-MoT weights, random prompt ids, seed-42 CPU noise)",
            "config": {"workload": (f"BAGEL-7B-MoT image edit {R}x{R} (BASELINE configs[4]): VAE-encode + SigLIP(980^2) + {args.prompt_tokens}+2 prompt "
                                    f"tokens -> {edit_state.get('context_tokens', (0,))[0]}-token context, {T} timesteps ({T - 1} Euler steps x [cond + CFG-text 4.0 "
                                    f"+ CFG-img 2.0]), text_channel renorm, 1 request/GPU, VAE decode included") if args.workload == "edit" else
                                   f"BAGEL-7B-MoT text->image {R}x{R}, {T} timesteps ({T - 1} Euler steps x [cond + CFG-text 4.0]), "
                                   f"global renorm, timestep_shift 3, prompt {args.prompt_tokens}+2 tokens, {B} samples/GPU, VAE decode included",
                       "global_batch": world * B, "query_tokens_per_sample": n_img + 2, "parallelism": f"dp{world}",
                       "vae_precision": ("bf16 convolutions, fp32 GroupNorm: the VAE inside the inferencer's autocast region (inferencer.py:233)" if args.workload == "edit"
                                         else "fp32: the VAE outside any autocast region, as eval/gen/gen_images_mp.py:93 decodes"),
                       # execution options in force (promoted in round 2 after measurement)
                       "options": {"cfg_batched": bool(getattr(model, "cfg_batched", False)),
                                   "und_side_path": bool(getattr(model, "cfg_batched", False) and getattr(model, "und_side_path", False))}},
            "roofline": {"bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS,
                         "traffic": pmc_traffic(names.get(dom, str(dom))) if (args.workload == "t2i" and args.batch == 4 and R == 1024) else None,
                         "traffic_source": PMC_SOURCE + f" ({os.path.relpath(PMC_SUMMARY, ROOT)}); null when the sources changed since", "kernel": names.get(dom, str(dom)), "launches": len(records),
                         "avg_launch_ms": ms / max(len(records), 1), "gemm_time_share": ms * 1e-3 / dt},
            "attention": attention_object(arecords, args, R),
            "outputs_finite": bool(finite),
            # the job as the collective library saw it (an all-reduce of ones at start-up) and the conditioning-KV broadcast of the timed
            # steps as rank 0 timed it (HIP events on the launch stream; null at N = 1: there is no exchange)
            "ranks_seen": ranks_seen, "collective_backend": backend, "rccl_version": rccl_version,
            "per_rank_ms_per_step": {"min": min(per_rank_ms), "max": max(per_rank_ms), "ranks": per_rank_ms},
            "broadcast": None if not bcast_timed else {
                "calls": len(bcast_timed), "bytes_per_call": bcast_timed[0][2],
                "ms_per_call": (sum(a.elapsed_time(b) for a, b, _ in bcast_timed) if cuda else sum((b - a) * 1e3 for a, b, _ in bcast_timed)) / len(bcast_timed),
                "what": "conditioning KV of the shared prompt: rank 0 prefills, one flat bf16 buffer [L][2][rows][nkv*D] + an int64 header to every rank"},
            "understanding": und,
            "edit": edit,
            "taylorseer": ts,
            "fp8_gen_expert": fp8,
            "training_forward": trainf,
        }
        if args.workload == "t2i":
            # the whole path against the MFMA roof: denoise FLOPs (linear + attention of the 98 forwards, SURVEY.md 8d: 5.904 PFLOP
            # per image at the default shapes) over the whole step time incl. prefill, glue and VAE decode
            C = args.prompt_tokens + 2
            pf = (T - 1) * L * (_layer_flops(n_img + 2, C, cfg["llm"]["hidden_size"]) + _layer_flops(n_img + 2, 0, cfg["llm"]["hidden_size"]))
            out["whole_path_roofline"] = {"bound": "mfma", "achieved": pf * images / dt / 1e12 / world, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                          "frac": pf * images / dt / 1e12 / world / PEAK_BF16_TFLOPS, "denoise_pflop_per_image": pf / 1e15}
        if args.workload == "edit":
            out["metric"] = "images/sec (image edit 1024^2, 50-step, 3-forward CFG), 7B-MoT"
        if args.layers is not None or args.no_vae or R != 1024 or T != 50 or args.standins:
            out["valid"] = False
            out["note"] = "debug flags reduce the workload: not a benchmark number"
        if store_info is not None:
            out["valid"] = False
            out["weight_store"] = store_info
            out["note"] = "weight_store option (quantised decoder projections: changes results): beside the bf16 headline, never as it"
        if args.standins:
            out["standins"] = "tests/mock_ops.py torch stand-ins on the CPU, tiny model, gloo: exercises bench.main()'s host logic only"
            out["latents_checksum"] = [float(x.double().sum()) for x in latents]
        if world > 1:
            out["cpu_baseline"] = None      # timed on rank 0 at N=1 only: the host cores are shared by N ranks here
        elif not args.no_cpu_baseline:
            try:
                gpu = None if (args.layers is not None or args.workload != "t2i") else dict(model=model, tok=tok, ids=ids)
                out["cpu_baseline"] = cpu_baseline(args, cfg, gpu)
            except Exception as e:   # the baseline is reported, never required for the GPU number
                out["cpu_baseline"] = {"error": repr(e)}
        if ranks_seen != args.gpus or ranks_seen != world:
            # a job that did not run on the N ranks it was asked for has no line: a throughput quoted for N GPUs over fewer (or more) ranks
            # would be read as a scaling point
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but the process group has {world} rank(s) and its all-reduce saw {ranks_seen}: refusing to print a "
                             "benchmark line (launch with torch.distributed.run --nproc-per-node N, or let bench.py --gpus N start the ranks itself)\n")
            refused = True
        else:
            refused = False
            print(json.dumps(out), flush=True)
    else:
        refused = False
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if refused:
        sys.exit(3)


if __name__ == "__main__":
    main()
