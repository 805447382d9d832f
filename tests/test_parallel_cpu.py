"""world_size-2 gloo test of the DP driver pieces (sharding + conditioning-KV broadcast) on CPU."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bagel_amd.parallel import shard_range


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 32, 33):
        for ws in (1, 2, 4, 8):
            seen = []
            for r in range(ws):
                lo, hi = shard_range(n, r, ws)
                seen += list(range(lo, hi))
            assert seen == list(range(n))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, ws, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from bagel_amd.modeling.bagel.qwen2_navit import NaiveCache
    from bagel_amd.parallel import broadcast_cache
    L, nkv, hd, dp = 3, 2, 32, 64
    lens = [5, 9]
    cache = NaiveCache(L)
    if rank == 0:
        g = torch.Generator().manual_seed(0)
        cache._nkv, cache._hd, cache._dp, cache._total = nkv, hd, dp, sum(lens)
        for i in range(L):
            cache._k[i] = torch.randn(64, nkv * dp, generator=g).to(torch.bfloat16)   # capacity > used rows
            cache._v[i] = torch.randn(64, nkv * dp, generator=g).to(torch.bfloat16)
            cache._lens[i] = list(lens)
    got = broadcast_cache(cache, src=0)
    g = torch.Generator().manual_seed(0)
    ok = got.seq_lens == sum(lens) and got.lens(0) == lens
    for i in range(L):
        k = torch.randn(64, nkv * dp, generator=g).to(torch.bfloat16)[:14]
        v = torch.randn(64, nkv * dp, generator=g).to(torch.bfloat16)[:14]
        ok = ok and torch.equal(got.key_cache[i], k.view(14, nkv, dp)[..., :hd]) and torch.equal(got.value_cache[i], v.view(14, nkv, dp)[..., :hd])
    empty = broadcast_cache(NaiveCache(2), src=0)
    ok = ok and empty.seq_lens == 0
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_cache_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    assert res == {0: True, 1: True}


def _worker_edit_context(rank, ws, port, q):
    """The conditioning context of an image-edit request at its REAL geometry (configs[4]: 28 layers x 9 032 rows x 4 KV heads x 128 = 518 MB of bf16 K + V)
    through broadcast_cache: header + payload arrive bit for bit, the byte count is the payload's (VERDICT r05 'missing 6': the path had only seen 14-row caches)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from bagel_amd.modeling.bagel.qwen2_navit import NaiveCache
    from bagel_amd.parallel import broadcast_cache
    L, nkv, hd, dp, rows = 28, 4, 128, 128, 9032
    cache = NaiveCache(L)

    def layer(i, which):                         # cheap, reproducible on both ranks: a per-layer integer ramp in bf16 (exactly representable steps)
        base = torch.arange(rows * nkv * dp, dtype=torch.int32).view(rows, nkv * dp)
        return (((base * (2 * i + which + 1)) % 251) - 125).to(torch.bfloat16)
    if rank == 0:
        cache._nkv, cache._hd, cache._dp, cache._total = nkv, hd, dp, rows
        for i in range(L):
            k = torch.empty((rows + 40, nkv * dp), dtype=torch.bfloat16)           # capacity > used rows, like an appended-to context
            v = torch.empty_like(k)
            k[:rows], v[:rows] = layer(i, 0), layer(i, 1)
            cache._k[i], cache._v[i], cache._lens[i] = k, v, [rows]
    stats = {}
    got = broadcast_cache(cache, src=0, stats=stats)
    ok = got.seq_lens == rows and got.lens(0) == [rows]
    for i in (0, 13, 27):
        ok = ok and torch.equal(got.key_cache[i], layer(i, 0).view(rows, nkv, dp)) and torch.equal(got.value_cache[i], layer(i, 1).view(rows, nkv, dp))
    payload = L * 2 * rows * nkv * dp * 2
    ok = ok and payload <= stats["bytes"] <= payload + 4096
    q.put((rank, bool(ok), stats["bytes"]))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_cache_at_the_edit_requests_geometry_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker_edit_context, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=600) for _ in ps]
    for p in ps:
        p.join(timeout=120)
    assert sorted((r, ok) for r, ok, _ in res) == [(0, True), (1, True)], res
    assert all(b >= 28 * 2 * 9032 * 512 * 2 for _, _, b in res)


# ---- data-parallel text->image end to end on the host logic (2 ranks, gloo, torch stand-ins for the launch wrappers) --------
def _dp_setup(B, monkeypatch=None):
    """(model, cfg, prompt inputs for B identical prompts, sizes).  The launch wrappers are the CPU stand-ins of tests/mock_ops.py
    (test infrastructure): what is under test is the host-side DP protocol, not the kernels."""
    from oracle.configs import TINY, NEW_TOKEN_IDS_TINY, StubTokenizer
    from tests import mock_ops
    from tests.test_host_logic_cpu import cpu_model

    class _MP:
        def setattr(self, obj, name, val):
            setattr(obj, name, val)
    mock_ops.install(monkeypatch or _MP())      # worker processes keep the stand-ins for life; the pytest process restores them
    model = cpu_model(TINY)
    tok = StubTokenizer(TINY["llm"]["vocab_size"])
    return model, TINY, tok, NEW_TOKEN_IDS_TINY


def _dp_generate(model, cfg, tok, ids, cache, B, noise, renorm):
    from bagel_amd.modeling.bagel.qwen2_navit import NaiveCache
    L = cfg["llm"]["num_hidden_layers"]
    sizes = [(64, 32)] * B
    _, lens, ropes = model.prepare_prompts([0] * B, [0] * B, ["a small red cube"] * B, tok, ids)
    li = model.prepare_vae_latent(lens, ropes, sizes, ids)
    li["packed_init_noises"] = noise
    ci = model.prepare_vae_latent_cfg([0] * B, [0] * B, sizes)
    return model.generate_image(past_key_values=cache, num_timesteps=4, timestep_shift=3.0, cfg_text_scale=4.0, cfg_interval=[0, 1.0],
                                cfg_renorm_min=0.0, cfg_renorm_type=renorm, cfg_text_past_key_values=NaiveCache(L),
                                cfg_text_packed_position_ids=ci["cfg_packed_position_ids"],
                                cfg_text_packed_query_indexes=ci["cfg_packed_query_indexes"],
                                cfg_text_key_values_lens=ci["cfg_key_values_lens"],
                                cfg_text_packed_key_value_indexes=ci["cfg_packed_key_value_indexes"], **li)


def _dp_worker(rank, ws, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from bagel_amd.modeling.bagel.qwen2_navit import NaiveCache
    from bagel_amd.parallel import broadcast_cache, shard_range
    B_total, n_img = 4, 4 * 2
    per = B_total // ws
    model, cfg, tok, ids = _dp_setup(per)
    L = cfg["llm"]["num_hidden_layers"]
    # the conditioning context: rank 0 prefills, everybody receives it (bench.py / SURVEY.md 8e.1)
    cache = NaiveCache(L)
    if rank == 0:
        gi, _, _ = model.prepare_prompts([0] * per, [0] * per, ["a small red cube"] * per, tok, ids)
        cache = model.forward_cache_update_text(cache, **gi)
    cache = broadcast_cache(cache, src=0)
    all_noise = torch.randn(B_total * n_img, 64, generator=torch.Generator().manual_seed(42))
    lo, hi = shard_range(B_total, rank, ws)
    mine = all_noise[lo * n_img:hi * n_img]
    out = {}
    for tag, renorm, allreduce in (("channel", "channel", False), ("global_local", "global", False), ("global_allreduce", "global", True)):
        model.global_renorm_allreduce = allreduce
        out[tag] = [t.clone() for t in _dp_generate(model, cfg, tok, ids, cache, per, mine, renorm)]
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_text_to_image_world2(monkeypatch):
    """Two ranks x two samples vs ONE process with all four samples, identical prompts, the job-global seed-42 noise stream sharded
    by rank: per-token renorm -> every sample bit-identical to the single-process batch (no tensor crosses a
    rank during sampling); 'global' renorm -> per-rank statistics by default (what the reference's torchrun drivers do: differs from
    the 4-sample batch), equal to it with model.global_renorm_allreduce (two fp32 sums all-reduced per step)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=300) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    # single-process reference: all four samples in one batch
    from bagel_amd.modeling.bagel.qwen2_navit import NaiveCache
    model, cfg, tok, ids = _dp_setup(4, monkeypatch)
    gi, _, _ = model.prepare_prompts([0] * 4, [0] * 4, ["a small red cube"] * 4, tok, ids)
    cache = model.forward_cache_update_text(NaiveCache(cfg["llm"]["num_hidden_layers"]), **gi)
    noise = torch.randn(4 * 8, 64, generator=torch.Generator().manual_seed(42))
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()  # noqa: E731
    one = {r: _dp_generate(model, cfg, tok, ids, cache, 4, noise, r) for r in ("channel", "global")}
    dp = {tag: res[0][tag] + res[1][tag] for tag in res[0]}
    assert all(torch.equal(a, b) for a, b in zip(dp["channel"], one["channel"]))
    assert max(rel(a, b) for a, b in zip(dp["global_allreduce"], one["global"])) < 1e-3
    assert max(rel(a, b) for a, b in zip(dp["global_local"], one["global"])) > 1e-3, "per-rank statistics should be visible"


def test_bench_self_launches_n_ranks():
    """`python bench.py --gpus 2` outside torchrun re-execs itself as 2 ranks (torch.distributed.run on 127.0.0.1) and rank 0
    prints ONE JSON line with n_gpus = 2; the same command line under an existing torchrun environment must not fork again.
    (--launch-check = the distributed skeleton of bench.main() without the model; gloo on a box without GPUs.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["CUDA_VISIBLE_DEVICES"] = env["HIP_VISIBLE_DEVICES"] = ""        # the skeleton's gloo leg, also on a GPU box
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0", "--launch-check"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["config"]["global_batch"] == 8 and out["config"]["parallelism"] == "dp2"
    assert out["steps"] == 2 and out["ms_per_step"] > 0
    # N = 1 stays in-process
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--launch-check"],
                        env=env, capture_output=True, text=True, timeout=300)
    assert r1.returncode == 0 and json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][0])["n_gpus"] == 1
    # a WORLD_SIZE that contradicts --gpus is still refused (the driver's torchrun line always agrees)
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], env=env2, capture_output=True, text=True, timeout=300)
    assert r2.returncode != 0 and "WORLD_SIZE" in r2.stderr


def test_bench_one_step_on_two_gloo_ranks():
    """bench.py's ACTUAL main() -- one_step() (prompt prefill on rank 0, conditioning-KV broadcast, rank-sharded seed-42 noise,
    generate_image with the stream-batched CFG forward, VAE decode), the fence, the max-over-ranks timing and the JSON assembly -- on 2
    gloo ranks with the torch stand-ins in place of the launch wrappers (``--standins``: test-only mode, line flagged invalid).
    The line must carry what the first real 8-GPU run will be read by: n_gpus, ranks_seen (an all-reduce of ones), the collective
    backend, the broadcast's size and time, a whole-job value; and rank 0's samples must be bit-identical to the N = 1 run's (it owns
    the same rows of the job-global noise stream and the default `global` renorm is per rank, like the reference's torchrun drivers:
    gen_images_mp.py:127-130,188-190)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["CUDA_VISIBLE_DEVICES"] = env["HIP_VISIBLE_DEVICES"] = ""
    env["OMP_NUM_THREADS"] = "2"
    common = ["--standins", "--batch", "2", "--resolution", "64", "--num-timesteps", "4", "--prompt-tokens", "6"]

    def run(n, steps, warmup):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", str(steps), "--warmup", str(warmup)] + common,
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        return json.loads(lines[0])
    two = run(2, 2, 1)
    assert two["n_gpus"] == 2 and two["ranks_seen"] == 2 and two["collective_backend"] == "gloo" and two["valid"] is False
    assert two["steps"] == 2 and two["warmup"] == 1 and two["scaling"] == "weak" and two["outputs_finite"] is True
    assert two["config"]["global_batch"] == 4 and two["config"]["parallelism"] == "dp2"
    assert abs(two["value"] - 4 * 2 / (two["ms_per_step"] * 2 / 1e3)) < 1e-6 * two["value"]          # whole-job images / max-over-ranks time
    b = two["broadcast"]
    # 2 layers x (K, V) x 2 samples x 8 prompt rows x (2 KV heads x 64 padded lanes) bf16 + the int64 header (count, L, nkv, hd, dp, 2 lens)
    assert b["calls"] == 2 and b["bytes_per_call"] == 2 * 2 * 16 * 128 * 2 + 8 * (1 + 4 + 2) and b["ms_per_call"] > 0
    assert two["cpu_baseline"] is None and two["understanding"] is None
    tr = two["training_forward"]                        # the training legs run on every rank (fence inside): forward, then forward with tape + backward
    assert tr["outputs_finite"] is True and tr["value"] > 0
    ts = tr["training_step"]
    assert "error" not in ts and ts["finite"] is True and ts["grad_norm"] > 0 and ts["trainable_params"] > 0 and ts["ms_backward"] > 0
    one = run(1, 1, 0)
    assert one["n_gpus"] == 1 and one["ranks_seen"] == 1 and one["broadcast"] is None and one["collective_backend"] is None
    assert one["latents_checksum"] == two["latents_checksum"], "rank 0 of the 2-rank job must reproduce the 1-rank job's samples"


def _ddp_worker(rank, ws, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from tests import mock_ops
    mock_ops.install_permanently()
    from oracle.configs import TINY
    from oracle import bagel_oracle as O
    from tests.util_models import oracle_weights, pack_training_batch
    from bagel_amd.factory import build_bagel
    W, _ = oracle_weights(TINY)
    model, _ = build_bagel(TINY, device="cpu", with_vae=False)
    model.load_state_dict(W, strict=True)
    model = model.to(torch.bfloat16).eval()
    for n, p in model.named_parameters():
        p.requires_grad_(not n.startswith(("vit_pos_embed.", "latent_pos_embed.")))

    def batch_of(r):
        samples = [[("text", 3, True), ("vit", 28, 42), ("text", 2 + r, True)], [("text", 2, False), ("vae", 32, 32 + 16 * r, True)]]
        return pack_training_batch(TINY, samples, 100 + r)
    local = {}
    for r in range(ws):                                    # every rank's gradients, computed here without any exchange
        b, noise, _, _ = batch_of(r)
        for p in model.parameters():
            p.grad = None
        O.training_step_loss(model(noise=noise, **b)).backward()
        local[r] = {n: p.grad.float().clone() for n, p in model.named_parameters() if p.grad is not None}
    for p in model.parameters():
        p.grad = None
    ddp = torch.nn.parallel.DistributedDataParallel(model)
    b, noise, _, _ = batch_of(rank)
    O.training_step_loss(ddp(noise=noise, **b)).backward()  # this rank's batch; DDP all-reduces what the packed autograd node returns
    worst, n_grads = 0.0, 0
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        want = sum(local[r][n] for r in range(ws)) / ws
        if float(want.norm()) == 0.0:
            continue
        worst = max(worst, float((p.grad.float() - want).norm() / want.norm()))
        n_grads += 1
    q.put((rank, worst, n_grads))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_training_step_under_ddp_gloo_world2():
    """Training shards by batch and its exchange step is the gradient all-reduce: the packed forward is ONE autograd node that hands
    ordinary gradient tensors to torch, so torch's DistributedDataParallel (RCCL on the GPUs, gloo here) averages them like autograd's --
    two ranks with different packs end with the mean of the two ranks' gradients (to one bf16 rounding of the averaged tensors)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=240) for _ in range(2)]
    for p in ps:
        p.join(60)
    for rank, worst, n_grads in res:
        assert n_grads >= 100 and worst < 1e-2, (rank, worst, n_grads)
