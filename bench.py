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
    ap.add_argument("--edit-batch", type=lambda v: [int(x) for x in v.split(",") if x], default=[2, 4],
                    help="edit.batched (the first value) / edit.batched_<n> (the others): this many edit requests per GPU served as ONE batch, beside the one-request "
                         "latency; every sample of the first size is also checked against its own single-request run")
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
    ap.add_argument("--no-memory-leg", action="store_true",
                    help="skip memory.t2i_b1_request (three B = 1 requests in front of the warm-up): the rocprofv3 summaries of a bench step must not average their launches in")
    ap.add_argument("--no-oracle-jobs", action="store_true",
                    help="run the CPU-oracle sides of understanding.parity_at_full_depth and edit.parity_at_depth in this process, after the GPU legs, instead of in "
                         "worker processes beside the headline loop")
    ap.add_argument("--oracle-job", choices=["und_depth", "edit_depth"], default=None, help="internal: worker mode (see start_oracle_jobs)")
    ap.add_argument("--job-dir", default=None, help="internal: worker mode")
    ap.add_argument("--job-threads", type=int, default=0, help="internal: worker mode")
    ap.add_argument("--job-node", type=int, default=-1, help="internal: worker mode (NUMA node to pin to, -1 = none)")
    ap.add_argument("--und-oracle-out", default=None, help="internal: directory of the und_depth worker whose result the understanding child compares with")
    ap.add_argument("--wall-budget-s", type=float, default=1200.0,
                    help="the CPU-oracle parity legs are dropped (and the line says so) once the run would pass this many seconds (the driver's limit is 1800)")
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


T_START = time.time()
EDIT_PARITY_LAYERS = 4      # depth of the model edit.parity_at_depth runs (tests/test_full_depth_gpu.py::test_edit_request_at_full_context uses the same)


def numa_nodes():
    """-> [[cpu ids of node 0], [node 1], ...] from sysfs, or [] when it cannot be read."""
    import glob
    import re
    nodes = []
    for path in sorted(glob.glob("/sys/devices/system/node/node[0-9]*/cpulist"), key=lambda p: int(re.search(r"node(\d+)", p).group(1))):
        cpus = []
        try:
            for part in open(path).read().strip().split(","):
                if part:
                    a, _, b = part.partition("-")
                    cpus += list(range(int(a), int(b or a) + 1))
        except (OSError, ValueError):
            return []
        nodes.append(cpus)
    return nodes


def oracle_job_main(args):
    """WORKER MODE (``--oracle-job``): the CPU-oracle side of one parity leg in a process of its own, so that it runs BESIDE the parent's GPU-bound headline loop
    instead of after it (round-5 verdict, weak 16: the two full-depth oracle legs were 5.5 min of a 15-min run with the GPU idle).  The worker builds the same
    name-seeded random model on the GPU (bagel_amd/factory.py init_random_: every tensor's generator is seeded by its NAME, so this process, the parent and the
    understanding child hold identical weights -- the compare phase checks a fingerprint), copies the weights to the host, RELEASES the GPU (``gpu_released`` flag:
    the parent waits for it before its timed region) and only then starts the oracle on its own share of the host cores.  Checker use of the oracle only."""
    d = args.job_dir
    try:
        nodes = numa_nodes()
        if 0 <= args.job_node < len(nodes) and len(nodes) > 1:
            os.sched_setaffinity(0, nodes[args.job_node])
        threads = args.job_threads or max(1, physical_cores() // 2)
        torch.set_num_threads(threads)
        from bagel_amd.factory import BAGEL_7B_MOT as cfg, NEW_TOKEN_IDS_QWEN25 as ids, build_bagel, init_random_
        dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
        real_edit = args.oracle_job == "edit_depth"
        L = EDIT_PARITY_LAYERS if real_edit else (args.layers or cfg["llm"]["num_hidden_layers"])
        model, vae = build_bagel(cfg, device=dev, num_layers=L, with_vae=real_edit)
        init_random_(model, seed=0)
        model.llm2vae.weight.data.normal_(0, cfg["llm"]["hidden_size"] ** -0.5, generator=torch.Generator(device=dev).manual_seed(1))
        cfgL = dict(cfg, llm=dict(cfg["llm"], num_hidden_layers=L))
        VW = None
        if real_edit:
            init_random_(vae, seed=0)
            inp = edit_depth_inputs(args, cfgL, model, ids, real=True)
            VW = {k: v.detach().float().cpu() for k, v in vae.state_dict().items()}
            keep = EDIT_KEEP_REAL
        else:
            inp = und_depth_inputs(args, model, ids)
            keep = UND_KEEP
        W = {k: v.detach().to("cpu") for k, v in model.state_dict().items() if k.startswith(keep)}
        del model, vae
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        open(os.path.join(d, "gpu_released"), "w").write(str(time.time()))
        t0 = time.time()
        out = edit_depth_oracle(cfgL, W, VW, inp, threads) if real_edit else und_depth_oracle(cfgL, W, inp, threads)
        out["worker"] = {"seconds": time.time() - t0, "seconds_note": "wall clock, INCLUDING the time this process spent stopped while the parent's timed regions ran", "threads": threads, "numa_node": args.job_node if len(nodes) > 1 else None,
                         "cpus_allowed": len(os.sched_getaffinity(0)), "started_after_bench_start_s": t0 - float(os.environ.get("BAGEL_BENCH_T0", t0))}
        torch.save(out, os.path.join(d, "out.pt.tmp"))
        os.rename(os.path.join(d, "out.pt.tmp"), os.path.join(d, "out.pt"))
    except BaseException as e:        # the parent falls back to the in-process oracle
        import traceback
        open(os.path.join(d, "gpu_released"), "a").write("")
        open(os.path.join(d, "error.txt"), "w").write(repr(e) + "\n" + traceback.format_exc())
        raise


def start_oracle_jobs(args, local, kinds):
    """Start one worker per kind (see ``oracle_job_main``); -> {kind: dict(dir, proc)}.  The workers share the host cores between them and leave 1/8 to the
    parent (its launch thread) -- on a two-socket box each worker is pinned to its own NUMA node."""
    import subprocess
    import tempfile
    jobs = {}
    if not kinds:
        return jobs
    phys, nodes = physical_cores(), numa_nodes()
    threads = max(4, (phys - max(8, phys // 8)) // len(kinds))
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK=str(local), LOCAL_WORLD_SIZE="1", BAGEL_BENCH_T0=str(T_START), OMP_NUM_THREADS=str(threads),
               BAGEL_PAUSE_PIDS="")
    for i, kind in enumerate(kinds):
        d = tempfile.mkdtemp(prefix=f"bagel_oracle_{kind}_")
        cmd = [sys.executable, os.path.abspath(__file__), "--oracle-job", kind, "--job-dir", d, "--job-threads", str(threads),
               "--job-node", str(i % len(nodes) if len(nodes) > 1 else -1), "--resolution", str(args.resolution), "--und-image", str(args.und_image)]
        if args.layers is not None:
            cmd += ["--layers", str(args.layers)]
        log = open(os.path.join(d, "log.txt"), "w")
        jobs[kind] = dict(dir=d, proc=subprocess.Popen(cmd, env=env, stdout=log, stderr=subprocess.STDOUT), threads=threads)
        Steady.pause_pids.append(jobs[kind]["proc"].pid)

    def _cleanup():          # a parent that dies inside a timed region must not leave stopped workers behind
        import signal
        for j in jobs.values():
            if j["proc"].poll() is None:
                try:
                    os.kill(j["proc"].pid, signal.SIGCONT)
                    j["proc"].kill()
                except OSError:
                    pass
    import atexit
    atexit.register(_cleanup)
    return jobs


def wait_gpu_released(jobs, timeout=300.0):
    """Block until every worker has copied its weights off the GPU (or died): nothing but this process may be on the device inside a timed region."""
    t0 = time.time()
    for j in jobs.values():
        while not os.path.exists(os.path.join(j["dir"], "gpu_released")) and j["proc"].poll() is None and time.time() - t0 < timeout:
            time.sleep(0.25)
        j["gpu_released"] = os.path.exists(os.path.join(j["dir"], "gpu_released"))
        if not j["gpu_released"] and j["proc"].poll() is None:      # still on the GPU after the timeout: stop it, the in-process oracle takes over
            j["proc"].kill()
            j["proc"].wait()
    return time.time() - t0


def wait_oracle_job(d, kind, timeout=1500.0, proc=None):
    """-> the worker's result, or None (failed / timed out: the caller computes the oracle itself)."""
    t0 = time.time()
    out = os.path.join(d, "out.pt")
    while not os.path.exists(out):
        if os.path.exists(os.path.join(d, "error.txt")) or (proc is not None and proc.poll() not in (None, 0)) or time.time() - t0 > timeout:
            sys.stderr.write(f"bench.py: oracle worker {kind} gave no result ({d}): falling back to the in-process oracle\n")
            return None
        time.sleep(0.5)
    r = torch.load(out, weights_only=False)
    r["waited_s"] = time.time() - t0
    return r


def over_budget(args, need_s, what):
    """The wall-clock guard of the CPU-oracle legs: None, or the reason the leg is dropped."""
    used = time.time() - T_START
    if used + need_s > args.wall_budget_s:
        return {"skipped": f"{what}: {used:.0f} s into the run, the leg needs ~{need_s:.0f} s, --wall-budget-s {args.wall_budget_s:.0f} (the driver stops bench.py at 1800 s)"}
    return None


class Steady:
    """Memory bracket of ONE timed region: the peak of live tensor bytes inside it (after ``reset_peak_memory_stats``) and what the region asked of the
    allocator.  A timed region is in the steady state iff the caching allocator served it WITHOUT going to the driver: ``num_device_alloc`` /
    ``num_device_free`` (hipMalloc / hipFree calls) must not move -- a fresh hipMalloc inside the clock has corrupted two records before (r04 batched decode,
    r05 prefill: 85 vs 160 ms).  ``cached_block_requests`` (``allocation.all.allocated``: tensors handed out from the cache) is reported, not gated: the host
    code allocates its per-step tensors from the cache by design.  No GPU (the --standins test mode): every figure is None and the region counts as steady."""

    # worker processes of this run (start_oracle_jobs) that are STOPPED for the duration of every timed region: measured on MI355X (round 6, first visit), two
    # CPU-oracle workers beside the GPU legs cost the one-request edit leg 15 % (10.09 vs 8.77 s/image), TaylorSeer 10 %, the FP8 leg 5 % and the headline step
    # 4 % -- the launch thread shares cores, caches and the host memory system with them.  SIGSTOP / SIGCONT: the workers only run beside UNTIMED work (model
    # builds, warm-ups, the checker's own product-side runs); the understanding child inherits the pids through BAGEL_PAUSE_PIDS.
    pause_pids = [int(x) for x in os.environ.get("BAGEL_PAUSE_PIDS", "").split(",") if x]

    @staticmethod
    def _is_worker(pid):
        """Still one of OUR oracle workers?  (A finished worker's pid may be recycled by the system: never signal a process that is not `bench.py --oracle-job`.)"""
        try:
            with open(f"/proc/{pid}/cmdline", "rb") as f:
                cmd = f.read()
            with open(f"/proc/{pid}/stat") as f:
                zombie = f.read().rsplit(")", 1)[1].split()[0] == "Z"
            return b"--oracle-job" in cmd and not zombie
        except (OSError, IndexError):
            return False

    @classmethod
    def _signal_workers(cls, sig):
        for pid in list(cls.pause_pids):
            if not cls._is_worker(pid):
                cls.pause_pids.remove(pid)
                continue
            try:
                os.kill(pid, sig)
            except (ProcessLookupError, PermissionError):
                cls.pause_pids.remove(pid)

    def __init__(self, dev=None):
        self.cuda = torch.cuda.is_available() and (dev is None or torch.device(dev).type == "cuda")
        self.dev = dev

    def __enter__(self):
        import signal
        self._signal_workers(signal.SIGSTOP)
        if self.cuda:
            torch.cuda.synchronize(self.dev)
            torch.cuda.reset_peak_memory_stats(self.dev)
            self._s0 = torch.cuda.memory_stats(self.dev)
        return self

    def __exit__(self, *exc):
        if self.cuda:
            torch.cuda.synchronize(self.dev)
            s0, s1 = self._s0, torch.cuda.memory_stats(self.dev)
            d = lambda k: int(s1.get(k, 0) - s0.get(k, 0))  # noqa: E731
            self.report = {"peak_mem_gb": torch.cuda.max_memory_allocated(self.dev) / 2 ** 30, "peak_reserved_gb": torch.cuda.max_memory_reserved(self.dev) / 2 ** 30,
                           "device_allocs_in_timed_region": d("num_device_alloc"), "device_frees_in_timed_region": d("num_device_free"),
                           "segments_delta": d("segment.all.current"), "cached_block_requests": d("allocation.all.allocated"),
                           "alloc_retries": d("num_alloc_retries")}
            self.report["steady"] = self.report["device_allocs_in_timed_region"] == 0 and self.report["device_frees_in_timed_region"] == 0
        else:
            self.report = {"peak_mem_gb": None, "steady": True}
        import signal
        self._signal_workers(signal.SIGCONT)
        return False


def timed_steady(fn, dev, fence, attempts=2):
    """One timed region, fenced on both sides, under the ``Steady`` bracket; a region that made the allocator go to the driver is run again (the first pass was
    its warm-up) -- and a leg that is STILL not steady is failed by its caller (``steady`` False in the report) rather than quoted.  -> (fn's result, seconds, report)"""
    for attempt in range(attempts):
        fence()
        with Steady(dev) as m:
            t1 = time.perf_counter()
            r = fn()
            fence()
            dt = time.perf_counter() - t1
        if m.report["steady"]:
            break
    return r, dt, dict(m.report, attempts=attempt + 1)


def unsteady(leg, mem):
    """Fail a leg whose timed region allocated from the driver even on its second pass (round-5 verdict, weak 17)."""
    if leg is not None and not mem.get("steady", True):
        leg["error"] = (f"{mem['device_allocs_in_timed_region']} hipMalloc / {mem['device_frees_in_timed_region']} hipFree call(s) inside the timed region on "
                        f"pass {mem['attempts']}: not a steady-state measurement, value not to be quoted")
    return leg


def resident_weight_bytes(model, vae=None):
    """What the model keeps resident, itemised: the reference-layout nn.Parameters / buffers (what a checkpoint holds) and the MI355X-layout packed copies the
    engines keep BESIDE them (modeling/packed.py: fused Wqkv, 16-row interleaved gate/up, padded heads; tensors that merely alias a parameter are not counted)."""
    seen = set()

    def nbytes(ts):
        n = 0
        for t in ts:
            if torch.is_tensor(t) and t.numel() and t.untyped_storage().data_ptr() not in seen:
                seen.add(t.untyped_storage().data_ptr())
                n += t.untyped_storage().nbytes()
        return n
    out = {"parameters_and_buffers_gb": nbytes(list(model.parameters()) + list(model.buffers())) / 2 ** 30}
    if vae is not None:
        out["vae_parameters_gb"] = nbytes(list(vae.parameters()) + list(vae.buffers())) / 2 ** 30
    packed = {}
    try:
        eng = model.language_model.engine()
        for name in ("wqkv", "bqkv", "wo", "wgu", "wd"):
            packed["llm." + name] = nbytes([t for P in eng.layers for t in getattr(P, name)]) / 2 ** 30
    except Exception as e:        # a quantised store has another layout: reported as such
        packed["llm"] = repr(e)
    try:                          # the SigLIP tower's packed copies exist once it has run (heads padded 72 -> 128, MLP width padded to the k-tile)
        vp = getattr(model.vit_model, "_packed", None)
        if vp:
            for name in ("wqkv", "bqkv", "wo"):
                packed["vit." + name] = nbytes([L[name] for L in vp["layers"]]) / 2 ** 30
            packed["vit.fc1p"] = nbytes([L["fc1p"][0] for L in vp["layers"]] + [L["fc1p"][1] for L in vp["layers"]]) / 2 ** 30
            packed["vit.fc2p"] = nbytes([L["fc2p"][0] for L in vp["layers"]]) / 2 ** 30
    except Exception as e:
        packed["vit"] = repr(e)
    out["packed_copies_gb"] = packed
    out["packed_copies_total_gb"] = sum(v for v in packed.values() if isinstance(v, float))
    return out


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
        # THE TIMED PATH: the default stream-batched generate_image; its per-stream velocities are taken BEFORE the combine (model.velocity_hook) and each is
        # gated like a single forward (round-5 verdict: a defect confined to the batched work list must not hide in the CFG-combined band)
        got = {}
        model.velocity_hook = lambda batched, vs: got.update(batched=batched, vs=[None if v is None else v.float().cpu() for v in vs])  # noqa: E731
        try:
            lat = model.generate_image(past_key_values=cache, num_timesteps=2, cfg_text_scale=4.0, cfg_interval=[0, 1.0], cfg_renorm_min=0.0,
                                       cfg_renorm_type="global", timestep_shift=3.0, **ckw, **li)
        finally:
            model.velocity_hook = None
        v_gpu = x0 - torch.cat([t.float().cpu() for t in lat])
        out["rel_l2"] = rel(v_gpu, v_cpu)
        out["stream_batched"] = {"ran_batched": bool(got.get("batched")),
                                 "rel_l2_cond_forward": rel(got["vs"][0], parts["v_cond"].float()), "rel_l2_cfg_text_forward": rel(got["vs"][1], parts["v_cfg_text"].float()),
                                 "vs_own_sequential_forwards": [rel(got["vs"][0], v_c.float().cpu()), rel(got["vs"][1], v_u.float().cpu())],
                                 "tolerance_forward": FULL_DEPTH_TOL_FORWARD,
                                 "what": "per-stream velocities of the ONE stream-batched forward inside generate_image (the timed path), taken before the CFG combine, "
                                         "each against the oracle's single forward"}
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
    sc, sb = out["cfg_combine_self_consistency"], out["stream_batched"]
    out["within_tolerance"] = bool(out["rel_l2_cond_forward"] <= FULL_DEPTH_TOL_FORWARD and out["rel_l2_cfg_text_forward"] <= FULL_DEPTH_TOL_FORWARD
                                   and sb["ran_batched"] and sb["rel_l2_cond_forward"] <= FULL_DEPTH_TOL_FORWARD and sb["rel_l2_cfg_text_forward"] <= FULL_DEPTH_TOL_FORWARD
                                   and sc["sequential_forward_flow"] <= FULL_DEPTH_TOL_COMBINE and sc["generate_image_sequential"] <= FULL_DEPTH_TOL_COMBINE
                                   and sc["generate_image_stream_batched"] <= FULL_DEPTH_TOL)
    out["gate"] = (f"every single forward <= {FULL_DEPTH_TOL_FORWARD} (1.5 x the reference's own single-forward accumulation-order noise): the two sequential forwards AND "
                   f"the two per-stream velocities of the stream-batched forward that generate_image (the timed path) runs; AND the combine step on the product's own two "
                   f"forwards <= {FULL_DEPTH_TOL_COMBINE}; the CFG-combined rel_l2 vs the oracle is information only")
    out["cfg_combined_inside_noise_band"] = bool(out["rel_l2"] <= FULL_DEPTH_TOL and out["rel_l2_sequential_forward_flow"] <= FULL_DEPTH_TOL)
    out["noise_floor"] = {"cfg_combined_velocity": 0.080, "single_forward_velocity": 0.016, "source": "profiles/r03_full_depth_noise_floor.log"}
    return out


EDIT_DEPTH_TOL_BATCHED = 0.25     # three forwards, CFG 4.0 x 2.0: the single-forward noise is amplified ~2x further than in the two-forward step
EDIT_CONTEXT_TOL_KV = 0.0226      # per-layer K / V of a prefilled context: the understanding leg's bound (1.5 x the reference's own noise at depth 28: UND_DEPTH_TOL_KV)


def _cpu_tree(x):
    """Host copies of every tensor in a (nested) packer output."""
    if torch.is_tensor(x):
        return x.detach().cpu()
    if isinstance(x, dict):
        return {k: _cpu_tree(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_cpu_tree(v) for v in x)
    return x


def edit_depth_inputs(args, cfg, model, ids, ctx_tokens=(576, 30), real=False, prompt_tokens=30, vit_side=980):
    """PHASE 1 (host only): every input of the edit-request parity step -- the packer outputs that build the three contexts (the product's packers are bit-exact
    with the reference's, so one set feeds product and oracle), the denoise step's index tensors, the seeded noise.
    ``real=False``: the image context is STOOD IN FOR by ``ctx_tokens[0]`` text tokens.  ``real=True``: the request chain of configs[4] itself
    (inferencer.py:62-97 update_context_image(vae=True, vit=True), :40-60 update_context_text; bagel.py:417-550,299-415): the R x R source image for the VAE
    (seeded reparameterisation draw), its ``vit_side``^2 view for the SigLIP tower, then the prompt => 9 032 / 9 000 / 32-key contexts at R = 1024."""
    R = args.resolution
    inp = {"real": bool(real), "R": R}
    if not real:
        g = torch.Generator().manual_seed(11)
        V = cfg["llm"]["vocab_size"]
        tok_img = FixedTokenizer(torch.randint(8, V - 8, (ctx_tokens[0],), generator=g).tolist())
        tok_txt = FixedTokenizer(torch.randint(8, V - 8, (ctx_tokens[1],), generator=g).tolist())
        gi1, l1, r1 = model.prepare_prompts([0], [0], ["image"], tok_img, ids)
        gi2, l2, r2 = model.prepare_prompts(l1, r1, ["prompt"], tok_txt, ids)
        gi3, l3, r3 = model.prepare_prompts([0], [0], ["prompt"], tok_txt, ids)
        inp.update(image_text=gi1, prompt=gi2, prompt_alone=gi3, cond=(l2, r2), text=(l1, r1), img=(l3, r3), chain="text stand-in for the image context")
    else:
        gsrc = torch.Generator().manual_seed(3)
        src_vae = torch.rand(3, R, R, generator=gsrc) * 2 - 1                   # vae_transform(image) stand-in
        src_vit = torch.rand(3, vit_side, vit_side, generator=gsrc) * 2 - 1     # vit_transform(image) stand-in
        inp["enc_noise"] = torch.randn(1, cfg["vae"]["z_channels"], R // 8, R // 8, generator=torch.Generator().manual_seed(43))
        tok = FixedTokenizer(torch.randint(8, min(151643, cfg["llm"]["vocab_size"] - 8), (prompt_tokens,), generator=torch.Generator().manual_seed(1)).tolist())
        ident = lambda t: t  # noqa: E731
        vi, l1, r1 = model.prepare_vae_images([0], [0], [src_vae], ident, ids)
        ti, l2, r2 = model.prepare_vit_images(l1, r1, [src_vit], ident, ids)
        pi, l3, r3 = model.prepare_prompts(l2, r2, ["p"], tok, ids)
        pi2, l4, r4 = model.prepare_prompts([0], [0], ["p"], tok, ids)
        inp.update(vae_image=vi, vit_image=ti, prompt=pi, prompt_alone=pi2, cond=(l3, r3), text=(l2, r2), img=(l4, r4),
                   chain=f"{R}^2 VAE-encode (bf16 autocast, seeded draw) -> gen-mode prefill -> {vit_side}^2 SigLIP + connector -> und-mode prefill -> "
                         f"{prompt_tokens}+2 prompt tokens")
    (l2, r2), (l1, r1), (l3, r3) = inp["cond"], inp["text"], inp["img"]
    li = model.prepare_vae_latent(l2, r2, [(R, R)], ids)
    li["packed_init_noises"] = torch.randn(li["packed_init_noises"].shape, generator=torch.Generator().manual_seed(4343))
    inp.update(latent=li, cfg_text=model.prepare_vae_latent_cfg(l1, r1, [(R, R)]), cfg_img=model.prepare_vae_latent_cfg(l3, r3, [(R, R)]))
    return _cpu_tree(inp)


EDIT_KEEP = ("language_model.model.", "time_embedder.", "vae2llm.", "llm2vae.", "latent_pos_embed.")
EDIT_KEEP_REAL = EDIT_KEEP + ("vit_model.", "connector.", "vit_pos_embed.")


def weight_fingerprint(W):
    """A few numbers that identify a set of weights (the oracle may run in ANOTHER process than the product it is compared with: both sides must hold the same
    name-seeded random initialisation, bagel_amd/factory.py init_random_)."""
    names = sorted(W)
    pick = [names[0], names[len(names) // 2], names[-1]] + [n for n in names if n.endswith("llm2vae.weight") or n.endswith("layers.0.mlp_moe_gen.down_proj.weight")]
    return {n: float(W[n].float().double().sum()) for n in pick}


def check_fingerprint(model, ora, L, who):
    """The oracle worker's weights == this model's (fp64 sums of the fingerprinted tensors; the summation order differs between host and device)."""
    sd = model.state_dict()
    mine = {n: float(sd[n].float().double().sum()) for n in ora["weights"]}
    bad = [n for n, v in ora["weights"].items() if abs(mine[n] - v) > 1e-6 * max(1.0, abs(v))]
    if bad or ora["layers"] != L:
        raise RuntimeError(f"{who}: the oracle worker ran on other weights than this model's (layers {ora['layers']} vs {L}; {[(n, ora['weights'][n], mine[n]) for n in bad]})")


def edit_depth_oracle(cfg, W, VW, inp, threads):
    """PHASE 2 (CPU only, checker): the three contexts and the 3-forward step of ``inp`` through the oracle; may run in a worker process beside the GPU legs
    (``--oracle-job``).  -> the per-layer K / V of the three contexts, the three single-forward velocities and the combined one."""
    import copy
    from oracle import bagel_oracle as O
    torch.set_num_threads(threads)
    L = cfg["llm"]["num_hidden_layers"]
    tt = {}
    t0 = t1 = time.time()
    if not inp["real"]:
        ocache = O.forward_cache_update_text(W, cfg, O.OracleCache(L), **inp["image_text"])
    else:
        try:
            O.VAE_AUTOCAST = "cuda"      # the policy of the device the reference runs on (group_norm in fp32): what the MI355X bf16 VAE implements
            ocache = O.forward_cache_update_vae(W, cfg, VW, O.OracleCache(L), sample_noise=inp["enc_noise"].to(torch.bfloat16), **inp["vae_image"])
        finally:
            O.VAE_AUTOCAST = None
        tt["vae_encode_plus_prefill"] = time.time() - t1
        t1 = time.time()
        ocache = O.forward_cache_update_vit(W, cfg, ocache, **inp["vit_image"])
        tt["siglip_plus_prefill"] = time.time() - t1
    octext = copy.deepcopy(ocache)
    t1 = time.time()
    ocache = O.forward_cache_update_text(W, cfg, ocache, **inp["prompt"])
    ocimg = O.forward_cache_update_text(W, cfg, O.OracleCache(L), **inp["prompt_alone"])
    tt["text_prefills"] = time.time() - t1
    li, ct, cim = inp["latent"], inp["cfg_text"], inp["cfg_img"]
    x0 = li["packed_init_noises"]
    ts = torch.tensor([1.0] * x0.shape[0])
    od = lambda c, d: dict(cache=c, position_ids=d["cfg_packed_position_ids"], query_indexes=d["cfg_packed_query_indexes"],  # noqa: E731
                           key_values_lens=d["cfg_key_values_lens"], key_value_indexes=d["cfg_packed_key_value_indexes"])
    t1 = time.time()
    parts = {}        # the three single-forward velocities of the SAME pass, before the combine (the oracle is deterministic: a stand-alone forward on one context gives these bits)
    out = {"v_cpu": O.forward_flow(W, cfg, x0, ts, li, ocache, od(octext, ct), od(ocimg, cim), 4.0, 2.0, 0.0, "text_channel", parts=parts).float()}
    out.update(o_c=parts["v_cond"].float(), o_t=parts["v_cfg_text"].float(), o_i=parts["v_cfg_img"].float())
    tt["denoise_step_3_forwards"] = time.time() - t1
    out["kv"] = {n: [(c.key_cache[i], c.value_cache[i]) for i in range(L)] for n, c in (("cond", ocache), ("cfg_text", octext), ("cfg_img", ocimg))}
    out.update(cpu_seconds=time.time() - t0, cpu_seconds_by_phase=tt, threads=threads, weights=weight_fingerprint(W), layers=L)
    return out


def edit_depth_step(args, cfg, model, ids, threads, ctx_tokens=(576, 30), vae=None, oracle_out=None):
    """The THREE-forward Euler step of an image-edit request (app.py:224-228 defaults; bagel.py:854-905): cond forward on [image context | prompt], CFG-text
    forward on [image context] (a prefix of the cond context, as ``copy.deepcopy(gen_context)`` before the prompt makes it, inferencer.py:230-253), CFG-img
    forward on [prompt] alone, cfg_text_scale 4.0, cfg_img_scale 2.0, ``text_channel`` renorm -- through the oracle on this box's host cores WITH THE GPU MODEL'S
    OWN WEIGHTS and through the HIP engine (sequential ``_forward_flow`` and the stream-batched three-forward batch inside ``generate_image``).
    ``vae=None``: the image context is stood in for by ``ctx_tokens[0]`` text tokens (what is under test is the denoise step on three different contexts).
    ``vae=<AutoEncoder>``: the REAL request chain at its own context lengths (round-5 verdict: VAE-encode -> gen-mode prefill -> SigLIP -> und-mode prefill -> prompt
    => 9 032 / 9 000 / 32-key contexts at 1024^2), see ``edit_depth_inputs``.  ``oracle_out``: the result of ``edit_depth_oracle`` computed elsewhere (a worker
    process of this bench run, ``--oracle-job``, on a model with the same name-seeded weights -- checked by fingerprint); None = computed here.
    Gates like ``full_depth_step``: every single forward -- the three sequential ones AND the three per-stream velocities of the stream-batched forward, the timed
    path -- within FULL_DEPTH_TOL_FORWARD of the oracle, the per-layer K / V of the three contexts within EDIT_CONTEXT_TOL_KV, and the combine step on the product's
    own three velocities within FULL_DEPTH_TOL_COMBINE.  Checker use of the oracle only."""
    import copy
    from oracle import bagel_oracle as O
    from bagel_amd.modeling.bagel.qwen2_navit import NaiveCache
    L = model.config.llm_config.num_hidden_layers
    real = vae is not None
    cfg = dict(cfg, llm=dict(cfg["llm"], num_hidden_layers=L))          # (a depth-reduced model in the tests: the oracle walks the layers the model has)
    inp = edit_depth_inputs(args, cfg, model, ids, ctx_tokens, real)
    keep = EDIT_KEEP_REAL if real else EDIT_KEEP
    if oracle_out is None:
        W = {k: v.detach().to("cpu") for k, v in model.state_dict().items() if k.startswith(keep)}
        VW = {k: v.detach().float().cpu() for k, v in vae.state_dict().items()} if real else None
        ora = edit_depth_oracle(cfg, W, VW, inp, threads)
        del W, VW
    else:
        ora = oracle_out
        check_fingerprint(model, ora, L, "edit_depth_step")
    # ---- product: the same chain
    if not real:
        cache = model.forward_cache_update_text(NaiveCache(L), **inp["image_text"])
    else:
        enc_noise = inp["enc_noise"]

        class _FixedNoiseVae:      # the reference draws randn_like inside encode; both sides get the same seeded CPU draw
            def encode(self, x):
                return vae.encode(x.to(model.device), sample_noise=enc_noise, precision="bf16")
        cache = model.forward_cache_update_vae(_FixedNoiseVae(), NaiveCache(L), **inp["vae_image"])
        cache = model.forward_cache_update_vit(cache, **inp["vit_image"])
    cfg_text_cache = copy.deepcopy(cache)
    cache = model.forward_cache_update_text(cache, **inp["prompt"])
    cfg_img_cache = model.forward_cache_update_text(NaiveCache(L), **inp["prompt_alone"])
    (l2, r2), (l1, r1), (l3, r3) = inp["cond"], inp["text"], inp["img"]
    relk = lambda a, b: float((a.float().cpu().reshape(b.shape) - b.float()).norm() / b.float().norm())  # noqa: E731
    kv_err = {n: max(max(relk(c.key_cache[i], ora["kv"][n][i][0]), relk(c.value_cache[i], ora["kv"][n][i][1])) for i in range(L))
              for n, c in (("cond", cache), ("cfg_text", cfg_text_cache), ("cfg_img", cfg_img_cache))}
    li, ct, cim = inp["latent"], inp["cfg_text"], inp["cfg_img"]
    x0 = li["packed_init_noises"]
    ts = torch.tensor([1.0] * x0.shape[0])
    v_cpu, o_c, o_t, o_i = ora["v_cpu"], ora["o_c"], ora["o_t"], ora["o_i"]
    lis = lambda d: dict(li, packed_position_ids=d["cfg_packed_position_ids"], packed_indexes=d["cfg_packed_query_indexes"],  # noqa: E731
                         key_values_lens=d["cfg_key_values_lens"], packed_key_value_indexes=d["cfg_packed_key_value_indexes"])
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
    got = {}
    model.velocity_hook = lambda batched, vs: got.update(batched=batched, vs=[None if v is None else v.float().cpu() for v in vs])  # noqa: E731
    try:
        lat = model.generate_image(past_key_values=cache, num_timesteps=2, cfg_text_scale=4.0, cfg_img_scale=2.0, cfg_interval=[0, 1.0], cfg_renorm_min=0.0,
                                   cfg_renorm_type="text_channel", timestep_shift=3.0, **kw, **li)
    finally:
        model.velocity_hook = None
    v_gpu = x0 - torch.cat([t.float().cpu() for t in lat])
    v_self = O.cfg_combine(v_c.cpu(), v_t.cpu(), v_i.cpu(), 4.0, 2.0, 0.0, "text_channel").float()
    out = {"what": f"one Euler step (t = 1) of an image-edit request: cond / CFG-text / CFG-img forwards of {L} MoT layers over {x0.shape[0] + 2} tokens on "
                   f"{int(l2[0])} / {int(l1[0])} / {int(l3[0])}-token contexts, CFG 4.0 / 2.0, text_channel renorm, 7B shapes, identical weights and inputs",
           "contexts_from": inp["chain"], "layers": L, "contexts": [int(l2[0]), int(l1[0]), int(l3[0])], "cpu_seconds": ora["cpu_seconds"],
           "cpu_seconds_by_phase": ora["cpu_seconds_by_phase"], "threads": ora["threads"],
           "oracle_ran": "in this process" if oracle_out is None else "in a worker process of this run, beside the GPU legs (same name-seeded weights: fingerprint checked)",
           "context_kv_rel_l2_max": kv_err,
           "rel_l2_cond_forward": rel(v_c, o_c), "rel_l2_cfg_text_forward": rel(v_t, o_t), "rel_l2_cfg_img_forward": rel(v_i, o_i),
           "rel_l2_combined_sequential": rel(v_seq, v_cpu), "rel_l2_combined_stream_batched": float((v_gpu - v_cpu).norm() / v_cpu.norm()),
           "cfg_combine_self_consistency": {"sequential_forward_flow": rel(v_seq, v_self), "generate_image_stream_batched": float((v_gpu - v_self).norm() / v_self.norm())},
           "stream_batched": {"ran_batched": bool(got.get("batched")), "rel_l2_cond_forward": float((got["vs"][0] - o_c).norm() / o_c.norm()),
                              "rel_l2_cfg_text_forward": float((got["vs"][1] - o_t).norm() / o_t.norm()), "rel_l2_cfg_img_forward": float((got["vs"][2] - o_i).norm() / o_i.norm()),
                              "what": "per-stream velocities of the ONE three-stream forward inside generate_image (the timed path), before the combine, vs the oracle's"},
           "tolerance_forward": FULL_DEPTH_TOL_FORWARD, "tolerance_combine": FULL_DEPTH_TOL_COMBINE, "tolerance_stream_batched": EDIT_DEPTH_TOL_BATCHED,
           "tolerance_context_kv": EDIT_CONTEXT_TOL_KV}
    sc, sb = out["cfg_combine_self_consistency"], out["stream_batched"]
    out["within_tolerance"] = bool(max(out["rel_l2_cond_forward"], out["rel_l2_cfg_text_forward"], out["rel_l2_cfg_img_forward"]) <= FULL_DEPTH_TOL_FORWARD
                                   and sb["ran_batched"] and max(sb["rel_l2_cond_forward"], sb["rel_l2_cfg_text_forward"], sb["rel_l2_cfg_img_forward"]) <= FULL_DEPTH_TOL_FORWARD
                                   and max(kv_err.values()) <= EDIT_CONTEXT_TOL_KV
                                   and sc["sequential_forward_flow"] <= FULL_DEPTH_TOL_COMBINE and sc["generate_image_stream_batched"] <= EDIT_DEPTH_TOL_BATCHED)
    return out


PORT_OVER_REFERENCE = "1.03 +- 0.05"          # seconds per layer-forward, oracle port / unmodified reference classes, measured in the build container (round 6)
PORT_OVER_REFERENCE_LOG = os.path.join(ROOT, "profiles", "r06_cpu_baseline_build_container.log")


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
            full = over_budget(args, 240, "cpu_baseline.parity_at_full_depth (one Euler step of the 28-layer model through the oracle)")
            if full is None:
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
               value_from=("one MEASURED Euler step at full depth (cond + CFG forward of all layers, B = 1) x Euler steps" if full and "error" not in full and "skipped" not in full
                           else "one measured layer-forward x layers x 2 forwards x Euler steps"),
               sample=f"{'unmodified reference (generate_image, 1 Euler step = 2 forwards' if kind == 'reference' else 'oracle MoT decoder layer (gen mode'}, "
                      f"{Lq} query tokens on a {C}-token context, 7B shapes, {nl} layer(s)), after one warm-up pass: {dt:.2f} s per layer-forward on "
                      f"{threads} threads = {fl / dt / 1e12:.2f} TFLOP/s; extrapolated x{llm['num_hidden_layers']} layers x2 forwards x{steps} Euler "
                      f"steps (glue, prefill and VAE excluded)"
                      + (f"; and ONE MEASURED Euler step at full depth through the oracle with the GPU model's weights (cond + CFG-text forward of "
                         f"{full['layers']} layers, B = 1): {full['cpu_seconds_per_euler_step']:.1f} s -> value = 1 / ({steps} x that)"
                         if full and "error" not in full and "skipped" not in full else "")
                      + ("" if kind == "reference" else f"; port / unmodified-reference time on the same cores = {PORT_OVER_REFERENCE} "
                         f"({os.path.relpath(PORT_OVER_REFERENCE_LOG, ROOT)}: three alternating pairs of this same one-layer sample in the build container, where "
                         "/root/reference exists)"))
    if kind != "reference":
        out["port_over_reference_time_ratio"] = {"value": 1.03, "spread": 0.05, "source": os.path.relpath(PORT_OVER_REFERENCE_LOG, ROOT)}
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


UND_KEEP = ("language_model.", "vit_model.", "connector.", "vit_pos_embed.")


def und_request(args):
    """The synthetic understanding request of configs[1]: a seeded U(-1, 1) image and 32 random prompt ids."""
    g = torch.Generator().manual_seed(2)
    image = torch.rand(3, args.und_image, args.und_image, generator=g) * 2 - 1
    prompt_ids = torch.randint(0, 151643, (32,), generator=torch.Generator().manual_seed(1)).tolist()
    return image, FixedTokenizer(prompt_ids)


def und_depth_inputs(args, model, ids):
    """PHASE 1 (host only): the packer outputs of the understanding request (ViT image block, prompt, start tokens)."""
    image, tok = und_request(args)
    ident = lambda t: t  # noqa: E731
    ti, l1, r1 = model.prepare_vit_images([0], [0], [image], ident, ids)
    pi, l2, r2 = model.prepare_prompts(l1, r1, ["p"], tok, ids)
    st = model.prepare_start_tokens(l2, r2, ids)
    return _cpu_tree(dict(vit_image=ti, prompt=pi, start=st, context_tokens=int(l2[0])))


def und_depth_oracle(cfg, W, inp, threads, n_tokens=6):
    """PHASE 2 (CPU only, checker): SigLIP + connector + the non-causal prefill of the ViT block + the causal text prefill + ``n_tokens`` greedy steps through the
    oracle; may run in a worker process beside the GPU legs (``--oracle-job``)."""
    from oracle import bagel_oracle as O
    torch.set_num_threads(threads)
    L = cfg["llm"]["num_hidden_layers"]
    t1 = time.time()
    ocache = O.forward_cache_update_vit(W, cfg, O.OracleCache(L), **inp["vit_image"])
    t_vit = time.time() - t1
    ocache = O.forward_cache_update_text(W, cfg, ocache, **inp["prompt"])
    t_prefill = time.time() - t1
    okv = [(ocache.key_cache[i].clone(), ocache.value_cache[i].clone()) for i in range(L)]
    t1 = time.time()
    st = inp["start"]
    otoks, ologits = O.generate_text(W, cfg, ocache, st["packed_key_value_indexes"], st["key_values_lens"], st["packed_start_tokens"],
                                     st["packed_query_position_ids"], n_tokens, return_logits=True)
    t_decode = time.time() - t1
    return dict(kv=okv, tokens=otoks, logits=[x.clone() for x in ologits], n_tokens=n_tokens, layers=L, threads=threads, weights=weight_fingerprint(W),
                cpu_seconds={"vit_prefill": t_vit, "vit_plus_text_prefill": t_prefill, "decode": t_decode})


def understanding_full_depth(args, cfg, model, ids, threads, n_tokens=6, oracle_out=None):
    """configs[1] at the DEPTH it is measured at (VERDICT r04 missing-2): the 26-layer SigLIP encoder + connector + the 28-layer non-causal prefill of the
    ViT block + the causal text prefill + ``n_tokens`` greedy decode steps through the oracle on this box's host cores WITH THE GPU MODEL'S OWN WEIGHTS, against
    the product on the same inputs: per-layer K / V of the whole context (max rel-L2), the first decode step's logits (rel-L2), and the greedy ids up to the
    first reference near-tie (the rule of tests/test_und_shapes_gpu.py: a differing id must sit within 2^-6 max|logit| of the reference's top-1).
    ``oracle_out``: ``und_depth_oracle``'s result from a worker process of this run (same name-seeded weights, fingerprint checked); None = computed here.
    Reference: bagel.py:362-415 (forward_cache_update_vit), :321-360 (text), :930-1000 (generate_text); siglip_navit.py:389-402.  Checker use of the oracle only."""
    import copy
    from bagel_amd.modeling.bagel.qwen2_navit import NaiveCache
    L = model.config.llm_config.num_hidden_layers
    inp = und_depth_inputs(args, model, ids)
    t_copy = None
    if oracle_out is None:
        t0 = time.time()
        W = {k: v.detach().to("cpu") for k, v in model.state_dict().items() if k.startswith(UND_KEEP)}
        t_copy = time.time() - t0
        ora = und_depth_oracle(dict(cfg, llm=dict(cfg["llm"], num_hidden_layers=L)), W, inp, threads, n_tokens)
        del W
    else:
        ora = oracle_out
        n_tokens = ora["n_tokens"]
        check_fingerprint(model, ora, L, "understanding_full_depth")
    okv, otoks, ologits = ora["kv"], ora["tokens"], ora["logits"]
    ti, pi, st = inp["vit_image"], inp["prompt"], inp["start"]
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
    out = {"what": f"{cfg['vit']['num_hidden_layers']}-layer SigLIP + connector + {L}-layer prefill of a {inp['context_tokens']}-token context + {n_tokens} greedy decode steps, 7B shapes, "
                   "identical weights and inputs: HIP engines vs oracle",
           "layers": L, "context_tokens": inp["context_tokens"], "kv_rel_l2_max": max(ek + ev), "k_rel_l2_by_layer": [round(e, 5) for e in ek],
           "v_rel_l2_by_layer": [round(e, 5) for e in ev], "first_step_logits_rel_l2": e_logits,
           "first_step_top1_margin_over_max_logit": float((lg0.max() - lg0.topk(2).values[1]) / lg0.abs().max()),
           "greedy_ids_agree_until_step": agree + 1 if tie is None else tie["step"], "greedy_steps_compared": n_tokens - 1, "first_mismatch": tie,
           "tokens_gpu": [int(x) for x in toks[:, 0]], "tokens_oracle": [int(x) for x in otoks[:, 0]],
           "cpu_seconds": dict(ora["cpu_seconds"], weights_copy=t_copy), "threads": ora["threads"],
           "oracle_ran": "in this process" if oracle_out is None else "in a worker process of this run, beside the GPU legs (same name-seeded weights: fingerprint checked)",
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
    image, tok = und_request(args)
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
    n = args.und_new_tokens
    decode(cache, lens, ropes, n)              # the timed decode's own length once (session buffers, page pools and the merged cache it leaves behind)
    cache, lens, ropes, _ = prefill()
    fence()
    # the prefill is an 85 ms region launched from Python: ONE sample of it read 97.9 ms on a box whose other samples were 82-84 (round 6, visit 16; steady
    # state, no allocation) -- three samples, the median is reported and all three are in the line
    pre_samples = []
    with Steady(dev) as mem_prefill:
        for _ in range(3):
            t0 = time.perf_counter()
            cache, lens, ropes, t_vit = prefill()
            t1 = time.perf_counter()
            fence()
            pre_samples.append(((t_vit - t0) * 1e3, (t1 - t_vit) * 1e3))
    pre_med = sorted(pre_samples)[1]
    with Steady(dev) as mem_decode:
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
            with Steady(dev) as m8:
                t4 = time.perf_counter()
                model.generate_text(past_key_values=c8, max_length=n, do_sample=False, end_token_id=None, weight_quant=mode, **s8)
                fence()
                dt8 = time.perf_counter() - t4
            return {"value": n / dt8, "unit": "tokens/s", "decode_ms_per_token": dt8 / n * 1e3, "weights": weights, "peak_mem_gb": m8.report["peak_mem_gb"],
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
            with Steady(dev):
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
    def batched_decode(nb, nn=160):      # (step 0 runs eagerly and the hipGraph capture follows it: amortised over the run)
        try:
            cb, lb, rb, _ = prefill(nb)
            sb = model.prepare_start_tokens(lb, rb, ids)
            # warm-up with the SAME length: the call re-allocates the merged caches on the way out (nb x (4936 + 160) rows x 28 layers), and a warm-up
            # of another length left the timed call ~80 ms of first-time hipMalloc (0.5 ms per step of 160) -- a serving process is past that
            model.generate_text(past_key_values=cb, max_length=nn, do_sample=False, end_token_id=None, **sb)
            cb, lb, rb, _ = prefill(nb)
            sb = model.prepare_start_tokens(lb, rb, ids)
            fence()
            with Steady(dev) as mem_bd:
                t4 = time.perf_counter()
                tb = model.generate_text(past_key_values=cb, max_length=nn, do_sample=False, end_token_id=None, **sb)
                fence()
                dtb = time.perf_counter() - t4
            # algorithmic bytes of a step: ONE pass over the und expert + lm_head serves the batch, every request reads its own KV (average context of the span)
            H_, I_, nkv_, V_ = llm["hidden_size"], llm["intermediate_size"], llm["num_key_value_heads"], llm["vocab_size"]
            hd_ = H_ // llm["num_attention_heads"]
            b_step = 2.0 * (L * (2 * H_ * H_ + 2 * H_ * nkv_ * hd_ + 3 * H_ * I_) + V_ * H_) + nb * 2.0 * nkv_ * hd_ * 2 * L * (int(lb[0]) + nn / 2.0)
            out = {"value": nb * nn / dtb, "unit": "tokens/s", "batch": nb, "new_tokens": nn, "decode_ms_per_step": dtb / nn * 1e3,
                   "roofline": {"bound": "hbm", "achieved": b_step / (dtb / nn) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": b_step / (dtb / nn) / 1e9 / HBM_PEAK_GBS,
                                "algorithmic_bytes_per_step": b_step},
                   "context_tokens": int(lb[0]), "outputs_ok": bool(tb.shape == (nn, nb)),
                   # the nb requests are the SAME request (one image, one prompt): every column must carry the same ids, and they must be the batch-1 run's ids up to
                   # the first near-tie flip (other kernels, other fp32 summation orders; tests/test_wide_gpu.py::test_wide_model_batched_decode_is_batch_invariant)
                   "all_requests_agree": bool((tb == tb[:, :1]).all()),
                   "ids_equal_to_the_batch1_run_for_steps": int(next((i for i in range(min(nn, int(toks.shape[0]))) if int(tb[i, 0]) != int(toks[i, 0])), min(nn, int(toks.shape[0])))),
                   "memory": mem_bd.report,
                   "note": "batched multi-request decode (SURVEY 8f.4b): beside the batch-1 headline, never as it"}
            unsteady(out, dict(mem_bd.report, attempts=1))
            del cb
            return out
        except Exception as e:
            return {"error": repr(e)}
    bd = bd32 = None
    if UB == 1 and not args.no_batched_decode:
        bd = batched_decode(16)
        # round 6: two blocks of 16 request rows share every weight fragment (csrc/gemv_mb.hip MB = 2): 32 requests per weight pass
        if torch.cuda.is_available():
            torch.cuda.empty_cache()         # (this leg peaks at 96 GB allocated / 122 GB reserved; the parent bench process lives on the same GPU)
        bd32 = batched_decode(32)
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
            ora = wait_oracle_job(args.und_oracle_out, "und_depth", timeout=max(60.0, args.wall_budget_s - (time.time() - T_START))) if args.und_oracle_out else None
            skip = None if ora is not None else over_budget(args, 230, "understanding.parity_at_full_depth with the oracle in this process")
            if skip is None:
                depth = understanding_full_depth(args, cfg, model, ids, physical_cores(), oracle_out=ora)
                if ora is not None:
                    depth["worker"] = dict(ora.get("worker", {}), child_waited_s=ora.get("waited_s"))
            else:
                depth = skip
        except Exception as e:
            import traceback
            depth = {"error": repr(e), "trace": traceback.format_exc()[-1500:]}
    w_bytes = 2.0 * (L * (2 * H * H + 2 * H * nkv * hd + 3 * H * I) + V * H)
    ctx = lens[0]
    kv_bytes = 2.0 * nkv * hd * 2 * L * (ctx + n / 2.0)          # average context over the decoded span
    bpt = w_bytes + UB * kv_bytes            # one weight pass serves the whole batch; every request reads its own KV
    tps = UB * n / dt
    return {"metric": "understanding tokens/sec", "value": world * tps, "unit": "tokens/s", "per_gpu_tokens_per_s": tps,
            "new_tokens": int(toks.shape[0]), "batch_per_gpu": UB, "context_tokens": int(ctx),
            "prefill_ms": {"vit_encoder_plus_llm_prefill": pre_med[0], "text_prefill": pre_med[1], "samples": pre_samples, "reported": "median of 3"},
            "decode_ms_per_step": dt / n * 1e3, "decode_ms_per_token": dt / n / UB * 1e3, "hip_graph": sess.graph is not None, "hip_graph_error": sess.graph_error,
            "kv_cache": f"paged, {sess.paged.PAGE}-token pages, {sess.paged.num_pages} pages/layer", "cpu_baseline": cpu,
            "memory": {"prefill_timed_region": mem_prefill.report, "decode_timed_region": mem_decode.report, "resident": resident_weight_bytes(model),
                       "what": f"a process that holds the model (no VAE) and serves understanding requests at batch {UB}: peak of live tensor bytes inside each timed region"},
            "steady_state": bool(mem_prefill.report["steady"] and mem_decode.report["steady"]),
            "int8_rowwise_weights": w8, "mxfp4_weights": w4, "nf4_weights": wn, "batched_decode": bd, "batched_decode_32": bd32, "sampled_decode": sampled, "parity_at_full_depth": depth,
            "roofline": {"bound": "hbm", "achieved": bpt * (tps / UB) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": bpt * (tps / UB) / 1e9 / HBM_PEAK_GBS, "traffic": pmc_decode_traffic() if (UB == 1 and args.und_image == 980) else None,
                         "kernel": "gemv_kernel (decode step)",
                         "algorithmic_bytes_per_step": bpt},
            "workload": f"BAGEL-7B-MoT image understanding: {args.und_image}x{args.und_image} image -> {(args.und_image // 14) ** 2} ViT tokens (+2 markers) + 32+2 "
                        f"prompt tokens prefill, greedy decode of {n} tokens, bf16, batch {UB}/GPU"}


PMC_SUMMARY = os.path.join(ROOT, "profiles", "r06_pmc_summary.json")


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
    out = {"kernel": "attn2_kernel<128, false> (+ attn2_combine_kernel<128>)", "bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
           "frac": ach / PEAK_BF16_TFLOPS, "launches": len(big), "avg_launch_ms": ms / len(big), "samples_per_launch": big[0][3]}
    d = _pmc_entry("attention", None, ["attention2.hip", "common.h"]) if (args.workload == "t2i" and args.batch == 4 and R == 1024) else None
    ks = (d or {}).get("kernels", {})
    k = ks.get("attn2_kernel<128, false>") or ks.get("attn2_kernel<128>")          # (round 6: a second template argument; "false" = the unified ring the product launches)
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


def understanding_subprocess(args, local, job=None):
    """Run the configs[1] leg in a child process on the same GPU (a replica per rank): a fault or hang there can never
    take the text->image number down with it."""
    import subprocess
    env = dict(os.environ)
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK=str(local), LOCAL_WORLD_SIZE="1", BAGEL_PAUSE_PIDS=",".join(str(x) for x in Steady.pause_pids))
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--only-understanding"] + (["--no-cpu-baseline"] if args.no_cpu_baseline or int(os.environ.get("WORLD_SIZE", 1)) != 1 else []) + [
           "--und-new-tokens", str(args.und_new_tokens), "--und-image", str(args.und_image), "--und-batch", str(args.und_batch)] + (
           ["--no-int8"] if args.no_int8 else []) + (["--no-batched-decode"] if args.no_batched_decode else []) + (["--no-full-depth"] if args.no_full_depth else [])
    if args.layers is not None:
        cmd += ["--layers", str(args.layers)]
    if job is not None:
        cmd += ["--und-oracle-out", job["dir"]]
    cmd += ["--wall-budget-s", str(args.wall_budget_s - (time.time() - T_START) - 250.0)]      # what is left of this run's budget, less the parent's own full-depth leg
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
    if args.oracle_job:
        return oracle_job_main(args)
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

    # the CPU-oracle sides of the two parity legs that do not time the CPU start NOW, in worker processes, and run beside the GPU-bound headline loop
    jobs = {}
    if (cuda and world == 1 and not args.standins and not args.no_oracle_jobs and not args.no_cpu_baseline and not args.no_full_depth and args.workload == "t2i"
            and not args.only_understanding and args.layers is None and not args.weight_store):
        jobs = start_oracle_jobs(args, local, ([] if args.no_understanding or args.und_batch != 1 else ["und_depth"]) + ([] if args.no_edit else ["edit_depth"]))
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

    def one_step(taylorseer=False, B=B, T=T):
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
        li["packed_init_noises"] = my_noise[: B * n_img]
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

    def edit_request(i):
        """Inputs of edit request ``i`` of this rank (seeded; request 0 is the one the one-request latency is quoted on)."""
        if i not in edit_state:
            gsrc = torch.Generator().manual_seed(3 + 100 * i)
            edit_state[i] = dict(src_vae=(torch.rand(3, R, R, generator=gsrc) * 2 - 1).to(dev),        # vae_transform(image) stand-in
                                 src_vit=(torch.rand(3, 980, 980, generator=gsrc) * 2 - 1).to(dev),    # vit_transform(image) stand-in
                                 enc_noise=torch.randn(1, 16, R // 8, R // 8, generator=torch.Generator().manual_seed(43 + 100 * i)),
                                 noise=(all_noise[rank * n_img:(rank + 1) * n_img] if i == 0 else
                                        torch.randn(n_img, pdim, generator=torch.Generator().manual_seed(4242 + 100 * i + rank))).to(dev))
        return edit_state[i]

    def edit_step(taylorseer=False, timesteps=None, reqs=(0,)):
        """``reqs``: the edit requests served TOGETHER as one NaViT batch (the packers and the engines take batches; every sample keeps its own contexts, its own
        per-token text_channel renorm and its own noise, so a request's result does not depend on its batch mates beyond fp32 summation order in split tiles)."""
        import copy
        rq = [edit_request(i) for i in reqs]
        nb = len(rq)
        ident = lambda t: t  # noqa: E731
        enc_noise = torch.cat([r_["enc_noise"] for r_ in rq], 0)

        class _FixedNoiseVae:      # the reference draws randn_like inside encode; feed a seeded CPU draw
            def encode(self, x):       # the edit request is the app's / inferencer's path: its VAE runs inside torch.autocast(bf16) (inferencer.py:233)
                return vae.encode(x, sample_noise=enc_noise, **({} if args.standins else {"precision": "bf16"}))

        z = [0] * nb
        vi, l1, r1 = model.prepare_vae_images(z, z, [r_["src_vae"] for r_ in rq], ident, ids)
        cache = model.forward_cache_update_vae(_FixedNoiseVae(), NaiveCache(L), **vi)
        ti, l2, r2 = model.prepare_vit_images(l1, r1, [r_["src_vit"] for r_ in rq], ident, ids)
        cache = model.forward_cache_update_vit(cache, **ti)
        cfg_text_cache = copy.deepcopy(cache)
        pi, l3, r3 = model.prepare_prompts(l2, r2, ["p"] * nb, tok, ids)
        cache = model.forward_cache_update_text(cache, **pi)
        pi2, l4, r4 = model.prepare_prompts(z, z, ["p"] * nb, tok, ids)
        cimg_cache = model.forward_cache_update_text(NaiveCache(L), **pi2)
        li = model.prepare_vae_latent(l3, r3, [(R, R)] * nb, ids)
        li["packed_init_noises"] = torch.cat([r_["noise"] for r_ in rq], 0)
        ct = model.prepare_vae_latent_cfg(l2, r2, [(R, R)] * nb)
        cim = model.prepare_vae_latent_cfg(l4, r4, [(R, R)] * nb)
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
    # ---- memory, the reference's only published figure for this path ("80 GiB is sufficient": app.py:77 max_memory, README.md:139-151): ONE text->image request
    #      (B = 1, full resolution, VAE decode included; 3 timesteps reach the same peak as 50) in a process that holds nothing but the model -- before any B = 4 workspace exists
    mem_b1 = None
    if cuda and args.workload == "t2i" and not args.standins and world == 1 and args.layers is None and not args.no_memory_leg:      # (not under the debug flags of the PMC passes: its B = 1 launches would dilute their per-kernel averages)
        try:
            one_step(B=1, T=3)
            one_step(B=1, T=3)
            with Steady(dev) as m1:
                one_step(B=1, T=3)
            mem_b1 = dict(m1.report, what="bf16 model (parameters + packed copies" + (" + VAE" if vae is not None else "") + ") + one 1024^2 text->image request, B = 1, "
                          "cond + CFG forward stream-batched, fp32 VAE decode included: peak of live tensor bytes in a process that holds nothing else",
                          fits_80gb=bool(m1.report["peak_mem_gb"] <= 80.0))
            model.language_model.engine().release_workspaces()
        except Exception as e:
            mem_b1 = {"error": repr(e)}
    for _ in range(args.warmup):
        one_step()
    jobs_wait_s = wait_gpu_released(jobs) if jobs else 0.0          # the workers' model builds overlap the warm-up; nothing else is on the GPU from here on
    records, orig_gemm = [], ops.gemm
    arecords, orig_attn = [], ops.attn_planned
    if cuda:
        records, orig_gemm, timed_gemm = gemm_profile_hook()
        ops.gemm = timed_gemm
        arecords, orig_attn, timed_attn = attn_profile_hook()
        ops.attn_planned = timed_attn
    del bcast[:]
    fence()
    with Steady(dev) as mem_head:
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
    # a timed leg that went to the driver for memory is run once more -- but only in a one-rank job: the legs contain collectives (the conditioning-KV broadcast, the
    # fence's barrier), and a retry that one rank takes and another does not would leave them waiting for each other
    tries = 2 if world == 1 else 1
    ts = None
    if not args.no_taylorseer:
        # the reference's own accelerator option (generate_image(enable_taylorseer=True), bagel.py:678-689): same workload,
        # 19 instead of 49 full backbone forwards per stream.  It CHANGES the samples, so it is reported beside the headline
        # number, never as it.
        (lat_ts, _), dt_ts, mem_ts = timed_steady(lambda: one_step(taylorseer=True), dev, fence, attempts=tries)
        if world > 1:
            tt = torch.tensor([dt_ts], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_ts = float(tt.item())
        st = model._last_taylor_states[0]
        ts = {"value": world * B / dt_ts, "unit": "images/s", "ms_per_step": dt_ts * 1e3, "full_forwards_per_stream": st.full_steps,
              "extrapolated_forwards_per_stream": st.taylor_steps, "outputs_finite": all(torch.isfinite(x).all().item() for x in lat_ts), "memory": mem_ts,
              "note": "enable_taylorseer=True (reference option, changes the samples): not the headline metric"}
        unsteady(ts, mem_ts)
    fp8 = None
    if args.workload == "t2i" and not args.no_fp8 and not args.only_understanding:
        # option model.gen_weight_quant = "fp8": the gen expert's projections on the OCP-e4m3 MFMA with row-wise scales (SURVEY.md 8f.4;
        # the MI355X counterpart of the reference's quantised load modes).  It CHANGES the samples (a few percent, tests/test_fp8_gpu.py):
        # reported beside the headline number, never as it.
        try:
            model.gen_weight_quant = "fp8"
            one_step(T=3)                               # warm-up: quantises the 28 x 4 gen-expert matrices once, allocates the fp8 workspaces
            (lat_8, _), dt_8, mem_8 = timed_steady(one_step, dev, fence, attempts=tries)
            if world > 1:
                tt = torch.tensor([dt_8], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt_8 = float(tt.item())
            dev_l2 = max(float(((a.float() - b.float()).norm() / b.float().norm()).item()) for a, b in zip(lat_8, latents))
            fp8 = {"value": world * B / dt_8, "unit": "images/s", "ms_per_step": dt_8 * 1e3, "outputs_finite": all(torch.isfinite(x).all().item() for x in lat_8),
                   "latents_rel_l2_vs_bf16_run": dev_l2, "memory": mem_8, "weights": "gen expert q/k/v/o/gate/up/down in OCP e4m3, row-wise absmax scales; activations "
                   "quantised per row on the fly; und expert, attention, norms, residual stream bf16",
                   "note": "gen_weight_quant='fp8' option (changes results): not the headline metric"}
            unsteady(fp8, mem_8)
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
            if cuda and not args.standins:
                model.language_model.engine().release_workspaces(quantised=True)      # the peak reported below is the model + THIS request's working set
            edit_step(timesteps=3)
            edit_step(timesteps=3)          # twice: the second request is set up while the first one's caches are still referenced (cf. the understanding prefill)
            (lat_e, _), dt_e, mem_e = timed_steady(edit_step, dev, fence, attempts=tries)
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
                    "outputs_finite": all(torch.isfinite(x).all().item() for x in lat_e), "memory": mem_e,
                    "vae_precision": "bf16 convolutions, fp32 GroupNorm (the VAE inside the inferencer's autocast region, inferencer.py:233), encode and decode",
                    "workload": f"BASELINE configs[4]: VAE-encode + SigLIP(980^2) + {args.prompt_tokens}+2 prompt tokens, {T} timesteps x [cond + CFG-text 4.0 + "
                                f"CFG-img 2.0], text_channel renorm, VAE decode included, 1 request/GPU"}
            unsteady(edit, mem_e)
            # THROUGHPUT form, beside the one-request latency and never as it: TWO requests per GPU in one NaViT batch (96 row tiles instead of 48: the
            # o / down projections run 5.25 rounds of the 256 persistent workgroups instead of 2.63, qkv 6.75 instead of 3.375 -- the partial last rounds that
            # idle 12-16 % of the chip at one request).  The batch driver of the reference serves one image per rank at a time (gen_images_mp_imgedit.py:278-303).
            def edit_batched(NB, check):
                breqs = tuple(range(NB))

                def same_as_single(nt=3):
                    """Each sample of the batch against its own single-request run (``nt`` timesteps).  The launches of the two runs cut their work differently
                    (another row count: other GEMM tiles are K-split, the attention planner splits other items along the key axis), so a sample's arithmetic
                    differs in fp32 summation ORDER only: its three per-stream velocities of the FIRST Euler step -- before any CFG amplification -- are compared
                    (a single forward's accumulation-order noise is ~1e-2 at this depth, bench.FULL_DEPTH_TOL_FORWARD = 2.4e-2), and the latents after ``nt - 1``
                    steps, where CFG 4.0 x 2.0 has amplified that noise like it does between any two execution orders of the same request."""
                    got = {}

                    def hook(batched, vs):
                        if "vs" not in got:
                            got["vs"] = [None if v is None else v.float().clone() for v in vs]
                    res = []
                    try:
                        model.velocity_hook = hook
                        both, _ = edit_step(timesteps=nt, reqs=breqs)
                        vb = got.pop("vs")
                        for i in breqs:
                            one, _ = edit_step(timesteps=nt, reqs=(i,))
                            v1 = got.pop("vs")
                            a, b_ = both[i].float(), one[0].float()
                            n1 = v1[0].shape[0]
                            res.append({"first_step_velocity_rel_l2_per_stream": [float((vb[s_][i * n1:(i + 1) * n1] - v1[s_]).norm() / v1[s_].norm()) for s_ in range(3)],
                                        "latents_bit_identical": bool(torch.equal(a, b_)), "latents_rel_l2": float((a - b_).norm() / b_.norm())})
                    finally:
                        model.velocity_hook = None
                    return res
                same = same_as_single() if check else None
                edit_step(timesteps=3, reqs=breqs)
                edit_step(timesteps=3, reqs=breqs)
                (lat_b, _), dt_b, mem_b = timed_steady(lambda: edit_step(reqs=breqs), dev, fence, attempts=tries)
                if world > 1:
                    tt = torch.tensor([dt_b], dtype=torch.float64, device=dev)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    dt_b = float(tt.item())
                return unsteady({"value": NB * world / dt_b, "unit": "images/s", "requests_per_gpu": NB, "seconds_per_batch": dt_b, "seconds_per_image": dt_b / NB,
                                 "speedup_over_one_request_per_gpu": (NB / dt_b) / (1 / dt_e), "outputs_finite": all(torch.isfinite(x).all().item() for x in lat_b),
                                 "each_sample_vs_its_single_request_run": same, "memory": mem_b,
                                 "whole_path_roofline_frac": NB * pf / dt_b / 1e12 / PEAK_BF16_TFLOPS,
                                 "note": f"{NB} independent edit requests (own images, contexts, noise) served as one NaViT batch: THROUGHPUT, beside the "
                                         "one-request latency above, never as it"}, mem_b)
            for bi, NB in enumerate(args.edit_batch):
                key = "batched" if bi == 0 else f"batched_{NB}"
                try:
                    edit[key] = edit_batched(NB, check=bi == 0)
                except Exception as e:
                    import traceback
                    edit[key] = {"error": repr(e), "trace": traceback.format_exc()[-1200:]}
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
            with Steady(dev) as mem_tf:
                t1 = time.perf_counter()
                for _ in range(3):
                    o_ = model(noise=tn, **tb)
                fence()
                dtt = (time.perf_counter() - t1) / 3
            ntok = tb["sequence_length"]
            trainf = {"value": world * ntok / dtt, "unit": "tokens/s", "tokens_per_forward": ntok, "ms_per_forward": dtt * 1e3,
                      "linear_tflops": 13.0506e-3 * ntok / dtt, "outputs_finite": bool(torch.isfinite(o_["ce"]).all() and torch.isfinite(o_["mse"]).all()), "memory": mem_tf.report,
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
                    with Steady(dev):                    # (the step's own peak is read below from max_memory_allocated: this bracket resets the counter and stops the workers)
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
                    # what a training loop keeps RESIDENT between its steps -- the tape-buffer pool (35-75 GB at 7B) and the transposed weight images of
                    # dX = dY W (2 bytes per parameter) -- goes back to the device before the understanding child starts on the same GPU: with it in
                    # place the child's 32-request decode leg (96 GB peak) met 156 MB of free memory (visit 15 of round 6)
                    lm_ = model.language_model
                    if hasattr(lm_.model, "release_train_buffers"):
                        lm_.model.release_train_buffers()
                    eng_ = getattr(lm_, "_engine", None)
                    for P_ in (getattr(eng_, "layers", None) or []):
                        if hasattr(P_, "wt"):
                            P_.wt = {}
                    if cuda:
                        torch.cuda.empty_cache()
            del tb, tn, o_
        except Exception as e:
            import traceback
            trainf = {"error": repr(e), "trace": traceback.format_exc()[-1200:]}
    und = None
    if not args.no_understanding:
        # the understanding child has SHORT timed regions (an 85 ms prefill): it starts once the workers are gone -- at the driver's flags that is a wait of ~25 s.
        # (Measured, round 6: with two workers alive -- stopped during the child's timed regions like during the parent's -- the child's prefill read 177 ms instead
        # of 84; the parent's own legs, seconds long each, were unaffected.  Not waiting is not worth finding out why.)
        jobs_wait_end_s = 0.0
        if jobs:
            t_w = time.time()
            for j in jobs.values():
                try:
                    j["proc"].wait(timeout=max(30.0, args.wall_budget_s - (time.time() - T_START) - 500.0))
                except Exception:
                    pass
            jobs_wait_end_s = time.time() - t_w
        und = understanding_subprocess(args, local, jobs.get("und_depth"))
        if world > 1:   # replicas: aggregate tokens/s = sum over ranks
            v = und.get("per_gpu_tokens_per_s", 0.0) if isinstance(und, dict) else 0.0
            tt = torch.tensor([v], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.SUM)
            if isinstance(und, dict) and "value" in und:
                und["value"] = float(tt.item())
                und["replicas"] = world

    if (isinstance(edit, dict) and cuda and rank == 0 and world == 1 and not args.standins and not args.no_cpu_baseline and not args.no_full_depth
            and args.layers is None):   # (after the understanding child: its start-up is untimed time in which the worker advances)
        # configs[4] against the oracle AT THE CONTEXT LENGTHS IT IS TIMED AT, on a depth-reduced model (the oracle's 9 032-token chain at depth 28 is ~10 min of
        # host time): the real request chain, the three contexts' K / V, every single forward incl. the stream-batched ones (edit_depth_step)
        try:
            job = jobs.get("edit_depth")
            ora = wait_oracle_job(job["dir"], "edit_depth", proc=job["proc"]) if job else None
            skip = None if ora is not None else over_budget(args, 330 + 240, "edit.parity_at_depth with the oracle in this process")
            if skip is None:
                m4, _ = build_bagel(cfg, device=dev, num_layers=EDIT_PARITY_LAYERS, with_vae=False)
                init_random_(m4, seed=0)
                m4.llm2vae.weight.data.normal_(0, cfg["llm"]["hidden_size"] ** -0.5, generator=torch.Generator(device=dev).manual_seed(1))
                par = edit_depth_step(args, cfg, m4, ids, physical_cores(), vae=vae, oracle_out=ora)
                if ora is not None:
                    par["worker"] = dict(ora.get("worker", {}), parent_waited_s=ora.get("waited_s"))
                del m4
                torch.cuda.empty_cache()
            else:
                par = skip
        except Exception as e:
            import traceback
            par = {"error": repr(e), "trace": traceback.format_exc()[-1500:]}
        if isinstance(edit, dict):
            edit["parity_at_depth"] = par

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
            "memory": {"resident": resident_weight_bytes(model, vae) if cuda and not args.standins else None,
                       "t2i_b1_request": mem_b1,
                       "headline_timed_region": dict(mem_head.report, what=f"the {args.steps} timed steps at B = {B}/GPU (fp32 VAE decode included)"),
                       "note": "peak_mem_gb = torch.cuda.max_memory_allocated after reset_peak_memory_stats (live tensor bytes: weights + packed copies + the leg's "
                               "working set); every leg's own figure sits in its object (edit.memory, understanding.memory, taylorseer.memory, fp8_gen_expert.memory, "
                               "training_forward.training_step.peak_mem_gb); steady = no hipMalloc / hipFree inside the timed region"},
            "oracle_workers": None if not jobs else {
                "what": "the CPU-oracle sides of understanding.parity_at_full_depth and edit.parity_at_depth ran in worker processes beside the GPU legs (bench.py "
                        "--oracle-job: same name-seeded weights, fingerprint-checked by the compare phase); the parent waited for them to leave the GPU before its "
                        "timed region and STOPS them (SIGSTOP / SIGCONT) for the duration of every timed region of every leg", "kinds": sorted(jobs), "threads_each": next(iter(jobs.values()))["threads"], "parent_waited_for_gpu_release_s": jobs_wait_s,
                "parent_waited_for_the_workers_to_end_s": jobs_wait_end_s if not args.no_understanding else None,
                "gpu_released_before_timed_region": all(j.get("gpu_released") for j in jobs.values())},
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
                for j in jobs.values():                    # the CPU baseline is TIMED: no worker may still be on the host cores
                    try:
                        j["proc"].wait(timeout=600)
                    except Exception:
                        j["proc"].kill()
                gpu = None if (args.layers is not None or args.workload != "t2i") else dict(model=model, tok=tok, ids=ids)
                out["cpu_baseline"] = cpu_baseline(args, cfg, gpu)
            except Exception as e:   # the baseline is reported, never required for the GPU number
                out["cpu_baseline"] = {"error": repr(e)}
        out["wall_seconds"] = time.time() - T_START
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
