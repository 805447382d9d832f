"""Persistent decode engine vs the launch form on the projection chain of a 7B decoder layer (no attention): per-layer time of
[o(+x) -> norm + gate/up (SwiGLU) -> down(+x) -> norm + qkv] as 4 bagel_gemv_bf16 launches and as ONE bagel_decode_engine_bf16 launch, both
replayed from a hipGraph of 28 layers over NSETS distinct weight sets (cycled: > 256 MB, so nothing is Infinity-Cache resident).
Sweeps the engine's knobs (BAGEL_ENGINE_DEPTH / _NT / _WAVES / _SLOTS are read at every launch).

    python tools/decode_engine_probe.py [--sets 6] [--layers 28] [--reps 20]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bagel_amd import ops  # noqa: E402

BF16 = torch.bfloat16
DEV = "cuda"
H, I, NQKV = 3584, 18944, 4608


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sets", type=int, default=6)
    ap.add_argument("--layers", type=int, default=28)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--quick", action="store_true", help="the launch form and the default engine only (PMC passes)")
    ap.add_argument("--trace", action="store_true", help="one traced engine launch per weight set: where a layer's time goes (event times of every workgroup)")
    a = ap.parse_args()
    g = torch.Generator(device=DEV).manual_seed(0)
    rn = lambda *s, scale=1.0: (torch.randn(*s, generator=g, device=DEV) * scale).to(BF16)  # noqa: E731
    sets = [dict(wo=rn(H, H, scale=H ** -0.5), wgu=rn(2 * I, H, scale=H ** -0.5), wd=rn(H, I, scale=I ** -0.5), wqkv=rn(NQKV, H, scale=H ** -0.5),
                 b=rn(NQKV, scale=0.1), ln1=(1 + 0.1 * torch.randn(H, generator=g, device=DEV)).to(BF16),
                 ln2=(1 + 0.1 * torch.randn(H, generator=g, device=DEV)).to(BF16)) for _ in range(a.sets)]
    att, x0 = rn(H), rn(H)
    x, act, qkv = x0.clone(), torch.empty(I, dtype=BF16, device=DEV), torch.empty(NQKV, dtype=BF16, device=DEV)
    nwords = ops.decode_engine_sync_words(4)
    sync = torch.zeros((a.layers, nwords), dtype=torch.int32, device=DEV)
    status = torch.zeros(4, dtype=torch.int32, device=DEV)
    bytes_layer = 2 * (H * H + 2 * I * H + H * I + NQKV * H)

    def phases(w):
        return [dict(A=att, W=w["wo"], C=x, residual=x), dict(A=x, W=w["wgu"], C=act, norm_w=w["ln1"], epilogue=ops.EPI_SWIGLU16),
                dict(A=act, W=w["wd"], C=x, residual=x), dict(A=x, W=w["wqkv"], C=qkv, norm_w=w["ln2"], bias=w["b"])]

    def launch_form():
        for li in range(a.layers):
            for ph in phases(sets[li % a.sets]):
                ops.gemv(ph["A"].view(1, -1), ph["W"], ph["C"].view(1, -1), bias=ph.get("bias"),
                         residual=None if ph.get("residual") is None else ph["residual"].view(1, -1), epilogue=ph.get("epilogue", 0),
                         norm_w=ph.get("norm_w"), eps=1e-6)

    def engine_form():
        sync.zero_()
        for li in range(a.layers):
            ops.decode_engine(phases(sets[li % a.sets]), 1e-6, sync[li], status)

    def timed(fn, label):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with ops.HipGraph.capture(side) as gr:
            fn()
        x.copy_(x0)
        gr.launch(); side.synchronize()            # warm-up (and the result the other form is compared with)
        res = (x.clone(), act.clone(), qkv.clone())
        ts = []
        for _ in range(a.reps):
            x.copy_(x0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            gr.launch()
            side.synchronize()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        med = ts[len(ts) // 2] * 1e6
        per = med / a.layers
        print(f"{label:58s} {med:9.1f} us / {a.layers} layers = {per:7.2f} us per layer = {bytes_layer / per / 1e6:5.2f} TB/s of weights   (min {ts[0] * 1e6 / a.layers:.2f})", flush=True)
        return res

    if a.trace:
        trace_report(a, sets, phases, sync, status, x, x0)
        return
    print(f"# chain o -> gate/up -> down -> qkv at H={H} I={I}: {bytes_layer / 1e6:.1f} MB of bf16 weights per layer; {a.sets} weight sets cycled; "
          f"{ops.decode_engine_workgroups()} workgroups; median of {a.reps} graph replays (host-timed: + ~10-16 us of replay floor per graph, /{a.layers} per layer)")
    ref = timed(launch_form, "launch form: 4 x bagel_gemv_bf16 per layer")
    for env in [dict(), dict(BAGEL_ENGINE_DEPTH="1"), dict(BAGEL_ENGINE_LOADERS="1", BAGEL_ENGINE_WAVES="4", BAGEL_ENGINE_DEPTH="3"), dict(BAGEL_ENGINE_NT="0"),
                dict(BAGEL_ENGINE_WAVES="5"), dict(BAGEL_ENGINE_WAVES="6"), dict(BAGEL_ENGINE_WAVES="4"),
                dict(BAGEL_ENGINE_SLOTS="4"),
                # timing-only ablations (wrong results by construction): what is left when one of the three parties is taken out
                dict(BAGEL_ENGINE_ABL="1"), dict(BAGEL_ENGINE_ABL="2"), dict(BAGEL_ENGINE_ABL="4"), dict(BAGEL_ENGINE_ABL="5"), dict(BAGEL_ENGINE_ABL="6"),
                dict(BAGEL_ENGINE_ABL="3"), dict(BAGEL_ENGINE_ABL="7")]:
        for k in ("BAGEL_ENGINE_DEPTH", "BAGEL_ENGINE_NT", "BAGEL_ENGINE_WAVES", "BAGEL_ENGINE_SLOTS", "BAGEL_ENGINE_ABL", "BAGEL_ENGINE_LOADERS"):
            os.environ.pop(k, None)
        if a.quick and env:
            break
        os.environ.update(env)
        got = timed(engine_form, "engine: " + (" ".join(f"{k[13:]}={v}" for k, v in env.items()) or "defaults (2 loaders x depth 2, 6 consumers, nt, 6 slots)"))
        code = int(status[0])
        same = all(torch.equal(p, q) for p, q in zip(ref, got))
        print(f"    bit-identical to the launch form: {same}; status 0x{code:x}", flush=True)
        if code:
            status.zero_()


def trace_report(a, sets, phases, sync, status, x, x0):
    """Eager traced launches (one per weight set, the first is a warm-up): per phase, over the workgroups, min / median / max of every event relative to
    the launch's first event, in microseconds."""
    import numpy as np
    nwg = ops.decode_engine_workgroups()
    names = ["loader first issue", "loader last issue", "loader blocked: no free slot (sum)", "loader blocked: counted DMA wait (sum)",
             "c0 hand-off begin", "c0 flags seen", "c0 activation staged", "c0 last unit done", "c1 last unit done", "c2 last unit done",
             "c0 waited for full slots (sum)", "c1 waited for full slots (sum)", "c2 waited for full slots (sum)", "workgroup flag stored"]
    durations = {2, 3, 10, 11, 12}
    for it in range(len(sets)):
        tr = torch.zeros((nwg, 4, 16), dtype=torch.int64, device=DEV)
        x.copy_(x0)
        sync.zero_()
        torch.cuda.synchronize()
        ops.decode_engine(phases(sets[it]), 1e-6, sync[0], status, trace=tr)
        torch.cuda.synchronize()
        if it == 0:
            continue
        t = tr.cpu().numpy().astype(np.float64) / 100.0          # 100 MHz ticks -> us
        ev = [k for k in range(14) if k not in durations]
        t0 = min(t[:, 0, k][t[:, 0, k] > 0].min() for k in ev if (t[:, 0, k] > 0).any())
        end = max(t[:, 3, k].max() for k in (7, 8, 9))
        print(f"## traced launch on weight set {it}: {end - t0:.1f} us from the first event to the last unit; status 0x{int(status[0]):x}")
        for ph, pname in enumerate(["o_proj", "gate/up", "down", "qkv"]):
            print(f"  phase {ph} ({pname})")
            for k in range(14):
                v = t[:, ph, k]
                v = v[v > 0] if k not in durations else v
                if v.size == 0:
                    continue
                if k not in durations:
                    v = v - t0
                print(f"    {names[k]:42s} min {v.min():8.2f}  p10 {np.percentile(v, 10):8.2f}  med {np.median(v):8.2f}  p90 {np.percentile(v, 90):8.2f}  max {v.max():8.2f}")


if __name__ == "__main__":
    main()
