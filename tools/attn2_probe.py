"""One-tile-per-workgroup attention (bagel_attn_varlen_bf16) vs the planned persistent kernel (bagel_attn_planned_bf16) at the launches the
benchmark makes, same device tensors, same box, back to back:  time, TFLOP/s (algorithmic FLOPs 4 * Lq * Lkv * nq * D per sample; causal
counts the visible half), max |difference|.  Run on the GPU box:  python tools/attn2_probe.py [--iters 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bagel_amd import ops  # noqa: E402

BF16, DEV = torch.bfloat16, "cuda"


def timeit(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def case(name, q_lens, ctx_lens, nq, nkv, causal, iters, D=128, split_min_tiles=0):
    B = len(q_lens)
    M, Ctot = sum(q_lens), sum(ctx_lens)
    g = torch.Generator(device=DEV).manual_seed(0)
    rn = lambda *s: torch.randn(*s, generator=g, device=DEV).to(BF16)  # noqa: E731
    qw, kw = nq * D, nkv * D
    qkv = rn(M, qw + 2 * kw)
    i32 = lambda x: torch.tensor(x, dtype=torch.int32, device=DEV)  # noqa: E731

    def cu(lens):
        out = [0]
        for n in lens:
            out.append(out[-1] + n)
        return out

    def cols(lens):
        out, c = [], 0
        for n in lens:
            out.append(c)
            c += (max(n, 1) + 63) // 64 * 64
        return out, c
    cu_q, cu_c = cu(q_lens), cu(ctx_lens)
    vcol, vtot = cols(q_lens)
    ccol, ctot = cols(ctx_lens)
    vt = torch.zeros((kw, (vtot + 255) // 256 * 256), dtype=BF16, device=DEV)
    ops.v_transpose(qkv[:, qw + kw:], vt, i32(cu_q), i32(vcol), B, max(q_lens), nkv, D)
    kw_ = {}
    kc = vtc = None
    if Ctot:
        kc, vc = rn(Ctot, kw), rn(Ctot, kw)
        vtc = torch.zeros((kw, (ctot + 255) // 256 * 256), dtype=BF16, device=DEV)
        ops.v_transpose(vc, vtc, i32(cu_c), i32(ccol), B, max(ctx_lens), nkv, D)
        kw_ = dict(k_ctx=kc, vt_ctx=vtc, cu_ctx=i32(cu_c), vt_ctx_col=i32(ccol))
    out1 = torch.empty((M, qw), dtype=BF16, device=DEV)
    out2 = torch.empty((M, qw), dtype=BF16, device=DEV)
    scale = D ** -0.5
    cq, cv = i32(cu_q), i32(vcol)
    run1 = lambda: ops.attn_varlen(qkv[:, :qw], qkv[:, qw:qw + kw], vt, out1, cq, cv, B, max(q_lens), nq, nkv, D, causal, scale, **kw_)  # noqa: E731
    ap = ops.AttnPlan(cu_q[:-1], q_lens, vcol, nq, nkv, D, causal, DEV, ctx_start=cu_c[:-1] if Ctot else None, ctx_len=ctx_lens if Ctot else None,
                      vt_ctx_col=ccol if Ctot else None, split_min_tiles=split_min_tiles)
    run2 = lambda: ops.attn_planned(qkv[:, :qw], qkv[:, qw:qw + kw], vt, out2, ap, scale, k_ctx=kc, vt_ctx=vtc)  # noqa: E731
    extra = []
    for nw in WORKERS:
        apx = ops.AttnPlan(cu_q[:-1], q_lens, vcol, nq, nkv, D, causal, DEV, ctx_start=cu_c[:-1] if Ctot else None, ctx_len=ctx_lens if Ctot else None,
                           vt_ctx_col=ccol if Ctot else None, split_min_tiles=split_min_tiles, n_workers=nw)
        outx = torch.empty((M, qw), dtype=BF16, device=DEV)
        extra.append((nw, apx, outx, (lambda a=apx, o=outx: ops.attn_planned(qkv[:, :qw], qkv[:, qw:qw + kw], vt, o, a, scale, k_ctx=kc, vt_ctx=vtc))))
    run1(); run2()
    for _, _, _, r in extra:
        r()
    torch.cuda.synchronize()
    diff = (out1.float() - out2.float()).abs().max().item()
    nan = not bool(torch.isfinite(out2.float()).all())
    fl = sum(4.0 * lq * (lq * (0.5 if causal else 1.0) + c) * nq * D for lq, c in zip(q_lens, ctx_lens))
    t1, t2 = timeit(run1, iters), timeit(run2, iters)
    print(f"{name:28s} tile kernel {t1:8.3f} ms {fl / t1 / 1e9:7.1f} TF | planned {t2:8.3f} ms {fl / t2 / 1e9:7.1f} TF  ({t1 / t2:5.3f}x)  "
          f"max|d| {diff:.3g}{' NON-FINITE' if nan else ''}  items {ap.n_items} split {ap.n_comb} makespan {ap.makespan} ideal {ap.total / ap.n_workers:.1f}",
          flush=True)
    for nw, apx, outx, r in extra:
        tx = timeit(r, iters)
        dx = (outx.float() - out2.float()).abs().max().item()
        print(f"{'':28s}   planned with {nw} workers ({'split 2+2 ring, 64 KB' if nw > CUS else 'unified ring'}) {tx:8.3f} ms {fl / tx / 1e9:7.1f} TF  ({t2 / tx:5.3f}x of the default)  "
              f"max|d vs default| {dx:.3g}  items {apx.n_items} split {apx.n_comb} makespan {apx.makespan} ideal {apx.total / apx.n_workers:.1f}", flush=True)


WORKERS = []
CUS = 256


def main():
    global CUS
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default=None)
    ap.add_argument("--workers", type=int, nargs="*", default=[], help="extra plans with these worker counts (more than the CU count = the split-ring kernel, two workgroups per CU asked for)")
    a = ap.parse_args()
    WORKERS[:] = a.workers
    CUS = torch.cuda.get_device_properties(0).multi_processor_count
    if a.workers:
        from bagel_amd import _lib
        L = _lib.lib()
        print(f"CUs {CUS}; resident workgroups per CU by the runtime's occupancy calculator: head_dim 128 unified ring {L.bagel_debug_attn_occupancy(128, 0)}, "
              f"split ring {L.bagel_debug_attn_occupancy(128, 1)}; head_dim 64 unified {L.bagel_debug_attn_occupancy(64, 0)}, split {L.bagel_debug_attn_occupancy(64, 1)}", flush=True)
    cases = [("denoise_b8 (configs[2])", [4098] * 8, [32] * 4 + [0] * 4, 28, 4, False),
             ("denoise_b4 (one stream)", [4098] * 4, [32] * 4, 28, 4, False),
             ("edit_3streams (configs[4])", [4098] * 3, [9032, 9000, 32], 28, 4, False),
             ("prefill_4936_causal", [4936], [0], 28, 4, True),
             ("siglip_4900", [4900], [0], 16, 16, False),
             ("prompt_34_causal", [34], [0], 28, 4, True)]
    for c in cases:
        if a.only and a.only not in c[0]:
            continue
        case(*c, iters=a.iters)


if __name__ == "__main__":
    main()
