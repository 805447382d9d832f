"""CPU: the host planner of the persistent attention kernel (``bagel_attn_plan`` in csrc/attention2.hip -- plain host C++ inside
libbagel_hip.so, called over ctypes WITHOUT a GPU) against the flash-attn definition.

``execute_plan`` below interprets a plan literally, item by item, in fp64 torch: every item = (query rows, head or GQA group in
head-per-wave form, a range of 64-key tiles over [context tiles | new tiles], bottom-right causal mask), key-split items leave
(max, sum, un-normalised O) partials that the combine table merges.  If the planner drops, duplicates or mis-addresses any (row, head,
key) the result differs from the definition -- so this pins the work decomposition the GPU kernel executes, on random ragged batches
and on the benchmark's launches, and checks the schedule it promises (balance, XCD grouping)."""
import math

import pytest
import torch

from bagel_amd import ops


def _cols(lens):
    out, c = [], 0
    for n in lens:
        out.append(c)
        c += (max(n, 1) + 63) // 64 * 64
    return out


def _starts(lens):
    out, c = [], 0
    for n in lens:
        out.append(c)
        c += n
    return out


def make_plan(q_lens, ctx_lens, nq, nkv, causal, n_workers=256, split_min_tiles=0, gap=0):
    """Plan in the layout attn_varlen uses: packed query rows (optionally with `gap` unused rows between samples, as explicit row
    ranges allow), packed context rows, 64-aligned V^T columns."""
    q_start = _starts([n + gap for n in q_lens])
    return ops.AttnPlan(q_start, q_lens, _cols(q_lens), nq, nkv, 128, causal, "cpu", ctx_start=_starts(ctx_lens), ctx_len=ctx_lens,
                        vt_ctx_col=_cols(ctx_lens), n_workers=n_workers, split_min_tiles=split_min_tiles)


def execute_plan(ap, q, k_new, v_new, k_ctx, v_ctx, scale):
    """q [rows, nq, D], k_new / v_new [rows, nkv, D], k_ctx / v_ctx [ctx rows, nkv, D] (fp64) -> out [rows, nq, D], cover [rows, nq]."""
    out = torch.full(q.shape, float("nan"), dtype=torch.float64)
    cover = torch.zeros(q.shape[:2], dtype=torch.int32)
    parts = {}
    for it in ap.items().tolist():
        q_row0, nrows, h, g, flags, q_rel0, kn_row0, l_new, kc_row0, l_ctx, vtn, vtc, t0, t1, part, _ = it
        hpw, causal, partial, G = flags & 1, bool(flags & 2), bool(flags & 4), (flags >> 8) & 255
        assert 0 <= t0 < t1 and 1 <= nrows <= (32 if hpw else 256)
        nt_ctx = (l_ctx + 63) // 64
        nt_all = nt_ctx + (l_new + 63) // 64
        assert t1 <= nt_all
        ks, vs, new_idx = [], [], []
        for tt in range(t0, t1):
            if tt < nt_ctx:
                a, b = tt * 64, min(l_ctx, tt * 64 + 64)
                ks.append(k_ctx[kc_row0 + a:kc_row0 + b, g]); vs.append(v_ctx[kc_row0 + a:kc_row0 + b, g])
                new_idx += [-1] * (b - a)
            else:
                a, b = (tt - nt_ctx) * 64, min(l_new, (tt - nt_ctx) * 64 + 64)
                ks.append(k_new[kn_row0 + a:kn_row0 + b, g]); vs.append(v_new[kn_row0 + a:kn_row0 + b, g])
                new_idx += list(range(a, b))
        K, V, idx = torch.cat(ks), torch.cat(vs), torch.tensor(new_idx)
        heads = [h + w for w in range(G)] if hpw else [h]
        for hh in heads:
            assert hh // (ap.nq // ap.nkv) == g
            Q = q[q_row0:q_row0 + nrows, hh]
            s = Q @ K.t() * scale
            if causal:
                qrel = q_rel0 + torch.arange(nrows)
                s = s.masked_fill(idx[None, :] > qrel[:, None], float("-inf"))
            m = s.max(dim=1).values
            p = torch.exp(s - torch.where(torch.isfinite(m), m, torch.zeros_like(m))[:, None])
            l, O = p.sum(1), p @ V
            if partial:
                parts.setdefault(part, {})[hh] = (m, l, O)
            else:
                out[q_row0:q_row0 + nrows, hh] = O / l[:, None]
                cover[q_row0:q_row0 + nrows, hh] += 1
    used = set()
    for c in ap.combines().tolist():
        q_row0, nrows, h, flags, slot0, nslots = c[:6]
        hpw, G = flags & 1, (flags >> 8) & 255
        for hh in ([h + w for w in range(G)] if hpw else [h]):
            ms = torch.stack([parts[slot0 + s][hh][0] for s in range(nslots)])
            M = ms.max(0).values
            w = torch.where(torch.isfinite(ms), torch.exp(ms - M[None]), torch.zeros_like(ms))
            L = sum(w[s] * parts[slot0 + s][hh][1] for s in range(nslots))
            O = sum(w[s][:, None] * parts[slot0 + s][hh][2] for s in range(nslots))
            out[q_row0:q_row0 + nrows, hh] = O / L[:, None]
            cover[q_row0:q_row0 + nrows, hh] += 1
        used |= set(range(slot0, slot0 + nslots))
    assert used == set(parts) == set(range(ap.n_slots)), "partial slots and the combine table disagree"
    return out, cover


def reference(q, k_new, v_new, k_ctx, v_ctx, q_start, q_lens, ctx_lens, nq, nkv, causal, scale):
    out = torch.full(q.shape, float("nan"), dtype=torch.float64)
    G = nq // nkv
    c0 = 0
    for b, (s0, Lq, C) in enumerate(zip(q_start, q_lens, ctx_lens)):
        for h in range(nq):
            g = h // G
            K = torch.cat([k_ctx[c0:c0 + C, g], k_new[s0:s0 + Lq, g]])
            V = torch.cat([v_ctx[c0:c0 + C, g], v_new[s0:s0 + Lq, g]])
            s = q[s0:s0 + Lq, h] @ K.t() * scale
            if causal:
                keep = torch.ones(Lq, C + Lq, dtype=torch.bool).tril(diagonal=C)
                s = s.masked_fill(~keep, float("-inf"))
            out[s0:s0 + Lq, h] = torch.softmax(s, -1) @ V
        c0 += C
    return out


CASES = [
    # q_lens, ctx_lens, nq, nkv, causal, gap
    ([5], [0], 4, 2, True, 0),
    ([300, 17, 256, 1], [0, 40, 7, 700], 8, 2, False, 0),
    ([300, 17, 256, 1], [0, 40, 7, 700], 8, 2, True, 3),
    ([258, 258], [32, 0], 28, 4, False, 0),                 # two 256 + 2 samples: head-per-wave tails
    ([1100], [0], 6, 6, False, 0),                          # G = 1 (SigLIP-like): no head-per-wave form
    ([34, 34, 34], [500, 0, 64], 28, 4, True, 0),           # short prompts on contexts: everything head-per-wave
    ([700], [1300], 16, 2, True, 0),                        # G = 8
    ([520, 40], [0, 0], 18, 2, False, 5),                   # G = 9 > 8 waves: tile form even for the short tail
]


@pytest.mark.parametrize("q_lens,ctx_lens,nq,nkv,causal,gap", CASES)
@pytest.mark.parametrize("n_workers,split_min", [(256, 0), (16, 2), (8, 1)])  # the chip; few workers = many rounds; every tail split
def test_plan_reproduces_attention(q_lens, ctx_lens, nq, nkv, causal, gap, n_workers, split_min):
    D = 8                                               # the plan does not depend on head_dim; a small one keeps the test fast
    g = torch.Generator().manual_seed(sum(q_lens) * 3 + nq + n_workers)
    ap = make_plan(q_lens, ctx_lens, nq, nkv, causal, n_workers=n_workers, split_min_tiles=split_min, gap=gap)
    rows = ap.q_start[-1] + q_lens[-1] + gap
    q = torch.randn(rows, nq, D, generator=g, dtype=torch.float64)
    kn, vn = torch.randn(rows, nkv, D, generator=g, dtype=torch.float64), torch.randn(rows, nkv, D, generator=g, dtype=torch.float64)
    crow = max(sum(ctx_lens), 1)
    kc, vc = torch.randn(crow, nkv, D, generator=g, dtype=torch.float64), torch.randn(crow, nkv, D, generator=g, dtype=torch.float64)
    scale = D ** -0.5
    got, cover = execute_plan(ap, q, kn, vn, kc, vc, scale)
    ref = reference(q, kn, vn, kc, vc, ap.q_start, q_lens, ctx_lens, nq, nkv, causal, scale)
    live = torch.zeros(rows, dtype=torch.bool)
    for s0, n in zip(ap.q_start, q_lens):
        live[s0:s0 + n] = True
    assert (cover[live] == 1).all() and (cover[~live] == 0).all(), "every (row, head) must be produced exactly once"
    assert torch.allclose(got[live], ref[live], rtol=1e-9, atol=1e-11)
    # bookkeeping the kernel relies on
    off = ap.worker_off()
    assert off[0] == 0 and off[-1] == ap.n_items and (off[1:] >= off[:-1]).all()
    items = ap.items()
    assert (items[:, 12] < items[:, 13]).all()
    for b, (s0, n) in enumerate(zip(ap.q_start, q_lens)):
        mine = items[(items[:, 0] >= s0) & (items[:, 0] < s0 + n)]
        assert (mine[:, 6] == s0).all() and (mine[:, 7] == n).all() and (mine[:, 9] == ctx_lens[b]).all()
        assert (mine[:, 10] == ap.vt_new_col[b]).all()
        if ctx_lens[b]:
            assert (mine[:, 8] == ap.ctx_start[b]).all() and (mine[:, 11] == ap.vt_ctx_col[b]).all()


def _loads(ap):
    off, items = ap.worker_off(), ap.items()
    return [int((items[off[w]:off[w + 1], 13] - items[off[w]:off[w + 1], 12]).sum()) for w in range(ap.n_workers)]


def test_schedule_of_the_denoise_launch():
    """BASELINE configs[2] as the stream-batched forward launches it: 8 samples (4 cond on 32-token contexts + 4 CFG without) of 4098
    rows, 28 / 4 heads.  16 full query tiles x 28 heads x 8 samples = 3584 tile items = 14 per worker; the 2-row tails are 32
    head-per-wave items; what does not fit 14 whole items per worker is cut along the key axis: the busiest worker runs 14 items + a
    fraction, not 15 (the one-tile-per-workgroup kernel: 3808 workgroups = 15 rounds)."""
    ap = make_plan([4098] * 8, [32] * 4 + [0] * 4, 28, 4, False)
    items = ap.items()
    hpw = items[:, 4] & 1 == 1
    assert len({(int(i[0]), int(i[3])) for i in items[hpw]}) == 32, "one head-per-wave tail per (sample, KV head) pair"
    assert (items[hpw, 1] == 2).all() and (items[~hpw, 1] == 256).all()
    whole = items[(items[:, 4] & 4) == 0]
    assert len(whole) >= 3584 + 32 - 40 and ap.n_comb <= 40 and ap.n_slots <= 320
    loads = _loads(ap)
    assert ap.makespan == max(loads) <= 14 * 66 + 8
    assert min(loads) >= 14 * 65
    assert ap.total == sum(loads)
    # every XCD works on whole (sample, KV head) pairs: the first items of its 32 workers share the pair
    off = ap.worker_off()
    for x in range(8):
        first = [items[off[w]] for w in range(x, 256, 8)]
        assert len({(int(i[6]), int(i[3])) for i in first}) == 1, "the first round of an XCD must be one (sample, KV head) pair"


def test_schedule_of_the_edit_and_prefill_launches():
    def ratio(ap):
        loads = _loads(ap)
        assert ap.makespan == max(loads)
        return max(loads) / (sum(loads) / 256)
    # the 3-stream edit forward: 12 (sample, KV head) pairs of very different cost (contexts 9032 / 9000 / 32) -> interleaved halves per
    # XCD, longest-first dealing, leftovers cut along the key axis: 5.25 rounds' worth of work, not 6
    ap = make_plan([4098] * 3, [9032, 9000, 32], 28, 4, False)
    assert ratio(ap) <= 1.02 and ap.n_comb > 0
    # one stream of the denoise forward; the causal LLM prefill of the understanding request (4936 tokens, batch 1; items below the split
    # threshold stay whole); SigLIP (16 heads, one 4900-token image: 320 tile items = 1.25 rounds)
    assert ratio(make_plan([4098] * 4, [32] * 4, 28, 4, False)) <= 1.02
    assert ratio(make_plan([4936], [0], 28, 4, True)) <= 1.08
    assert ratio(make_plan([4900], [0], 16, 16, False)) <= 1.03


def test_plan_buffer_too_small_is_reported():
    import numpy as np
    from bagel_amd._lib import lib
    a = np.asarray([0], dtype=np.int32)
    n = np.asarray([5000], dtype=np.int32)
    buf = np.zeros(64, dtype=np.int32)
    rc = lib().bagel_attn_plan(a.ctypes.data, n.ctypes.data, a.ctypes.data, a.ctypes.data, a.ctypes.data, a.ctypes.data, 1, 28, 4, 0, 256, 0,
                               buf.ctypes.data, 64)
    assert rc != 0 and b"plan needs" in lib().bagel_hip_last_error()
    rc = lib().bagel_attn_plan(a.ctypes.data, n.ctypes.data, a.ctypes.data, a.ctypes.data, a.ctypes.data, a.ctypes.data, 1, 28, 4, 0, 100, 0,
                               buf.ctypes.data, 64)
    assert rc != 0 and b"multiple of the 8" in lib().bagel_hip_last_error()
