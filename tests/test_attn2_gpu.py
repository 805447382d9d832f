"""GPU parity of the PLANNED, persistent attention kernel (csrc/attention2.hip: host work list, head-per-wave tail tiles, key-split
last round + combine, continuous K / V^T tile stream across item seams) through the C ABI:

  * against the flash-attn definition (oracle.attn_varlen) on every shape tests/test_ops_gpu.py runs the one-tile-per-workgroup kernel
    on -- ragged, causal, contexts, D = 64 / 128, GQA 1 / 2 / 7 -- with the chip's worker count, with 8 workers (long item runs per
    worker: many seams) and with every partial round split;
  * against bagel_attn_varlen_bf16 BIT FOR BIT where no item is key-split (the same MFMA order and rounding points per query row);
  * at the benchmark's launches: the stream-batched denoise forward (8 x 4098 rows), the edit forward (9032-key context), the causal
    4936-token prefill and the SigLIP image (4900 tokens, 16 heads)."""
import pytest
import torch

from tests.test_ops_gpu import ATTN_CASES, close, run_attention

pytestmark = pytest.mark.gpu


def _unsplit_rows(ap, M, nq):
    """mask [M, nq]: outputs produced by items that are NOT key-split."""
    m = torch.ones(M, nq, dtype=torch.bool)
    for c in ap.combines().tolist():
        q_row0, nrows, h, flags = c[:4]
        heads = range(h, h + ((flags >> 8) & 255)) if flags & 1 else [h]
        for hh in heads:
            m[q_row0:q_row0 + nrows, hh] = False
    return m


# "2xchip": a plan with more workers than CUs selects the SPLIT-RING form of the kernel (2 K + 2 V^T slots, two cursors: round 6) -- same arithmetic, same order
@pytest.mark.parametrize("q_lens,ctx_lens,nq,nkv,D,causal", ATTN_CASES)
@pytest.mark.parametrize("planned", [dict(), dict(n_workers=8), dict(n_workers=16, split_min_tiles=1), dict(n_workers="2xchip"), dict(n_workers="2xchip", split_min_tiles=1)],
                         ids=["chip", "w8", "w16_split", "split_ring", "split_ring_keysplit"])
def test_planned_attention_matches_definition_and_tile_kernel(q_lens, ctx_lens, nq, nkv, D, causal, planned):
    got, ref, got2, ap = run_attention(q_lens, ctx_lens, nq, nkv, D, causal, planned=planned)
    close(got2, ref, ulps=2, rel_l2=6e-3, what=f"planned attn q={q_lens} ctx={ctx_lens} D={D} causal={causal} {planned}")
    keep = _unsplit_rows(ap, got.shape[0], nq)
    a, b = got.cpu().view(torch.int16)[keep], got2.cpu().view(torch.int16)[keep]
    assert torch.equal(a, b), f"un-split items must equal the tile kernel bit for bit ({int((a != b).any(-1).sum())} rows differ)"


@pytest.mark.parametrize("q_lens,ctx_lens,nq,nkv,causal", [
    ([4098] * 8, [32] * 4 + [0] * 4, 28, 4, False),          # stream-batched denoise forward of BASELINE configs[2]
    ([4098] * 3, [9032, 9000, 32], 28, 4, False),            # 3-stream edit forward (configs[4])
    ([4936], [0], 28, 4, True),                              # causal LLM prefill of the understanding request (configs[1])
    ([4900], [0], 16, 16, False),                            # SigLIP: one 980^2 image
    ([34], [0], 28, 4, True),                                # short text prefill: four head-per-wave items
], ids=["denoise_b8", "edit_3streams", "prefill_4936_causal", "siglip_4900", "prompt_34"])
@pytest.mark.parametrize("planned", [dict(), dict(n_workers="2xchip")], ids=["unified_ring", "split_ring"])
def test_planned_attention_at_benchmark_launches(q_lens, ctx_lens, nq, nkv, causal, planned):
    got, ref, got2, ap = run_attention(q_lens, ctx_lens, nq, nkv, 128, causal, planned=planned)
    close(got2, ref, ulps=2, rel_l2=6e-3, what=f"planned attn q={q_lens} ctx={ctx_lens}")
    keep = _unsplit_rows(ap, got.shape[0], nq)
    assert torch.equal(got.cpu().view(torch.int16)[keep], got2.cpu().view(torch.int16)[keep])
    print(f"plan q={q_lens[:2]}.. ctx={ctx_lens[:2]}..: items {ap.n_items}, key-split {ap.n_comb} (slots {ap.n_slots}), makespan {ap.makespan} tile steps "
          f"vs {ap.total / ap.n_workers:.1f} ideal")


def test_planned_attention_is_deterministic_and_leaves_other_rows_alone():
    """Two launches on the same inputs agree bit for bit (counted waits across item seams: a race shows up as a sporadic mismatch), and
    rows outside every sample's range keep their contents."""
    from tests.test_ops_gpu import BF16, DEV, ops, rnd
    o = ops()
    nq, nkv, D = 28, 4, 128
    q_lens, ctx_lens = [1100, 258, 770], [0, 300, 64]
    M = sum(q_lens)
    g = lambda *s, seed: rnd(*s, seed=seed).to(DEV)  # noqa: E731
    qkv = g(M + 10, (nq + 2 * nkv) * D, seed=1)
    qw, kw = nq * D, nkv * D
    cu = [0, 1100, 1358, 2128]
    vcol = [0, 1152, 1472]
    i32 = lambda x: torch.tensor(x, dtype=torch.int32, device=DEV)  # noqa: E731
    vt = torch.zeros((kw, 2304), dtype=BF16, device=DEV)
    o.v_transpose(qkv[:, qw + kw:], vt, i32(cu), i32(vcol), 3, 1100, nkv, D)
    kc = g(364, kw, seed=2)
    vtc = torch.zeros((kw, 512), dtype=BF16, device=DEV)
    o.v_transpose(g(364, kw, seed=3), vtc, i32([0, 0, 300, 364]), i32([0, 64, 384]), 3, 300, nkv, D)
    ap = o.AttnPlan(cu[:-1], q_lens, vcol, nq, nkv, D, False, DEV, ctx_start=[0, 0, 300], ctx_len=ctx_lens, vt_ctx_col=[0, 64, 384], n_workers=16)
    outs = []
    for rep in range(4):
        out = torch.full((M + 10, qw), 7.0, dtype=BF16, device=DEV)
        o.attn_planned(qkv[:, :qw], qkv[:, qw:qw + kw], vt, out, ap, D ** -0.5, k_ctx=kc, vt_ctx=vtc)
        outs.append(out)
    assert all(torch.equal(outs[0].view(torch.int16), x.view(torch.int16)) for x in outs[1:])
    assert (outs[0][M:] == 7.0).all() and torch.isfinite(outs[0][:M].float()).all()
