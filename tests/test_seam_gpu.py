"""GPU: the Level-2 seam of INTEGRATION.md EXECUTED -- ``bagel_amd/integration/flash_attn`` (the stub a maintainer of the reference
would put in front of it: the reference's one by-name native import, qwen2_navit.py:24,361-370,579-588; siglip_navit.py:14,232-241)
is imported BY PATH, exactly as a stock reference tree would import it, binds the C ABI over its own ctypes handle, and is compared
with the flash-attn definition (oracle.attn_varlen) on the call shapes the reference makes:

  * SigLIP: packed images, 16 heads x 72 (zero-padded to 128 lanes inside the stub), Lq == Lk, non-causal;
  * the cached LLM forward: merged [context | new] keys per sample, GQA 28 / 4 x 128, causal (bottom-right) and non-causal;
  * a CFG forward without context (Lq == Lk, several samples)."""
import importlib.util
import os
import sys

import pytest
import torch

from tests.test_ops_gpu import BF16, DEV, close, rnd

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def stub():
    path = os.path.join(ROOT, "bagel_amd", "integration", "flash_attn", "__init__.py")
    spec = importlib.util.spec_from_file_location("flash_attn_seam_under_test", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _case(q_lens, k_lens, nq, nkv, D, seed):
    Tq, Tk = sum(q_lens), sum(k_lens)
    q, k, v = rnd(Tq, nq, D, seed=seed + 1), rnd(Tk, nkv, D, seed=seed + 2), rnd(Tk, nkv, D, seed=seed + 3)
    cu = lambda lens: torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)  # noqa: E731
    return q, k, v, cu(q_lens), cu(k_lens)


@pytest.mark.parametrize("q_lens,k_lens,nq,nkv,D,causal", [
    ([1225, 700, 64], [1225, 700, 64], 16, 16, 72, False),          # SigLIP so400m heads (siglip_navit.py:232-241)
    ([4900], [4900], 16, 16, 72, False),                            # one 980^2 image
    ([5, 300, 1], [45, 600, 130], 28, 4, 128, True),                # text prefill / decode on cached contexts (qwen2_navit.py:579-588)
    ([258, 66], [290, 66], 28, 4, 128, False),                      # latent rows on a text context; a sample without context
    ([130, 70], [130, 70], 4, 2, 64, True),                         # no cache (qwen2_navit.py:361-370)
], ids=["siglip_3img", "siglip_980", "llm_cached_causal", "latents_ctx", "nocache_causal_d64"])
def test_flash_attn_stub_matches_definition(stub, q_lens, k_lens, nq, nkv, D, causal):
    from oracle.bagel_oracle import attn_varlen
    q, k, v, cq, ck = _case(q_lens, k_lens, nq, nkv, D, seed=len(q_lens) * 7 + D)
    ref = attn_varlen(q, k, v, cq, ck, max(q_lens), max(k_lens), causal=causal)
    got = stub.flash_attn_varlen_func(q.to(DEV), k.to(DEV), v.to(DEV), cq.to(DEV), ck.to(DEV), max(q_lens), max(k_lens), causal=causal)
    assert got.shape == q.shape and got.dtype == BF16
    close(got, ref, ulps=2, rel_l2=6e-3, what=f"flash_attn stub q={q_lens} k={k_lens} D={D} causal={causal}")


def test_flash_attn_stub_fails_loudly(stub):
    q, k, v, cq, ck = _case([8], [8], 2, 2, 64, seed=1)
    with pytest.raises(RuntimeError):
        stub.flash_attn_varlen_func(q, k, v, cq, ck, 8, 8)                       # host tensors: no CPU fallback
    with pytest.raises(RuntimeError):
        stub.flash_attn_varlen_func(q.to(DEV), k.to(DEV)[:4], v.to(DEV)[:4], cq.to(DEV), torch.tensor([0, 4], dtype=torch.int32, device=DEV), 8, 4)


@pytest.mark.parametrize("lens,nq,nkv,D,causal", [([300, 64, 130], 16, 16, 72, False), ([257, 70], 4, 2, 64, True), ([140], 28, 4, 128, False)],
                         ids=["siglip_heads_72", "causal_d64_gqa", "gqa_d128"])
def test_flash_attn_stub_backward_for_self_attention(stub, lens, nq, nkv, D, causal):
    """The stub under autograd, as the stock reference's SigLIP uses it in training (siglip_navit.py:232-241): gradients of q / k / v vs
    torch autograd of the flash-attn definition (the oracle's shim, fp32 softmax)."""
    from oracle import bagel_oracle as O
    q, k, v, cq, ck = _case(lens, lens, nq, nkv, D, seed=len(lens) * 5 + D)
    do = rnd(sum(lens), nq, D, seed=99)
    qf, kf, vf = (t.clone().requires_grad_(True) for t in (q, k, v))
    O.GRAD_ENABLED = True
    try:
        with torch.enable_grad():
            ref = O.attn_varlen(qf, kf, vf, cq, ck, max(lens), max(lens), causal=causal)
        ref.backward(do)
    finally:
        O.GRAD_ENABLED = False
    qd, kd, vd = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
    got = stub.flash_attn_varlen_func(qd, kd, vd, cq.to(DEV), ck.to(DEV), max(lens), max(lens), causal=causal)
    assert got.requires_grad and got.shape == q.shape
    close(got, ref, ulps=2, rel_l2=6e-3, what="forward under autograd")
    got.backward(do.to(DEV))
    for name, a, b in (("dq", qd.grad, qf.grad), ("dk", kd.grad, kf.grad), ("dv", vd.grad, vf.grad)):
        assert a.shape == b.shape and a.dtype == BF16
        rel = ((a.float().cpu() - b.float()).norm() / b.float().norm()).item()
        assert rel < 1.5e-2, (name, rel)
    # the cached two-segment form stays inference-only
    q2, k2, v2, cq2, ck2 = _case([5], [45], 4, 2, 64, seed=3)
    with pytest.raises(NotImplementedError):
        stub.flash_attn_varlen_func(q2.to(DEV).requires_grad_(True), k2.to(DEV), v2.to(DEV), cq2.to(DEV), ck2.to(DEV), 5, 45, causal=True)
