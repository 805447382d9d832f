"""CPU: the HOST logic of the training backward (bagel_amd/modeling/bagel/train_step.py, Bagel._backward_losses: the tape, the reverse
chain, MoT routing of dX / dW, the attention-backward work items, the unpacking of the MI355X weight layouts into the reference's
parameter shapes) on the torch stand-ins of the launch wrappers (tests/mock_ops.py) against the oracle's autograd -- which is pinned
bit for bit to the unmodified reference's own ``loss.backward()`` (oracle/make_golden_train_grads.py, tests/test_reference_crosscheck.py).
The kernels themselves are checked on the MI355X by tests/test_train_backward_gpu.py."""
import random

import pytest
import torch

from oracle import bagel_oracle as O
from oracle.configs import TINY, TINY_D128, TINY_DENSE, TINY_MOE, TINY_ROPE
from tests import mock_ops
from tests.util_models import oracle_weights, pack_training_batch

CFGS = {"tiny": TINY, "tiny_d128": TINY_D128, "tiny_rope": TINY_ROPE, "tiny_dense": TINY_DENSE, "tiny_moe": TINY_MOE}
FROZEN = ("vit_pos_embed.", "latent_pos_embed.")


def rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / b.norm().clamp_min(1e-20))


def cpu_model(cfg):
    from bagel_amd.factory import build_bagel
    W, _ = oracle_weights(cfg)
    model, _ = build_bagel(cfg, device="cpu", with_vae=False)
    model.load_state_dict(W, strict=True)
    return model.to(torch.bfloat16).eval()


def trainable(model):
    names = []
    for n, p in model.named_parameters():
        if not n.startswith(FROZEN):
            p.requires_grad_(True)
            names.append(n)
    return names


def product_step(model, batch, noise, w_ce, **extra):
    for p in model.parameters():
        p.grad = None
    out = model(noise=noise, **batch, **extra)
    loss = O.training_step_loss(out, w_ce)
    loss.backward()
    return float(loss.detach()), {n: p.grad for n, p in model.named_parameters() if p.grad is not None}, out


def compare(grads, ref, names, tol, what, key_bias_is_zero=True):
    worst = ("", 0.0)
    for n in names:
        assert n in ref, n
        rn = float(ref[n].float().norm())
        if key_bias_is_zero and n.startswith("vit_model.") and n.endswith("k_proj.bias"):
            # softmax is invariant to a constant added to every key of a row: the exact gradient of SigLIP's key bias is ZERO, what both
            # sides hold is rounding noise -- check that it is noise (against the query bias of the same layer), not that the noises agree
            qn = float(ref[n.replace("k_proj", "q_proj")].float().norm())
            assert rn <= 0.1 * qn and (n not in grads or float(grads[n].float().norm()) <= 0.1 * qn), (what, n)
            continue
        if rn == 0.0:
            assert n not in grads or float(grads[n].float().norm()) == 0.0, (what, n)
            continue
        assert n in grads, (what, n, "no gradient")
        assert grads[n].shape == ref[n].shape and grads[n].dtype == ref[n].dtype, (what, n)
        d = rel(grads[n], ref[n])
        if d > worst[1]:
            worst = (n, d)
        assert d < tol, (what, n, d)
    return worst


@pytest.mark.parametrize("name", ["tiny", "tiny_d128", "tiny_dense", "tiny_moe"])
def test_training_step_gradients_match_the_reference_batch(golden, monkeypatch, name):
    """The hand-packed batch of oracle/make_golden.scenario_train (und sample + gen sample with causal / full / noise splits): every
    trainable parameter's gradient vs the oracle's autograd; for 'tiny' also vs the committed reference gradients.  tiny_dense / tiny_moe: the
    reverse of Qwen2DecoderLayer (no routing) and Qwen2MoEDecoderLayer (shared attention and layer norms, per-modality MLP and final norm)."""
    mock_ops.install(monkeypatch)
    cfg = CFGS[name]
    g = golden(f"{name}_train")
    batch, noise = g["batch"], g["noise"]
    gen = torch.Generator().manual_seed(5)
    w_ce = torch.rand(g["ce"].shape[0], generator=gen) + 0.5
    W, _ = oracle_weights(cfg)
    model = cpu_model(cfg)
    names = trainable(model)
    rloss, rgrads, rout = O.training_step_grads(W, cfg, batch, noise, w_ce, names=set(names))
    loss, grads, out = product_step(model, batch, noise, w_ce)
    assert abs(loss - rloss) < 2e-2 * abs(rloss)
    assert out["mse"].requires_grad and out["ce"].requires_grad
    worst = compare(grads, rgrads, names, 4e-2, name)
    print(f"[{name}] {len(names)} gradients, worst rel-L2 {worst[1]:.2e} at {worst[0]}")
    if name == "tiny":
        fx = golden("tiny_train_grads")
        assert torch.equal(fx["ce_loss_weights"], w_ce)
        compare(grads, fx["grads"], names, 4e-2, "fixture")
    # the flat split API gives the same gradients bit for bit (same plan, same launches)
    b2 = {k: v for k, v in batch.items() if k != "nested_attention_masks"}
    _, grads2, _ = product_step(model, b2, noise, w_ce, split_lens=g["split_lens"], attn_modes=g["attn_modes"])
    for n in names:
        if n in grads:
            assert torch.equal(grads[n], grads2[n]), n


@pytest.mark.parametrize("seed", list(range(6)))
def test_training_step_gradients_on_random_batches(monkeypatch, seed):
    """Differential fuzz: random packs (1-3 samples, 1-5 splits, prompts / ViT images / clean and noised VAE images, CE / MSE anywhere;
    chunks longer than one 128-row work item included) -- gradients vs the oracle's autograd."""
    mock_ops.install(monkeypatch)
    rng = random.Random(3000 + seed)
    cfg = TINY if seed % 2 == 0 else TINY_D128
    samples = []
    for _ in range(rng.randint(1, 3)):
        sp = []
        for _ in range(rng.randint(1, 5)):
            kind = rng.choice(["text", "text", "vit", "vae", "vae"])
            if kind == "text":
                sp.append(("text", rng.randint(1, 7), rng.random() < 0.5))
            elif kind == "vit":
                sp.append(("vit", 14 * rng.randint(1, 4), 14 * rng.randint(1, 4)))
            else:
                sp.append(("vae", 16 * rng.randint(1, 4), 16 * rng.randint(1, 4), rng.random() < 0.6))
        samples.append(sp)
    samples[0] = [("text", 3, True), ("vit", 28, 42)] + samples[0]
    samples[-1].append(("vae", 32 * (1 + seed % 2), 48 * (1 + seed % 3), True))
    batch, noise, split_lens, attn_modes = pack_training_batch(cfg, samples, seed)
    W, _ = oracle_weights(cfg)
    model = cpu_model(cfg)
    names = trainable(model)
    n_ce = batch["ce_loss_indexes"].numel() if batch.get("ce_loss_indexes") is not None else 0
    w_ce = torch.rand(n_ce, generator=torch.Generator().manual_seed(seed)) + 0.5 if n_ce else None
    rloss, rgrads, _ = O.training_step_grads(W, cfg, batch, noise, w_ce, names=set(names))
    loss, grads, _ = product_step(model, batch, noise, w_ce)
    assert abs(loss - rloss) < 2e-2 * abs(rloss)
    compare(grads, rgrads, names, 8e-2, (seed, samples))


@pytest.mark.parametrize("name", ["tiny", "tiny_d128"])
def test_text_only_pack_takes_the_single_expert_path(monkeypatch, name):
    """A pack without images (two samples of causal text splits, CE on some): no gen rows, so the engines run without expert routing and
    only the und expert, the embeddings, the norms and lm_head receive gradients -- vs the oracle's primitives under autograd."""
    from tests.util_models import text_only_training_grads
    mock_ops.install(monkeypatch)
    cfg = CFGS[name]
    batch, _, split_lens, attn_modes = pack_training_batch(cfg, [[("text", 5, True), ("text", 3, False), ("text", 4, True)], [("text", 6, True)]], 3)
    w_ce = torch.rand(batch["ce_loss_indexes"].numel(), generator=torch.Generator().manual_seed(2)) + 0.5
    W, _ = oracle_weights(cfg)
    rloss, rgrads = text_only_training_grads(W, cfg, batch, w_ce)
    model = cpu_model(cfg)
    trainable(model)
    for p in model.parameters():
        p.grad = None
    out = model(**batch)
    assert out["mse"] is None and out["ce"].requires_grad
    loss = O.training_step_loss(out, w_ce)
    loss.backward()
    assert abs(float(loss.detach()) - rloss) < 2e-2 * abs(rloss)
    grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    assert set(grads) == set(rgrads), sorted(set(grads) ^ set(rgrads))[:6]
    for n, r in rgrads.items():
        if float(r.float().norm()) > 0:
            assert rel(grads[n], r) < 6e-2, (n, rel(grads[n], r))


def test_siglip_2d_rope_variant_backward(monkeypatch):
    """The SigLIP tower with 2-D RoPE instead of the learned position table (config.rope, siglip_navit.py:102-142,224-230): the reverse of
    the rotation is the rotation by the negated angle on the q / k gradient heads -- every gradient vs the oracle's autograd."""
    mock_ops.install(monkeypatch)
    cfg = TINY_ROPE
    samples = [[("text", 3, True), ("vit", 28, 42), ("text", 4, True)], [("text", 2, False), ("vit", 42, 14), ("vae", 32, 48, True)]]
    batch, noise, _, _ = pack_training_batch(cfg, samples, 21)
    w_ce = torch.rand(batch["ce_loss_indexes"].numel(), generator=torch.Generator().manual_seed(4)) + 0.5
    W, _ = oracle_weights(cfg)
    model = cpu_model(cfg)
    names = trainable(model)
    assert not any("position_embedding" in n for n in names)
    rloss, rgrads, _ = O.training_step_grads(W, cfg, batch, noise, w_ce, names=set(names))
    loss, grads, _ = product_step(model, batch, noise, w_ce)
    assert abs(loss - rloss) < 2e-2 * abs(rloss)
    worst = compare(grads, rgrads, names, 6e-2, "tiny_rope", key_bias_is_zero=False)      # a rotated key bias is a real gradient
    assert worst[1] > 0


def test_tape_options_do_not_change_the_gradients(golden, monkeypatch):
    """KEEP_GATE_UP (the tape keeps the un-activated gate/up projection instead of recomputing it) and CACHE_WT (transposed weight images
    kept with the packed layer across micro-steps): same gradients bit for bit in every combination, also on the second micro-step (cached
    images) and after a parameter update (the images must be rebuilt with the re-packed engine)."""
    from bagel_amd.modeling.bagel import train_step as TS
    mock_ops.install(monkeypatch)
    g = golden("tiny_train")
    w_ce = torch.rand(g["ce"].shape[0], generator=torch.Generator().manual_seed(5)) + 0.5
    model = cpu_model(TINY)
    names = trainable(model)
    ref = None
    for keep, cache in ((False, False), (True, False), (False, True), (True, True)):
        monkeypatch.setattr(TS, "KEEP_GATE_UP", keep)
        monkeypatch.setattr(TS, "CACHE_WT", cache)
        for _ in range(2):
            _, grads, _ = product_step(model, g["batch"], g["noise"], w_ce)
            if ref is None:
                ref = grads
            for n in names:
                if n in ref:
                    assert torch.equal(grads[n], ref[n]), (keep, cache, n)
    # an optimizer-like in-place update: the engine re-packs, the cached images follow
    with torch.no_grad():
        for p in model.parameters():
            if p.requires_grad:
                p.mul_(1.01)
    _, g1, _ = product_step(model, g["batch"], g["noise"], w_ce)
    monkeypatch.setattr(TS, "CACHE_WT", False)
    _, g2, _ = product_step(model, g["batch"], g["noise"], w_ce)
    changed = 0
    for n in names:
        if n in g1:
            assert torch.equal(g1[n], g2[n]), n
            changed += int(not torch.equal(g1[n], ref[n]))
    assert changed > len(names) // 2


def test_frozen_and_no_grad_paths(golden, monkeypatch):
    """No tape without grad mode or without a trainable parameter; a ViT parameter that requires grad is refused; only the parameters
    that require grad receive one."""
    mock_ops.install(monkeypatch)
    g = golden("tiny_train")
    model = cpu_model(TINY)
    out = model(noise=g["noise"], **g["batch"])
    assert not out["mse"].requires_grad and not out["ce"].requires_grad
    names = trainable(model)
    with torch.no_grad():
        out = model(noise=g["noise"], **g["batch"])
    assert not out["mse"].requires_grad
    for n, p in model.named_parameters():
        p.requires_grad_(n.endswith("mlp_moe_gen.down_proj.weight") or n == "llm2vae.bias")
    out = model(noise=g["noise"], **g["batch"])
    (out["mse"].mean() + out["ce"].mean()).backward()
    got = sorted(n for n, p in model.named_parameters() if p.grad is not None)
    assert got == sorted(n for n in names if n.endswith("mlp_moe_gen.down_proj.weight") or n == "llm2vae.bias")
    model.latent_pos_embed.pos_embed.requires_grad_(True)          # frozen in the reference (a fixed sin-cos table): refused, not ignored
    with pytest.raises(NotImplementedError):
        model(noise=g["noise"], **g["batch"])


def test_attention_backward_items_cover_the_block_mask():
    """AttnBackwardPlan vs the mask of data_utils.py:72-103 rebuilt densely: the union of what the query items and the key items allow
    is exactly the mask, every row is in one query item and one key item."""
    from bagel_amd.modeling.bagel.train_step import AttnBackwardPlan
    rng = random.Random(7)
    for _ in range(20):
        sample_lens, splits = [], []
        for _ in range(rng.randint(1, 3)):
            lens = [rng.choice([1, 3, 17, 64, 130, 257]) for _ in range(rng.randint(1, 5))]
            modes = [rng.choice(["causal", "full", "noise"]) for _ in lens]
            sample_lens.append(sum(lens)); splits.append((lens, modes))
        bp = AttnBackwardPlan("cpu", sample_lens, splits)
        M = sum(sample_lens)
        dense = torch.zeros((M, M), dtype=torch.bool)
        r = 0
        for n, (lens, modes) in zip(sample_lens, splits):
            dense[r:r + n, r:r + n] = torch.isfinite(O.attention_mask_per_sample(lens, modes))
            r += n
        bits = bp.noise_bits.tolist()
        noise = torch.tensor([(bits[c // 64] >> (c % 64)) & 1 for c in range(M)], dtype=torch.bool)
        fromq, fromk = torch.zeros_like(dense), torch.zeros_like(dense)
        for row0, nrows, kstart, sstart, send, causal, t0, t1 in bp.q_items.tolist():
            rows, keys = torch.arange(row0, row0 + nrows)[:, None], torch.arange(64 * t0, min(64 * t1, M))[None, :]
            allow = (keys >= kstart) & (((keys < sstart) & ~noise[keys[0]][None]) | ((keys >= sstart) & (keys < send) & ((keys <= rows) | (causal == 0))))
            assert not fromq[row0:row0 + nrows].any()
            fromq[row0:row0 + nrows, 64 * t0:min(64 * t1, M)] = allow
        for key0, nkeys, qbeg, qend, send, causal, _, _ in bp.k_items.tolist():
            rows, keys = torch.arange(qbeg, qend)[:, None], torch.arange(key0, key0 + nkeys)[None, :]
            assert not fromk[:, key0:key0 + nkeys].any()
            fromk[qbeg:qend, key0:key0 + nkeys] = (rows >= send) | (keys <= rows) | (causal == 0)
        assert torch.equal(fromq, dense) and torch.equal(fromk, dense), (sample_lens, splits)


def test_a_few_optimizer_steps_reduce_the_loss(golden, monkeypatch):
    """The reference's loop shape -- forward, loss, backward, torch optimizer step -- on the stand-ins: every step re-packs the engines'
    weight copies from the updated parameters (modeling/packed.py) and the loss of the same batch goes down."""
    mock_ops.install(monkeypatch)
    g = golden("tiny_train")
    model = cpu_model(TINY)
    trainable(model)
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=2e-2)
    losses = []
    for _ in range(4):
        opt.zero_grad(set_to_none=True)
        out = model(noise=g["noise"], **g["batch"])
        loss = O.training_step_loss(out)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(b < a for a, b in zip(losses, losses[1:])), losses
    assert losses[-1] < 0.97 * losses[0], losses


def test_optimizer_step_refreshes_the_engine_in_place(golden, monkeypatch):
    """A training loop rewrites the parameters in place after every backward.  The packed engine must SURVIVE that (same object, same
    packed tensors and transposed images, tape pool kept on the model) with its copies refreshed -- and a step on the refreshed engine
    must equal, bit for bit, the same step on a model freshly built from the updated weights."""
    mock_ops.install(monkeypatch)
    g = golden("tiny_train")
    w_ce = torch.rand(g["ce"].shape[0], generator=torch.Generator().manual_seed(5)) + 0.5
    model = cpu_model(TINY)
    names = trainable(model)
    product_step(model, g["batch"], g["noise"], w_ce)
    lm = model.language_model
    eng = lm._engine
    ptr = eng.layers[0].wgu[0].data_ptr()
    wt_ptr = eng.layers[0].wt["wd"][0].data_ptr()
    pool = lm.model.__dict__["_tape_pool"]
    assert eng is not None and len(pool) == 1
    with torch.no_grad():                                   # plain SGD in place
        for p in model.parameters():
            if p.grad is not None:
                p.add_(p.grad, alpha=-0.05)
    loss1, g1, _ = product_step(model, g["batch"], g["noise"], w_ce)
    assert lm._engine is eng and eng.layers[0].wgu[0].data_ptr() == ptr and eng.layers[0].wt["wd"][0].data_ptr() == wt_ptr
    assert lm.model.__dict__["_tape_pool"] is pool and len(pool) == 1
    fresh = cpu_model(TINY)
    fresh.load_state_dict(model.state_dict(), strict=True)
    trainable(fresh)
    loss2, g2, _ = product_step(fresh, g["batch"], g["noise"], w_ce)
    assert loss1 == loss2
    for n in names:
        if n in g2:
            assert torch.equal(g1[n], g2[n]), n
    # re-seated storage (param.data = new tensor) is NOT an in-place rewrite: the engine is rebuilt
    p0 = lm.model.layers[0].mlp.down_proj.weight
    p0.data = p0.data.clone()
    product_step(model, g["batch"], g["noise"], w_ce)
    assert lm._engine is not eng


def test_tape_pool_is_bounded_over_changing_pack_sizes(monkeypatch):
    """The reference's packs have a different sequence_length every step: the resident tape must not pin one buffer set per length.  The
    pool holds at most TAPE_POOL_SETS sets of flat buffers that grow to the largest pack seen and serve smaller ones as views."""
    from bagel_amd.modeling.bagel.train_step import TrainTape
    mock_ops.install(monkeypatch)
    model = cpu_model(TINY)
    trainable(model)
    sizes = []
    for seed, samples in enumerate(([[("text", 3, True), ("vae", 32, 48, True)]], [[("text", 9, True), ("vit", 28, 42), ("text", 4, True)], [("text", 2, False), ("vae", 32, 32, True)]],
                                    [[("text", 5, True)]], [[("text", 3, True), ("vae", 32, 48, True)]])):
        batch, noise, _, _ = pack_training_batch(TINY, samples, seed)
        out = model(noise=noise, **batch)
        loss = out["ce"].mean() + (out["mse"].mean() if out["mse"] is not None else 0.0)
        loss.backward()
        pool = model.language_model.model.__dict__["_tape_pool"]
        assert len(pool) <= TrainTape.TAPE_POOL_SETS
        sizes.append(sum(t.numel() * t.element_size() for s in pool for t in s.values() if t is not None))
    assert sizes[2] == sizes[1] and sizes[3] == sizes[1], sizes          # smaller packs re-use the largest set's storage
    assert sizes[1] >= sizes[0]


def test_fp32_trainable_parameters_are_refused_and_master_weights_work(golden, monkeypatch):
    """Trainable fp32 parameters must not be cast in place behind the optimizer's back (round-3 advisory): the training forward refuses
    them and points at train_utils.MasterWeightOptimizer, which keeps fp32 masters + optimizer state and rewrites the bf16 compute
    parameters in place.  An update far below a bf16 ulp has to accumulate in the master and reach the parameter after enough steps --
    a bf16-only optimizer would lose it every time."""
    from bagel_amd.factory import build_bagel
    from bagel_amd.train_utils import MasterWeightOptimizer
    mock_ops.install(monkeypatch)
    g = golden("tiny_train")
    W, _ = oracle_weights(TINY)
    model, _ = build_bagel(TINY, device="cpu", with_vae=False)
    model.load_state_dict(W, strict=True)
    model = model.float()
    trainable(model)
    with pytest.raises(TypeError, match="MasterWeightOptimizer"):
        model(noise=g["noise"], **g["batch"])
    assert model.llm2vae.weight.dtype == torch.float32, "the refused call must not have cast the masters"
    with torch.no_grad(), pytest.warns(UserWarning, match="cast to bfloat16"):
        model(noise=g["noise"], **g["batch"])                      # inference-style use still casts once, with the warning
    model = model.to(torch.bfloat16)
    trainable(model)
    opt = MasterWeightOptimizer(model, lambda ps: torch.optim.SGD(ps, lr=1.0))
    p = model.llm2vae.bias
    start = p.detach().clone()
    tiny = (p.detach().float().abs() * 2.0 ** -12).clamp_min(1e-8)    # 1/16 of a bf16 ulp of each element
    for step in range(40):
        opt.zero_grad()
        for q in opt.compute:
            q.grad = None
        p.grad = (-tiny).to(p.dtype)                                  # a constant tiny push upwards
        opt.step()
        if step == 3:
            assert torch.equal(p.detach(), start), "a sub-ulp update must not be visible yet"
    moved = (p.detach().float() - start.float()) / start.float().abs().clamp_min(1e-8)
    assert (moved > 2.0 ** -9).float().mean() > 0.9, "the accumulated master update never reached the bf16 parameter"
    idx = next(i for i, q in enumerate(opt.compute) if q is p)
    assert opt.master[idx].dtype == torch.float32 and torch.allclose(opt.master[idx].float(), start.float() + 40 * tiny, rtol=1e-5)
    # and the engines follow the in-place rewrite: a step after opt.step() equals a fresh model on the same weights
    w_ce = torch.ones(g["ce"].shape[0])
    loss1, g1, _ = product_step(model, g["batch"], g["noise"], w_ce)
    fresh = cpu_model(TINY)
    fresh.load_state_dict(model.state_dict(), strict=True)
    trainable(fresh)
    loss2, g2, _ = product_step(fresh, g["batch"], g["noise"], w_ce)
    assert loss1 == loss2 and all(torch.equal(g1[n], g2[n]) for n in g2)
