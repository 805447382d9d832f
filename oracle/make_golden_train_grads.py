"""Gradients of one training step from the UNMODIFIED reference, and the oracle's autograd pinned against them.

TEST INFRASTRUCTURE ONLY.  Run in the build container (needs /root/reference):

    python -m oracle.make_golden_train_grads      # writes tests/golden/<cfg>_train_grads.pt

The batch is the one of ``make_golden.scenario_train`` (tests/golden/<cfg>_train.pt: an understanding sample and a generation sample
with causal / full / noise splits).  The reference runs exactly as train/pretrain_unified_navit.py:683-735 runs it -- bf16 weights,
``torch.amp.autocast(bf16)`` around ``model(**data)``, then

    loss = (ce * w).sum() / w.sum() * ce_weight + mse.mean(-1).sum() / n_mse * mse_weight        (:705-727, one rank)

with a seeded ``ce_loss_weights`` vector (the ``ce_loss_reweighting`` branch, so that the upstream gradient differs per token) and
``loss.backward()``.  The oracle (oracle/bagel_oracle.py with GRAD_ENABLED) runs the same batch through torch's autograd; every
parameter gradient must agree with the reference's BIT FOR BIT before the fixture is written.  The fixture holds the loss weights, the
scalar loss and the reference's gradients of every parameter (ViT included: the product only builds the frozen-ViT configuration
today, ``--freeze_vit True`` of pretrain_unified_navit.py:386-389, and checks the rest)."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import bagel_oracle as O                      # noqa: E402
from oracle.configs import TINY, TINY_D128                # noqa: E402
from oracle.make_golden import GOLD, build                # noqa: E402

def step_loss(out, w_ce):
    return O.training_step_loss(out, w_ce)


def reference_grads(model, batch, w_ce):
    import contextlib
    import modeling.bagel.qwen2_navit as qn
    qn.sdpa_kernel = lambda *a, **k: contextlib.nullcontext()       # the SDPA backend pin of :468 is CUDA-only
    for p in model.parameters():
        p.requires_grad_(True)
        p.grad = None
    model.train()
    try:
        with torch.autocast("cpu", dtype=torch.bfloat16):
            torch.manual_seed(47)
            out = model(**batch)
        loss = step_loss(out, w_ce)
        loss.backward()
    finally:
        model.eval()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    for p in model.parameters():
        p.grad = None
        p.requires_grad_(False)
    return float(loss.detach()), grads, {k: v.detach() for k, v in out.items()}


def check_pair(cfg, model, W, batch, noise, w_ce):
    """Reference backward vs oracle autograd on one batch: -> (loss, reference grads); raises unless every gradient is bit-exact."""
    loss, grads, out = reference_grads(model, batch, w_ce)
    # the losses with and without a graph are the same numbers
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        model.train(); torch.manual_seed(47); plain = model(**batch); model.eval()
    assert torch.equal(out["ce"], plain["ce"]) and torch.equal(out["mse"], plain["mse"]), "the graph changed the forward"
    oloss, ograds, _ = O.training_step_grads(W, cfg, batch, noise, w_ce, names=set(grads))
    bad = [k for k, gr in grads.items() if k not in ograds or not torch.equal(gr, ograds[k])]
    assert loss == oloss and not bad, f"oracle autograd != reference autograd: loss {loss} vs {oloss}, {bad[:5]}"
    return loss, grads


def main():
    for cfg in (TINY, TINY_D128):
        model, vae, W, VW = build(cfg)
        fx = torch.load(os.path.join(GOLD, f"{cfg['name']}_train.pt"), weights_only=False)
        batch, noise = fx["batch"], fx["noise"]
        g = torch.Generator().manual_seed(5)
        w_ce = torch.rand(fx["ce"].shape[0], generator=g) + 0.5
        loss, grads = check_pair(cfg, model, W, batch, noise, w_ce)
        print(f"[{cfg['name']}] loss {loss:.6f}: {len(grads)} parameter gradients, oracle autograd == reference autograd bit-exact")
        if cfg is not TINY:
            continue            # the wider fixture would be 26 MB; tests/test_reference_crosscheck.py re-runs this check live instead
        path = os.path.join(GOLD, f"{cfg['name']}_train_grads.pt")
        torch.save(dict(ce_loss_weights=w_ce, loss=loss, grads=grads), path)
        print(f"[golden] {path}  ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
