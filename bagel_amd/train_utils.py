"""Mixed-precision training glue for the packed training step (``Bagel.forward`` + ``loss.backward()``).

The reference trains fp32 MASTER parameters with bf16 compute (FSDP ``MixedPrecision(param_dtype=bf16)``, train/fsdp_utils.py;
``loss.backward()`` of train/pretrain_unified_navit.py:683-735 fills fp32 gradients of fp32 masters).  The MI355X engines compute on bf16
parameters and hand back bf16 gradients; ``MasterWeightOptimizer`` supplies the other half of the recipe: fp32 master copies and fp32
optimizer state for every trainable parameter, the wrapped torch optimizer stepping on those, and the bf16 compute parameters rewritten IN
PLACE from the masters afterwards -- which is the case the engines refresh their packed copies in place for (``MoTEngine.refresh``).
Updates smaller than a bf16 ulp accumulate in the master instead of being rounded away step after step.

    model.to(torch.bfloat16)
    opt = MasterWeightOptimizer(model, lambda params: torch.optim.AdamW(params, lr=1e-4))
    for batch in loader:
        loss = weighted(model(**batch)); loss.backward(); opt.step(); opt.zero_grad()

Host-side only (torch tensors and a torch optimizer; runs on any device)."""
import torch


class MasterWeightOptimizer:
    def __init__(self, model, optimizer_factory, params=None):
        ps = list(params) if params is not None else [p for p in model.parameters() if p.requires_grad]
        self.compute = [p for p in ps if p.requires_grad]
        with torch.no_grad():
            self.master = [p.detach().to(torch.float32).clone().requires_grad_(True) for p in self.compute]
        self.optimizer = optimizer_factory(self.master)

    @property
    def param_groups(self):
        return self.optimizer.param_groups

    def zero_grad(self, set_to_none=True):
        for p in self.compute:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()
        self.optimizer.zero_grad(set_to_none=set_to_none)

    @torch.no_grad()
    def step(self, grad_scale=1.0):
        """fp32 copies of the compute gradients -> the wrapped optimizer on the masters -> compute parameters rewritten in place."""
        for p, m in zip(self.compute, self.master):
            if p.grad is None:
                m.grad = None
                continue
            g = p.grad.to(torch.float32)
            m.grad = g if grad_scale == 1.0 else g * grad_scale
        self.optimizer.step()
        for p, m in zip(self.compute, self.master):
            p.copy_(m)                       # rounds the master to the compute dtype; in place, so the engines refresh their packed copies

    def state_dict(self):
        return {"optimizer": self.optimizer.state_dict(), "master": [m.detach().clone() for m in self.master]}

    def load_state_dict(self, sd):
        self.optimizer.load_state_dict(sd["optimizer"])
        with torch.no_grad():
            for m, src, p in zip(self.master, sd["master"], self.compute):
                m.copy_(src)
                p.copy_(m)
