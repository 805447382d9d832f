"""Deterministic synthetic weights keyed by state-dict name.

TEST INFRASTRUCTURE ONLY.  There is no network, hence no checkpoint: every parity test and golden
fixture uses weights that can be regenerated bit-identically from (name, shape, seed) on any box,
so fixtures only have to carry inputs and outputs.  Key names are the reference's state-dict keys
(SURVEY.md Appendix B); the frozen sin-cos tables (``*.pos_embed``) are left as constructed
(modeling_utils.py:138-141).

Scaling is chosen so random-init activations are O(1) and attention logits are O(1) (a real
checkpoint's regime): matrices ~ N(0, 1/fan_in); norm gains ~ 1 + 0.1 N(0,1); biases ~ 0.02..0.05 N.
"""
import zlib

import torch


def _gen(name: str, seed: int) -> torch.Generator:
    return torch.Generator(device="cpu").manual_seed((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def synth_tensor(name: str, shape, seed: int = 0, dtype=torch.float32) -> torch.Tensor:
    g = _gen(name, seed)
    shape = tuple(shape)
    if name.endswith("pos_embed.pos_embed") or name.endswith("pos_embed"):
        raise ValueError("frozen sin-cos tables are not synthesised")
    is_norm = ("norm" in name) and name.endswith("weight") and len(shape) == 1
    if is_norm:
        t = 1.0 + 0.1 * torch.randn(shape, generator=g)
    elif name.endswith("bias"):
        scale = 0.05 if "language_model" in name else 0.02
        t = scale * torch.randn(shape, generator=g)
    elif "embed_tokens" in name or "position_embedding" in name:
        t = 0.5 * torch.randn(shape, generator=g)
    elif len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        t = torch.randn(shape, generator=g) * (fan_in ** -0.5)
    else:
        t = torch.randn(shape, generator=g)
    return t.to(dtype)


def synth_state_dict(shapes: dict, seed: int = 0, dtype=torch.float32) -> dict:
    """shapes: {state-dict key: shape}. Keys ending in 'pos_embed' are skipped (frozen tables)."""
    out = {}
    for name in sorted(shapes):
        if name.endswith("pos_embed") or ".rope." in name:      # frozen tables (sin-cos position tables, SigLIP 2-D RoPE)
            continue
        out[name] = synth_tensor(name, shapes[name], seed, dtype)
    return out


def load_synth(module: torch.nn.Module, seed: int = 0):
    """Fill ``module`` in place (keeps its dtype/device); returns the module."""
    sd = module.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd.items()}
    new = synth_state_dict(shapes, seed)
    with torch.no_grad():
        for k, v in new.items():
            sd[k].copy_(v.to(sd[k].dtype))
    return module
