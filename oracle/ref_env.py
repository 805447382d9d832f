"""Import scaffolding for running the UNMODIFIED reference (/root/reference) on CPU.

TEST INFRASTRUCTURE ONLY.  Works only in the build container (the reference tree does not travel
to the GPU box); used by ``oracle/make_golden.py`` to produce the fixtures under ``tests/golden/``
and by the CPU-only cross-check tests that are skipped when ``/root/reference`` is absent.

What it does:
  1. puts ``oracle/_shims`` (pure-torch ``flash_attn`` stand-in) and ``/root/reference`` on sys.path;
  2. transformers-5.x compat for code pinned to 4.49: re-register the 'default' RoPE init
     (modeling_qwen2.py:105 looks it up in ROPE_INIT_FUNCTIONS);
  3. exposes ``build_reference_model`` which mirrors the construction recipe of app.py:39-66 /
     eval/gen/gen_images_mp.py:137-176 for a given tiny config.
"""
import os
import sys

REFERENCE_ROOT = os.environ.get("BAGEL_REFERENCE_ROOT", "/root/reference")
_SHIMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_shims")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "modeling", "bagel"))


def activate():
    """Make ``import modeling.bagel`` resolve to the reference tree. Idempotent."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    for p in (REFERENCE_ROOT, _SHIMS):
        if p in sys.path:
            sys.path.remove(p)
    sys.path[:0] = [_SHIMS, REFERENCE_ROOT]
    import torch
    from transformers import modeling_rope_utils as mru

    def _default_rope(config=None, device=None, seq_len=None, **kw):
        D = getattr(config, "head_dim", None) or config.hidden_size // config.num_attention_heads
        inv = 1.0 / (config.rope_theta ** (torch.arange(0, D, 2, dtype=torch.int64).float().to(device) / D))
        return inv, 1.0

    mru.ROPE_INIT_FUNCTIONS.setdefault("default", _default_rope)


def build_reference_model(cfg: dict, seed: int = 0):
    """Reference Bagel + AutoEncoder for a tiny config dict (see oracle/configs.py)."""
    activate()
    import torch
    from modeling.bagel import (BagelConfig, Bagel, Qwen2Config, Qwen2ForCausalLM,
                                SiglipVisionConfig, SiglipVisionModel)
    from modeling.autoencoder import AutoEncoder, AutoEncoderParams

    torch.manual_seed(seed)
    llm_config = Qwen2Config(pad_token_id=None, **cfg["llm"])
    vit_config = SiglipVisionConfig(**cfg["vit"])
    vae_params = AutoEncoderParams(**cfg["vae"])
    bagel_config = BagelConfig(visual_gen=True, visual_und=True, llm_config=llm_config,
                               vit_config=vit_config, vae_config=vae_params, **cfg["bagel"])
    lm = Qwen2ForCausalLM(llm_config)
    vit = SiglipVisionModel(vit_config)
    model = Bagel(lm, vit, bagel_config)
    model.vit_model.vision_model.embeddings.convert_conv2d_to_linear(vit_config)
    vae = AutoEncoder(vae_params)
    # llm2vae is zero-initialised (bagel.py:96-99): re-init or every flow test is vacuous.
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        model.llm2vae.weight.copy_(torch.randn(model.llm2vae.weight.shape, generator=g) * cfg["llm2vae_std"])
        model.llm2vae.bias.copy_(torch.randn(model.llm2vae.bias.shape, generator=g) * 0.02)
        # de-correlate the gen expert from the und expert and make norm weights non-trivial
        for name, p in model.named_parameters():
            if "norm" in name and p.ndim == 1 and "layer_norm" not in name and "post_layernorm" not in name:
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif p.ndim == 1 and name.endswith("bias") and "language_model" in name:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    model = model.to(torch.bfloat16).eval()
    vae = vae.eval()
    return model, vae
