"""Model construction recipe of app.py:39-66 / eval/gen/gen_images_mp.py:137-176 for plain config dicts, plus
device-side random initialisation (there is no network, hence no checkpoint: benchmarks use random weights of the
named architecture)."""
import zlib

import torch

from .modeling.autoencoder import AutoEncoder, AutoEncoderParams
from .modeling.bagel import Bagel, BagelConfig, Qwen2Config, Qwen2ForCausalLM, SiglipVisionConfig, SiglipVisionModel

BAGEL_7B_MOT = dict(
    name="bagel_7b_mot",
    llm=dict(vocab_size=152064, hidden_size=3584, intermediate_size=18944, num_hidden_layers=28, num_attention_heads=28,
             num_key_value_heads=4, rope_theta=1000000.0, rms_norm_eps=1e-6, qk_norm=True,
             layer_module="Qwen2MoTDecoderLayer", tie_word_embeddings=False, max_position_embeddings=32768),
    vit=dict(hidden_size=1152, intermediate_size=4304, num_hidden_layers=26, num_attention_heads=16, num_channels=3,
             image_size=980, patch_size=14, rope=False),
    vae=dict(resolution=256, in_channels=3, downsample=8, ch=128, out_ch=3, ch_mult=[1, 2, 4, 4], num_res_blocks=2,
             z_channels=16, scale_factor=0.3611, shift_factor=0.1159),
    bagel=dict(latent_patch_size=2, max_latent_size=64, vit_max_num_patch_per_side=70, connector_act="gelu_pytorch_tanh",
               interpolate_pos=False, timestep_shift=1.0),
)
NEW_TOKEN_IDS_QWEN25 = dict(bos_token_id=151644, eos_token_id=151645, start_of_image=151652, end_of_image=151653)


def build_bagel(cfg, device="cuda", dtype=torch.bfloat16, with_vae=True, num_layers=None):
    """-> (model, vae_model): modules allocated directly on ``device`` (weights uninitialised)."""
    llm_kw = dict(cfg["llm"])
    if num_layers is not None:
        llm_kw["num_hidden_layers"] = num_layers
    llm_config = Qwen2Config(**llm_kw)
    vit_config = SiglipVisionConfig(**cfg["vit"])
    vae_params = AutoEncoderParams(**cfg["vae"])
    config = BagelConfig(visual_gen=True, visual_und=True, llm_config=llm_config, vit_config=vit_config,
                         vae_config=vae_params, **cfg["bagel"])
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        with torch.device(device):
            language_model = Qwen2ForCausalLM(llm_config)
            vit_model = SiglipVisionModel(vit_config)
            model = Bagel(language_model, vit_model, config)
            model.vit_model.vision_model.embeddings.convert_conv2d_to_linear(vit_config, meta=True)
    finally:
        torch.set_default_dtype(prev)
    model.eval()
    vae = None
    if with_vae:
        with torch.device(device):
            vae = AutoEncoder(vae_params).eval()
    return model, vae


@torch.no_grad()
def init_random_(module, seed=0, llm2vae_std=None):
    """Deterministic device-side random init: matrices ~ N(0, 1/fan_in), norm gains ~ 1 + 0.1 N, biases ~ 0.02 N,
    frozen sin-cos tables untouched.  (Values differ from oracle/weights.py's CPU stream; parity tests use that one.)"""
    for name, p in module.state_dict().items():
        if name.endswith("pos_embed"):
            continue
        g = torch.Generator(device=p.device).manual_seed((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
        if "norm" in name and name.endswith("weight") and p.dim() == 1:
            p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g, device=p.device, dtype=torch.float32))
        elif name.endswith("bias"):
            p.copy_(0.02 * torch.randn(p.shape, generator=g, device=p.device, dtype=torch.float32))
        elif p.dim() >= 2:
            fan_in = p[0].numel()
            std = fan_in ** -0.5
            if "embed_tokens" in name or "position_embedding" in name:
                std = 0.5
            # chunked to bound the fp32 temporary
            flat = p.view(p.shape[0], -1)
            step = max(1, (1 << 26) // max(flat.shape[1], 1))
            for r in range(0, flat.shape[0], step):
                blk = flat[r:r + step]
                blk.copy_(torch.randn(blk.shape, generator=g, device=p.device, dtype=torch.float32) * std)
        else:
            p.copy_(torch.randn(p.shape, generator=g, device=p.device, dtype=torch.float32))
    return module
