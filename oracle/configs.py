"""Model configurations used by the oracle, the golden generator and the tests.

TEST INFRASTRUCTURE ONLY.  ``TINY`` is BASELINE.json configs[0] ("Tiny random-init BAGEL, 2-layer
MoT, 128-d"); ``TINY_D128`` is the same plumbing with the flagship head_dim (128) and GQA group 2 so
the D=128 attention/QK-norm code path is pinned against the reference as well.
Hyper-parameter names are the reference's own (qwen2_navit.py:152-204, siglip_navit.py:71-99,
autoencoder.py:20-31, bagel.py:27-54).
"""

_VAE_TINY = dict(resolution=64, in_channels=3, downsample=8, ch=32, out_ch=3, ch_mult=[1, 2, 4, 4],
                 num_res_blocks=1, z_channels=16, scale_factor=0.3611, shift_factor=0.1159)

TINY = dict(
    name="tiny",
    llm=dict(vocab_size=512, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
             num_attention_heads=4, num_key_value_heads=2, rope_theta=1000000.0, rms_norm_eps=1e-6,
             qk_norm=True, layer_module="Qwen2MoTDecoderLayer", tie_word_embeddings=False,
             max_position_embeddings=32768),
    vit=dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
             num_channels=3, image_size=140, patch_size=14, rope=False),
    vae=_VAE_TINY,
    bagel=dict(latent_patch_size=2, max_latent_size=64, vit_max_num_patch_per_side=10,
               connector_act="gelu_pytorch_tanh", interpolate_pos=False, timestep_shift=1.0),
    llm2vae_std=0.5,
)

TINY_D128 = dict(
    name="tiny_d128",
    llm=dict(vocab_size=512, hidden_size=512, intermediate_size=1024, num_hidden_layers=2,
             num_attention_heads=4, num_key_value_heads=2, rope_theta=1000000.0, rms_norm_eps=1e-6,
             qk_norm=True, layer_module="Qwen2MoTDecoderLayer", tie_word_embeddings=False,
             max_position_embeddings=32768),
    vit=dict(hidden_size=144, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
             num_channels=3, image_size=140, patch_size=14, rope=False),   # head_dim 72 like so400m
    vae=_VAE_TINY,
    bagel=dict(latent_patch_size=2, max_latent_size=64, vit_max_num_patch_per_side=10,
               connector_act="gelu_pytorch_tanh", interpolate_pos=False, timestep_shift=1.0),
    llm2vae_std=0.25,
)

# the SigLIP 2-D RoPE variant (siglip_navit.py:102-142,224-230; switched off for BAGEL-7B by app.py:45): same plumbing as TINY
TINY_ROPE = dict(TINY, name="tiny_rope", vit=dict(TINY["vit"], rope=True))

# the alternates of Decoder_layer_dict (qwen2_navit.py:936-940): dense layers, and shared attention + per-modality MLP
TINY_DENSE = dict(TINY, name="tiny_dense", llm=dict(TINY["llm"], layer_module="Qwen2DecoderLayer"))
TINY_MOE = dict(TINY, name="tiny_moe", llm=dict(TINY["llm"], layer_module="Qwen2MoEDecoderLayer"))

# BAGEL-7B-MoT (public checkpoint config). Random-init in benchmarks.
BAGEL_7B = dict(
    name="bagel_7b_mot",
    llm=dict(vocab_size=152064, hidden_size=3584, intermediate_size=18944, num_hidden_layers=28,
             num_attention_heads=28, num_key_value_heads=4, rope_theta=1000000.0, rms_norm_eps=1e-6,
             qk_norm=True, layer_module="Qwen2MoTDecoderLayer", tie_word_embeddings=False,
             max_position_embeddings=32768),
    vit=dict(hidden_size=1152, intermediate_size=4304, num_hidden_layers=26, num_attention_heads=16,
             num_channels=3, image_size=980, patch_size=14, rope=False),
    vae=dict(resolution=256, in_channels=3, downsample=8, ch=128, out_ch=3, ch_mult=[1, 2, 4, 4],
             num_res_blocks=2, z_channels=16, scale_factor=0.3611, shift_factor=0.1159),
    bagel=dict(latent_patch_size=2, max_latent_size=64, vit_max_num_patch_per_side=70,
               connector_act="gelu_pytorch_tanh", interpolate_pos=False, timestep_shift=1.0),
    llm2vae_std=0.02,
)

NEW_TOKEN_IDS_TINY = dict(bos_token_id=1, eos_token_id=2, start_of_image=3, end_of_image=4)
NEW_TOKEN_IDS_7B = dict(bos_token_id=151644, eos_token_id=151645, start_of_image=151652, end_of_image=151653)


class StubTokenizer:
    """Deterministic stand-in for Qwen2Tokenizer (no vocab files exist offline).

    encode: one id per character, ids in [8, vocab); decode: inverse with the two chat markers the
    reference splits on (inferencer.py:203-204)."""

    def __init__(self, vocab_size=512):
        self.vocab_size = vocab_size

    def encode(self, s):
        return [8 + (ord(c) * 7 + i * 3) % (self.vocab_size - 8) for i, c in enumerate(s)]

    def decode(self, ids):
        out = []
        for t in [int(x) for x in ids]:
            if t == NEW_TOKEN_IDS_TINY["bos_token_id"]:
                out.append("<|im_start|>")
            elif t == NEW_TOKEN_IDS_TINY["eos_token_id"]:
                out.append("<|im_end|>")
            else:
                out.append(f"[{t}]")
        return "".join(out)

# BAGEL-7B-MoT WIDTH (hidden 3584, intermediate 18944, 28 / 4 heads of 128, MoT) at 2 layers and a 512-token vocabulary: the shapes
# the benchmark's kernels actually run (M = 4098 rows per 1024^2 sample, K = 3584 / 18944, GQA 7) at a size the unmodified
# reference finishes on 8 CPU cores in about a minute -- tests/golden/wide7b_t2i.pt (oracle/make_golden_wide.py).
WIDE7B = dict(
    name="wide7b",
    llm=dict(BAGEL_7B["llm"], vocab_size=512, num_hidden_layers=2),
    vit=TINY["vit"], vae=_VAE_TINY, bagel=TINY["bagel"], llm2vae_std=0.02,
)

# BASELINE.json configs[1] (image understanding) at the shapes it actually runs, cut down in DEPTH only: SigLIP so400m WIDTH (1152-d,
# 16 heads x 72, MLP 4304, patch 14, 70 x 70 learned position table; siglip_navit.py:145-245) at 2 layers on a 980 x 980 image
# (4900 patches) + the 7B-width LLM of WIDE7B at 2 layers -- tests/golden/wide7b_und.pt (oracle/make_golden_wide_und.py).
WIDE7B_UND = dict(
    name="wide7b_und",
    llm=WIDE7B["llm"],
    vit=dict(BAGEL_7B["vit"], num_hidden_layers=2),
    vae=_VAE_TINY, bagel=dict(BAGEL_7B["bagel"]), llm2vae_std=0.02,
)

# the real VAE (autoencoder.py:340-351) on its own -- tests/golden/vae_full.pt
VAE_FULL = dict(name="vae_full", vae=BAGEL_7B["vae"])
