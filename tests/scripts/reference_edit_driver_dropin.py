"""Run by tests/test_reference_crosscheck.py in a fresh interpreter: the UNMODIFIED image-edit batch driver of the reference
(/root/reference/eval/gen/gen_images_mp_imgedit.py, loaded by path; ``__main__`` does not run) resolves its imports through
``bagel_amd.install_as_reference()``, and its ``editing_image()`` -- VAE + ViT prefill of a PIL image through the ImageTransforms,
the three CFG contexts (deepcopy of the cache), the 3-forward sampler with text_channel renorm, un-patchify, VAE decode, uint8 --
drives the PRODUCT end to end (BASELINE configs[4]; host logic on the torch stand-ins of tests/mock_ops.py, no GPU here).
The image must match the oracle's restatement of the same pipeline on the same random draws."""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bagel_amd  # noqa: E402

bagel_amd.install_as_reference()
from tests import mock_ops  # noqa: E402


class _MP:
    def setattr(self, o, n, v):
        setattr(o, n, v)


mock_ops.install(_MP())
spec = importlib.util.spec_from_file_location("reference_gen_images_mp_imgedit", "/root/reference/eval/gen/gen_images_mp_imgedit.py")
R = importlib.util.module_from_spec(spec)
spec.loader.exec_module(R)

from PIL import Image  # noqa: E402
from bagel_amd.data.transforms import ImageTransform  # noqa: E402
from oracle import bagel_oracle as O  # noqa: E402
from oracle import packers as P  # noqa: E402
from oracle.configs import TINY as cfg, NEW_TOKEN_IDS_TINY as ids, StubTokenizer  # noqa: E402
from tests.test_host_logic_cpu import cpu_model_and_vae  # noqa: E402
from tests.util_models import oracle_weights  # noqa: E402

model, vae = cpu_model_and_vae(cfg)
tok = StubTokenizer(cfg["llm"]["vocab_size"])
vae_tf, vit_tf = ImageTransform(64, 32, 16, device="cpu"), ImageTransform(56, 28, 14, device="cpu")
R.gen_model = R.model = model
R.vae_model, R.tokenizer, R.new_token_ids, R.vae_transform, R.vit_transform = vae, tok, ids, vae_tf, vit_tf
src = Image.fromarray(np.random.default_rng(9).integers(0, 256, (60, 80, 3), dtype=np.uint8), "RGB")
prompt = "make it blue"
kw = dict(num_timesteps=4, cfg_text_scale=4.0, cfg_img_scale=2.0, cfg_interval=[0, 1.0], cfg_renorm_min=0.0, timestep_shift=3.0)
torch.manual_seed(51)
img = R.editing_image(images=[src], prompt=prompt, max_image_size=64, min_image_size=32, device="cpu", **kw)
assert img.size == (64, 48), img.size

# the same pipeline through the oracle, same random draws in the same order (VAE posterior noise, then the init latents)
W, VW = oracle_weights(cfg)
L = cfg["llm"]["num_hidden_layers"]
ident = lambda t: t  # noqa: E731
x_vae, x_vit = vae_tf(src), vit_tf(src)
torch.manual_seed(51)
enc_noise = torch.randn(1, cfg["vae"]["z_channels"], x_vae.shape[1] // 8, x_vae.shape[2] // 8)
vi, l1, r1 = P.prepare_vae_images([0], [0], [x_vae], ident, ids, 16, cfg["bagel"]["max_latent_size"])
oc = O.forward_cache_update_vae(W, cfg, VW, O.OracleCache(L), sample_noise=enc_noise, **vi)
ti, l2, r2 = P.prepare_vit_images(l1, r1, [x_vit], ident, ids, cfg["vit"]["patch_size"], cfg["bagel"]["vit_max_num_patch_per_side"])
oc = O.forward_cache_update_vit(W, cfg, oc, **ti)
octext = oc.clone()
pi2, l4, r4 = P.prepare_prompts([0], [0], [prompt], tok, ids)
ocimg = O.forward_cache_update_text(W, cfg, O.OracleCache(L), **pi2)
pi, l3, r3 = P.prepare_prompts(l2, r2, [prompt], tok, ids)
oc = O.forward_cache_update_text(W, cfg, oc, **pi)
h, w = 48, 64
li = P.prepare_vae_latent(l3, r3, [(h, w)], ids, 16, cfg["bagel"]["max_latent_size"], 64)
ct, cim = P.prepare_vae_latent_cfg(l2, r2, [(h, w)], 16), P.prepare_vae_latent_cfg(l4, r4, [(h, w)], 16)


def od(c, d):
    return dict(cache=c, position_ids=d["cfg_packed_position_ids"], query_indexes=d["cfg_packed_query_indexes"],
                key_values_lens=d["cfg_key_values_lens"], key_value_indexes=d["cfg_packed_key_value_indexes"])


lat = O.generate_image(W, cfg, li, oc, cfg_text=od(octext, ct), cfg_img=od(ocimg, cim), num_timesteps=4, timestep_shift=3.0,
                       cfg_renorm_min=0.0, cfg_renorm_type="text_channel", cfg_interval=[0, 1.0], cfg_text_scale=4.0, cfg_img_scale=2.0)
ref = O.latent_to_image_uint8(VW, cfg["vae"], lat[0], h, w, 16, 2, 16)
d = np.abs(np.asarray(img).astype(np.int32) - ref.numpy().astype(np.int32))
assert d.shape == (h, w, 3) and d.mean() <= 3.0 and np.percentile(d, 99) <= 14, (d.mean(), np.percentile(d, 99))
print("ok")
