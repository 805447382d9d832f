"""Run by tests/test_reference_crosscheck.py in a fresh interpreter: the UNMODIFIED batch driver of the reference
(/root/reference/eval/gen/gen_images_mp.py, loaded by path -- its ``__main__`` block does not run) resolves its imports through
``bagel_amd.install_as_reference()``, and its ``generate_image()`` -- prompt prefill, CFG latent preparation, the sampler, the
latent un-patchify einsum, VAE decode, uint8 conversion -- drives the PRODUCT's model and VAE (host logic on the torch stand-ins of
tests/mock_ops.py; no GPU here).  The images must match the oracle's restatement of the same pipeline."""
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
spec = importlib.util.spec_from_file_location("reference_gen_images_mp", "/root/reference/eval/gen/gen_images_mp.py")
R = importlib.util.module_from_spec(spec)
spec.loader.exec_module(R)

from oracle import bagel_oracle as O  # noqa: E402
from oracle import packers as P  # noqa: E402
from oracle.configs import TINY as cfg, NEW_TOKEN_IDS_TINY, StubTokenizer  # noqa: E402
from tests.test_host_logic_cpu import cpu_model_and_vae  # noqa: E402
from tests.util_models import oracle_weights  # noqa: E402

from bagel_amd.factory import build_bagel  # noqa: E402

_, vae = cpu_model_and_vae(cfg)
# fp32 master weights, exactly as gen_images_mp.py:174 leaves them (`model.to(device)`, no dtype): the product casts them once
model, _ = build_bagel(cfg, device="cpu", dtype=torch.float32, with_vae=False)
model.load_state_dict(oracle_weights(cfg)[0], strict=True)
model = model.eval()
tok = StubTokenizer(cfg["llm"]["vocab_size"])
R.gen_model = R.model = model          # the module-level names its generate_image() reads (gen_images_mp.py:137-176)
R.vae_model, R.tokenizer, R.new_token_ids = vae, tok, NEW_TOKEN_IDS_TINY
prompt, n, res = "a small red cube", 2, 64
kw = dict(num_timesteps=4, cfg_scale=4.0, cfg_interval=[0, 1.0], cfg_renorm_min=0.0, timestep_shift=3.0)
torch.manual_seed(42)
images = R.generate_image(prompt=prompt, num_images=n, resolution=res, device="cpu", **kw)
assert len(images) == n and images[0].size == (res, res)

# the same pipeline through the oracle
W, VW = oracle_weights(cfg)
L = cfg["llm"]["num_hidden_layers"]
gi, lens, ropes = P.prepare_prompts([0] * n, [0] * n, [prompt] * n, tok, NEW_TOKEN_IDS_TINY)
cache = O.forward_cache_update_text(W, cfg, O.OracleCache(L), **gi)
torch.manual_seed(42)
li = P.prepare_vae_latent(lens, ropes, [(res, res)] * n, NEW_TOKEN_IDS_TINY, 16, cfg["bagel"]["max_latent_size"], 64)
ci = P.prepare_vae_latent_cfg([0] * n, [0] * n, [(res, res)] * n, 16)
ocfg = dict(cache=O.OracleCache(L), position_ids=ci["cfg_packed_position_ids"], query_indexes=ci["cfg_packed_query_indexes"],
            key_values_lens=ci["cfg_key_values_lens"], key_value_indexes=ci["cfg_packed_key_value_indexes"])
lat = O.generate_image(W, cfg, li, cache, cfg_text=ocfg, num_timesteps=4, timestep_shift=3.0, cfg_renorm_min=0.0,
                       cfg_renorm_type="global", cfg_interval=[0, 1.0], cfg_text_scale=4.0)
for img, l_ in zip(images, lat):
    ref = O.latent_to_image_uint8(VW, cfg["vae"], l_, res, res, 16, 2, 16)
    d = np.abs(np.asarray(img).astype(np.int32) - ref.numpy().astype(np.int32))
    assert d.shape == (res, res, 3) and d.mean() <= 1.5 and np.percentile(d, 99) <= 8, (d.mean(), np.percentile(d, 99))
print("ok")
