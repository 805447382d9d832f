"""GPU: the InterleaveInferencer mirror (bagel_amd/inferencer.py) end to end against the REFERENCE's own inferencer outputs
(tests/golden/*_inferencer.pt, produced by oracle/make_golden.py from inferencer.py:22-313 with the reference's ImageTransform):
text -> image, image + text -> edited image, image + text -> text.  PIL in, PIL / str out, same seeds (the packers and the VAE
draw their noise from the host generator in the reference's order).

Tolerance for the uint8 images: the latents carry the bf16 accumulation-order noise of tests/test_model_gpu.py (rel-L2 <= 2-3e-2)
through the fp32 VAE: text -> image mean |diff| <= 1.5 grey levels, 99 % of the pixels within 8 (measured 0.5 / 2); the edit
path (VAE encode + ViT + 3 forwards per step, latents <= 3e-2) mean <= 3, p99 <= 14 (measured 1.6 / 7)."""
import re

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inferencer(cfg, vae_precision="fp32"):
    from bagel_amd.data.transforms import ImageTransform
    from bagel_amd.inferencer import InterleaveInferencer
    from oracle.configs import NEW_TOKEN_IDS_TINY, StubTokenizer
    from tests.util_models import product_model
    model, vae = product_model(cfg)
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    inf = InterleaveInferencer(model, vae, tok, ImageTransform(64, 32, 16), ImageTransform(56, 28, 14), NEW_TOKEN_IDS_TINY)
    # the *_inferencer.pt goldens were produced with the VAE held in fp32 OUTSIDE the reference's autocast region (oracle/make_golden.py
    # _Fp32Vae); the bf16-autocast VAE (the product's default inside interleave_inference) has its own golden and tests below
    inf.vae_precision_in_autocast = vae_precision
    return inf


def _compare(img, ref, what, mean_tol, p99_tol):
    a = np.asarray(img).astype(np.int32)
    b = ref.numpy().astype(np.int32)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    d = np.abs(a - b)
    mean, p99 = float(d.mean()), float(np.percentile(d, 99))
    print(f"{what}: mean |diff| {mean:.3f}, p99 {p99:.1f}, max {int(d.max())}")
    assert mean <= mean_tol and p99 <= p99_tol, f"{what}: image differs from the reference's (mean {mean:.2f}, p99 {p99:.1f})"


@pytest.mark.parametrize("name", ["tiny", "tiny_d128"])
def test_interleave_inferencer_matches_reference(golden, name):
    from PIL import Image
    from oracle.configs import TINY, TINY_D128
    cfg = {"tiny": TINY, "tiny_d128": TINY_D128}[name]
    g = golden(f"{name}_inferencer")
    inf = _inferencer(cfg)
    src = Image.fromarray(g["source_image"].numpy(), "RGB")
    torch.manual_seed(g["t2i"]["seed"])
    r = inf(text=g["t2i"]["text"], **g["t2i"]["kwargs"])
    assert isinstance(r["image"], Image.Image) and r["text"] is None
    _compare(r["image"], g["t2i"]["image"], "text -> image", 1.5, 8)
    torch.manual_seed(g["edit"]["seed"])
    r = inf(image=src, text=g["edit"]["text"], **g["edit"]["kwargs"])
    _compare(r["image"], g["edit"]["image"], "image + text -> image", 3.0, 14)
    r = inf(image=src, text=g["understanding"]["text"], **g["understanding"]["kwargs"])
    ours, ref = re.findall(r"\[(\d+)\]", r["text"]), re.findall(r"\[(\d+)\]", g["understanding"]["answer"])
    assert r["image"] is None and len(ours) == len(ref) and ours[0] == ref[0], (r["text"], g["understanding"]["answer"])
    # greedy ids may part ways at a near tie (random-init logits; the tie rule itself is tested in test_model_gpu.py): whatever
    # follows the first difference is incomparable, everything before it must be equal
    first = next((i for i, (a, b) in enumerate(zip(ours, ref)) if a != b), len(ref))
    assert ours[:first] == ref[:first]
    print(f"understanding: {first}/{len(ref)} leading tokens equal to the reference's")


def test_reference_construction_recipe_through_the_aliases(golden):
    """The construction recipe of app.py:39-66,137-145 written against the REFERENCE's module names (``from modeling.bagel import
    ...``), run in a fresh interpreter after ``bagel_amd.install_as_reference()``: modules built on the CPU in fp32, Conv2d patch
    embedding converted to Linear, ``load_state_dict(strict=False)``, ``.to('cuda', bfloat16).eval()``, InterleaveInferencer call.
    The produced image must match the reference inferencer's golden like the factory-built model's does."""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(tempfile.mkdtemp(), "img.pt")
    code = f"""
import sys, torch, numpy as np
sys.path.insert(0, {root!r})
import bagel_amd; bagel_amd.install_as_reference()
from data.transforms import ImageTransform
from inferencer import InterleaveInferencer
from modeling.autoencoder import AutoEncoder, AutoEncoderParams
from modeling.bagel import BagelConfig, Bagel, Qwen2Config, Qwen2ForCausalLM, SiglipVisionConfig, SiglipVisionModel
from oracle.configs import TINY as cfg, NEW_TOKEN_IDS_TINY, StubTokenizer
from tests.util_models import oracle_weights
W, VW = oracle_weights(cfg)
llm_config = Qwen2Config(**cfg["llm"])
vit_config = SiglipVisionConfig(**cfg["vit"])
vae_config = AutoEncoderParams(**cfg["vae"])
vae_model = AutoEncoder(vae_config)
vae_model.load_state_dict(VW, strict=True)
config = BagelConfig(visual_gen=True, visual_und=True, llm_config=llm_config, vit_config=vit_config, vae_config=vae_config, **cfg["bagel"])
language_model = Qwen2ForCausalLM(llm_config)
vit_model = SiglipVisionModel(vit_config)
model = Bagel(language_model, vit_model, config)
model.vit_model.vision_model.embeddings.convert_conv2d_to_linear(vit_config)
missing, unexpected = model.load_state_dict(W, strict=False)
assert not unexpected, unexpected
model = model.to("cuda", torch.bfloat16).eval()
vae_model = vae_model.to("cuda").eval()
inf = InterleaveInferencer(model, vae_model, StubTokenizer(cfg["llm"]["vocab_size"]), ImageTransform(64, 32, 16), ImageTransform(56, 28, 14), NEW_TOKEN_IDS_TINY)
g = torch.load({os.path.join(root, 'tests', 'golden', 'tiny_inferencer.pt')!r}, weights_only=False)
torch.manual_seed(g["t2i"]["seed"])
r = inf(text=g["t2i"]["text"], **g["t2i"]["kwargs"])
torch.save(torch.from_numpy(np.array(r["image"])), {out!r})
print("ok")
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-3000:]
    img = torch.load(out)
    _compare(img.numpy(), golden("tiny_inferencer")["t2i"]["image"], "reference recipe, text -> image", 1.5, 8)


def test_chat_entry_point_matches_inferencer_understanding(golden):
    """Bagel.chat (bagel.py:1004-1074, the eval/vlm entry): ViT prefill per image -> text prefill -> greedy decode; the same
    request through InterleaveInferencer(understanding_output=True) must give the same answer (and both match the reference's)."""
    from PIL import Image
    from oracle.configs import TINY, NEW_TOKEN_IDS_TINY, StubTokenizer
    g = golden("tiny_inferencer")
    inf = _inferencer(TINY)
    src = Image.fromarray(g["source_image"].numpy(), "RGB")
    want = inf(image=src, text=g["understanding"]["text"], **g["understanding"]["kwargs"])["text"]
    tok = StubTokenizer(TINY["llm"]["vocab_size"])
    # chat feeds the resized image through the ViT transform itself (bagel.py:1022-1035)
    got = inf.model.chat(tok, NEW_TOKEN_IDS_TINY, inf.vit_transform, [inf.vae_transform.resize_transform(src)], g["understanding"]["text"],
                         max_length=g["understanding"]["kwargs"]["max_think_token_n"])
    assert got == want == g["understanding"]["answer"], (got, want, g["understanding"]["answer"])


@pytest.mark.parametrize("name", ["tiny", "tiny_d128"])
def test_think_then_generate_flow(golden, name):
    """think=True (inferencer.py:236-266): system prompt -> greedy planning text -> the text is fed back as context -> image.  The
    planning text must match the reference's up to a greedy near-tie; when it matches entirely the image is comparable too."""
    from oracle.configs import TINY, TINY_D128
    cfg = {"tiny": TINY, "tiny_d128": TINY_D128}[name]
    g = golden(f"{name}_inferencer")["think"]
    inf = _inferencer(cfg)
    torch.manual_seed(g["seed"])
    r = inf(text=g["text"], **g["kwargs"])
    ours, ref = re.findall(r"\[(\d+)\]", r["text"]), re.findall(r"\[(\d+)\]", g["thought"])
    assert len(ours) == len(ref) and ours[0] == ref[0], (r["text"], g["thought"])
    assert r["image"].size == (g["image"].shape[1], g["image"].shape[0])
    if ours == ref:
        _compare(r["image"], g["image"], "think -> image", 1.5, 8)
    else:
        print(f"planning text diverged at a near tie ({r['text']} vs {g['thought']}): image not comparable")


def test_inferencer_runs_the_vae_in_bf16_inside_its_autocast_region(golden):
    """inferencer.py:233 wraps interleave_inference in torch.autocast(bfloat16): the reference's VAE encode / decode run with bf16 convolutions
    there.  The product mirrors that by default (InterleaveInferencer.vae_precision_in_autocast = "bf16": csrc/vae.hip bagel_conv_gemm_bf16);
    "fp32" keeps the VAE as scripts outside an autocast region run it.  (a) an edit request with the default runs BOTH the encode and the
    decode on the bf16 engine and never touches the fp32 one; the override does not outlive the call; (b) the SAME latents decoded at the
    two precisions give the same picture up to the bf16 VAE's own noise (~1e-2 rel-L2, tests/golden/vae_full_bf16.pt: about a grey
    level) -- the whole request is not compared across precisions: a bf16-encoded source image moves the context, and CFG 4 / 2 on a
    random-init model amplifies that into a different sample (measured: mean |diff| 8.6 grey levels)."""
    from PIL import Image
    from oracle.configs import TINY_D128
    g = golden("tiny_d128_inferencer")
    inf = _inferencer(TINY_D128, "bf16")
    inf.vae_model.invalidate_packed()
    torch.manual_seed(g["edit"]["seed"])
    r = inf(image=Image.fromarray(g["source_image"].numpy(), "RGB"), text=g["edit"]["text"], **g["edit"]["kwargs"])
    assert isinstance(r["image"], Image.Image) and r["image"].size == (g["edit"]["image"].shape[1], g["edit"]["image"].shape[0])
    assert inf._vae_precision is None, "the precision override must not outlive interleave_inference"
    assert inf.vae_model._engine_bf16 is not None and inf.vae_model._engine is None, "encode and decode of the request must run on the bf16 engine"
    lat = torch.randn(12, 64, generator=torch.Generator().manual_seed(5)) * 0.8          # a 4 x 3 latent grid -> 64 x 48 image
    imgs = {}
    for prec in ("bf16", "fp32"):
        inf._vae_precision = prec
        try:
            imgs[prec] = inf.decode_image(lat.cuda(), (64, 48))
        finally:
            inf._vae_precision = None
    _compare(imgs["bf16"], torch.from_numpy(np.asarray(imgs["fp32"])), "decode_image of the same latents: bf16-autocast VAE vs fp32 VAE", 2.0, 8)
