"""Run by tests/test_reference_crosscheck.py in a fresh interpreter (the module aliases must not leak into pytest's process):
the UNMODIFIED reference inferencer (/root/reference/inferencer.py, loaded by path) drives the PRODUCT's model, VAE and image
transforms through ``bagel_amd.install_as_reference()`` -- host logic on the torch stand-ins of tests/mock_ops.py (no GPU here) --
and must reproduce the outputs the reference inferencer produced with the reference model (tests/golden/tiny_inferencer.pt)."""
import importlib.util, re, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bagel_amd
bagel_amd.install_as_reference()
from tests import mock_ops
class MP:
    def setattr(self, o, n, v): setattr(o, n, v)
mock_ops.install(MP())
os.environ["BAGEL_DECODE_GRAPH"] = "0"
spec = importlib.util.spec_from_file_location("reference_inferencer", "/root/reference/inferencer.py")
R = importlib.util.module_from_spec(spec); spec.loader.exec_module(R)
from PIL import Image
from oracle.configs import TINY, NEW_TOKEN_IDS_TINY, StubTokenizer
from tests.test_host_logic_cpu import cpu_model_and_vae
from bagel_amd.data.transforms import ImageTransform
g = torch.load(os.path.join(sys.path[0], 'tests', 'golden', 'tiny_inferencer.pt'), weights_only=False)
model, vae = cpu_model_and_vae(TINY)
tok = StubTokenizer(TINY["llm"]["vocab_size"])
inf = R.InterleaveInferencer(model, vae, tok, ImageTransform(64, 32, 16, device="cpu"), ImageTransform(56, 28, 14, device="cpu"), NEW_TOKEN_IDS_TINY)
assert type(inf).__module__ == "reference_inferencer"
src = Image.fromarray(g["source_image"].numpy(), "RGB")
def d(img, ref):
    x = np.abs(np.asarray(img).astype(np.int32) - ref.numpy().astype(np.int32)); return x.mean(), np.percentile(x, 99)
torch.manual_seed(g["t2i"]["seed"])
r = inf(text=g["t2i"]["text"], **g["t2i"]["kwargs"])
m, p99 = d(r["image"], g["t2i"]["image"])
assert m <= 1.5 and p99 <= 8, ("text -> image", m, p99)
torch.manual_seed(g["edit"]["seed"])
r = inf(image=src, text=g["edit"]["text"], **g["edit"]["kwargs"])
m, p99 = d(r["image"], g["edit"]["image"])
assert m <= 3.0 and p99 <= 14, ("image + text -> image", m, p99)
r = inf(image=src, text=g["understanding"]["text"], **g["understanding"]["kwargs"])
ours, ref = re.findall(r"\[(\d+)\]", r["text"]), re.findall(r"\[(\d+)\]", g["understanding"]["answer"])
first = next((i for i, (a, b) in enumerate(zip(ours, ref)) if a != b), len(ref))
assert len(ours) == len(ref) and first >= 1, (r["text"], g["understanding"]["answer"])
print("ok")
