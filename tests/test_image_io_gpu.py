"""GPU parity of the device-side image pre/post-processing (data/transforms.py, inferencer.py:174-185): byte-exact against
the Pillow restatement and the reference's transform goldens."""
import pytest
import torch

from tests.test_ops_gpu import rnd

pytestmark = pytest.mark.gpu

BF16 = torch.bfloat16
DEV = "cuda"


# ------------------------------------------------------------------------------------------------------------
# image pre/post-processing on the device (data/transforms.py, inferencer.py:174-185) -- byte-exact
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("hw,out", [((37, 53), (24, 31)), ((24, 31), (56, 70)), ((100, 64), (33, 64)), ((64, 100), (64, 41)),
                                    ((17, 200), (140, 28)), ((301, 299), (98, 112)), ((8, 8), (224, 224)), ((700, 1100), (224, 352)),
                                    ((40, 40), (40, 40))])
def test_device_resize_equals_pillow_restatement(hw, out):
    import numpy as np
    from bagel_amd.data.transforms import resize_bicubic_u8
    from oracle import image_io as IO
    rng = np.random.default_rng(hw[0] * 1000 + hw[1])
    a = rng.integers(0, 256, hw + (3,), dtype=np.uint8)
    if hw == (8, 8):
        a[:] = np.array([[0, 255] * 4] * 8, dtype=np.uint8)[..., None]
    got = resize_bicubic_u8(torch.from_numpy(a).to(DEV), out[0], out[1]).cpu().numpy()
    assert np.array_equal(got, IO.resize_bicubic_u8(a, out[0], out[1]))


def test_image_transform_matches_reference_goldens(golden):
    """ImageTransform (PIL image in, normalised CHW fp32 on the GPU out) == the reference's data/transforms.py output, bit for bit;
    resize_transform keeps the reference's PIL -> PIL contract."""
    from PIL import Image
    from bagel_amd.data.transforms import ImageTransform
    for case in golden("image_io")["transform"]:
        mx, mn, st = case["limits"]
        t = ImageTransform(mx, mn, st)
        pil = Image.fromarray(case["image"].numpy(), "RGB")
        out = t(pil)
        assert out.is_cuda and out.dtype == torch.float32
        assert torch.equal(out.cpu(), case["out"])
        assert torch.equal(t(case["image"]).cpu(), case["out"])              # uint8 HWC tensor input
        r = t.resize_transform(pil)
        assert isinstance(r, Image.Image) and r.size == (case["out"].shape[2], case["out"].shape[1])
        assert t.stride == st


def test_decode_image_u8_conversion_exact():
    import numpy as np
    from bagel_amd.inferencer import InterleaveInferencer
    from oracle import image_io as IO
    x = (rnd(1, 3, 37, 53, seed=4, dtype=torch.float32) * 0.8)
    x[0, :, 0, :8] = torch.tensor([-1.0, -0.999, 0.0, 0.003, 0.999, 1.0, 1.5, -3.0])
    got = InterleaveInferencer.image_to_u8(x.to(DEV)).cpu().numpy()
    assert got.shape == (37, 53, 3) and np.array_equal(got, IO.image_to_u8(x[0].numpy()))
