"""CPU: image pre/post-processing.  The oracle (oracle/image_io.py) against
  (a) the golden vectors produced from the reference's unmodified data/transforms.py (oracle/make_golden_image.py),
  (b) Pillow itself -- the third-party dependency that holds the resampling arithmetic, importable on every box,
and the product's HOST arithmetic (size rule, fixed-point taps) against the oracle.  All comparisons are bit-exact."""
import numpy as np
import pytest
import torch

from oracle import image_io as IO


def test_oracle_matches_reference_transform_goldens(golden):
    g = golden("image_io")
    for (w, h, mx, mn, st, mp, n), got in g["size_rule"]:
        assert IO.target_size(w, h, mx, mn, st, mp, n) == tuple(got)
    for case in g["transform"]:
        a = case["image"].numpy()
        mx, mn, st = case["limits"]
        nw, nh = IO.target_size(a.shape[1], a.shape[0], mx, mn, st, 14 * 14 * 9 * 1024)
        out = torch.from_numpy(IO.to_tensor_normalize(IO.resize_bicubic_u8(a, nh, nw)))
        assert torch.equal(out, case["out"])


@pytest.mark.parametrize("hw,out", [((37, 53), (24, 31)), ((24, 31), (56, 70)), ((100, 64), (33, 64)), ((64, 100), (64, 41)),
                                    ((17, 200), (140, 28)), ((301, 299), (98, 112)), ((8, 8), (224, 224)), ((40, 40), (40, 40))])
def test_oracle_resize_equals_pillow(hw, out):
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(hw[0] * 1000 + hw[1])
    a = rng.integers(0, 256, hw + (3,), dtype=np.uint8)
    if hw == (8, 8):
        a[:] = np.array([[0, 255] * 4] * 8, dtype=np.uint8)[..., None]      # hard edges: overshoot -> the clamp path
    ref = np.asarray(Image.fromarray(a, "RGB").resize((out[1], out[0]), Image.BICUBIC))
    assert np.array_equal(IO.resize_bicubic_u8(a, out[0], out[1]), ref)


def test_product_host_arithmetic_equals_oracle(golden):
    """bagel_amd/data/transforms.py: the size rule and the fixed-point taps are host code -- identical integers."""
    from bagel_amd.data.transforms import MaxLongEdgeMinShortEdgeResize, bicubic_taps
    for a, b in [(53, 31), (31, 70), (200, 28), (17, 140), (299, 112), (8, 224), (400, 48), (1024, 980), (4000, 1024), (3, 7), (1, 5), (5, 1)]:
        ks, bo, kk = IO.resample_coeffs(a, b)
        bp, kp = bicubic_taps(a, b)
        assert kp.shape[1] == ks and np.array_equal(bo, bp) and np.array_equal(kk, kp), (a, b)
    for (w, h, mx, mn, st, mp, n), got in golden("image_io")["size_rule"]:
        assert MaxLongEdgeMinShortEdgeResize(mx, mn, st, mp, device="cpu").target_size(w, h, n) == tuple(got)


def test_image_to_u8_truncates():
    x = torch.tensor([[[-1.0, -0.999, 0.0, 0.003, 0.999, 1.0, 1.5, -3.0]]]).repeat(3, 1, 1).numpy()
    u = IO.image_to_u8(x)
    ref = ((torch.from_numpy(x) * 0.5 + 0.5).clamp(0, 1).permute(1, 2, 0) * 255).to(torch.uint8).numpy()
    assert np.array_equal(u, ref) and u[0, :, 0].tolist() == [0, 0, 127, 127, 254, 255, 255, 0]
