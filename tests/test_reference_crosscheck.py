"""Build-container only: oracle/shapes.py and oracle/weights.py agree with the real reference modules."""
import pytest
import torch

from oracle import ref_env
from oracle.configs import TINY, TINY_D128, TINY_DENSE, TINY_MOE, TINY_ROPE
from oracle.shapes import bagel_shapes, vae_shapes
from tests.util_models import oracle_weights

pytestmark = pytest.mark.skipif(not ref_env.reference_available(), reason="/root/reference not present")


@pytest.mark.parametrize("cfg", [TINY, TINY_D128, TINY_ROPE, TINY_DENSE, TINY_MOE], ids=lambda c: c["name"])
def test_shapes_and_weights_equal_reference(cfg):
    from oracle import make_golden as G
    model, vae, W, VW = G.build(cfg)
    assert {k: tuple(v.shape) for k, v in W.items()} == bagel_shapes(cfg)
    assert {k: tuple(v.shape) for k, v in VW.items()} == vae_shapes(cfg["vae"])
    oW, oVW = oracle_weights(cfg)
    for k in W:
        assert torch.equal(W[k], oW[k]), k
    for k in VW:
        assert torch.equal(VW[k], oVW[k]), k


def test_position_id_helpers_equal_reference():
    """oracle/packers.py vs the reference's data/data_utils.py:43-69 (patchify, both position-id variants): bit-exact integers."""
    ref_env.activate()
    from data import data_utils as R
    from oracle import packers as P
    for (h, w, p, side) in ((64, 64, 16, 64), (1024, 1024, 16, 64), (980, 980, 14, 70), (42, 56, 14, 10), (48, 80, 16, 32), (224, 448, 14, 70)):
        assert torch.equal(R.get_flattened_position_ids_extrapolate(h, w, p, side), P.position_ids_extrapolate(h, w, p, side))
        assert torch.equal(R.get_flattened_position_ids_interpolate(h, w, p, side), P.position_ids_interpolate(h, w, p, side))
    img = torch.randn(3, 42, 56)
    assert torch.equal(R.patchify(img, 14), P.patchify(img, 14))
