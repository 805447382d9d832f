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
