"""CPU: the restatement of the device-side next-token sampler (oracle/sampling.py) against PUBLISHED vectors and against the distribution it must draw from.

* Philox4x32-10: the known-answer vectors of the Random123 distribution (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11;
  ``kat_vectors``: philox4x32 10 rounds) -- counter / key all zeros, all ones, and the digits of pi.
* Gumbel-max == multinomial(softmax(logits / T)) (bagel.py:980-983): empirical frequencies over 40 000 draws within 4.5 sigma of the softmax probabilities, and the
  host path of ``generate_text(do_sample=True)`` (stand-in operators) is reproducible under ``torch.manual_seed`` and changes with the seed."""
import copy

import numpy as np
import pytest
import torch

from oracle import sampling as S


def test_philox4x32_10_known_answer_vectors():
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = S.philox4x32_10([ctr[0]], [ctr[1]], [ctr[2]], [ctr[3]], key[0], key[1])
        assert tuple(int(x[0]) for x in got) == want


def test_gumbel_max_draws_from_the_softmax_distribution():
    g = np.random.default_rng(0)
    V, T, rows = 13, 0.7, 8000
    logits = torch.tensor(g.normal(0, 2.0, V), dtype=torch.float32).to(torch.bfloat16).float().numpy()
    counts = np.zeros(V)
    for step in range(5):                                     # 5 steps x 8000 rows (the row is part of the counter): 40 000 independent draws
        tok = S.sample_gumbel(np.tile(logits, (rows, 1)), T, seed=0x1234567887654321 & (2 ** 62 - 1), step=step)
        counts += np.bincount(tok, minlength=V)
    z = S.bf16_round(logits / np.float32(T)).astype(np.float64)
    p = np.exp(z - z.max()); p /= p.sum()
    n = counts.sum()
    sigma = np.sqrt(n * p * (1 - p))
    assert (np.abs(counts - n * p) <= 4.5 * sigma + 1).all(), (counts, n * p)
    # different steps / rows / seeds give different numbers
    a = S.sample_gumbel(np.tile(logits, (64, 1)), T, seed=1, step=0)
    assert not np.array_equal(a, S.sample_gumbel(np.tile(logits, (64, 1)), T, seed=1, step=1))
    assert not np.array_equal(a, S.sample_gumbel(np.tile(logits, (64, 1)), T, seed=2, step=0))
    assert len(set(a.tolist())) > 3


def test_gumbel_value_is_finite_on_the_edge_draws():
    """x = 0xFFFFFFFF must not map to u == 1 (Gumbel +inf: that column would win whatever its logit) and x = 0 not to u == 0 (-inf).  The 24-bit form
    ((x >> 8) + 0.5) * 2^-24 rounds 0xFFFFFF + 0.5 up to 2^24 in fp32; the 23-bit form is exact."""
    x = np.array([0, 1, 0x1ff, 0x200, 0x7fffffff, 0x80000000, 0xfffffdff, 0xfffffe00, 0xffffffff], dtype=np.uint32)
    g = S.gumbel_of(x)
    assert np.isfinite(g).all(), g
    assert (np.diff(g) >= 0).all()                                # monotone in the draw
    u = ((x >> np.uint32(9)).astype(np.float64) + 0.5) * 2.0 ** -23
    assert (u > 0).all() and (u < 1).all() and u[-1] == 1 - 2.0 ** -24 and u[0] == 2.0 ** -24
    np.testing.assert_allclose(g, -np.log(-np.log(u)), rtol=2e-6, atol=2e-6)      # the fp32 evaluation agrees with fp64 on exact uniforms
    # every one of the 2^23 uniforms is exactly representable in fp32 (spot-check the top of the range, where the 24-bit form broke)
    top = np.arange(2 ** 23 - 4096, 2 ** 23, dtype=np.uint32)
    uf = (top.astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -23)
    assert (uf.astype(np.float64) == (top.astype(np.float64) + 0.5) * 2.0 ** -23).all() and (uf < 1).all()
    assert abs(float(g[-1]) - 16.6355) < 1e-3 and abs(float(g[0]) + 2.8114) < 1e-3   # -log(-log(1 - 2^-24)) = 24 ln 2, -log(24 ln 2)


def test_generate_text_sampling_host_path_is_seeded(monkeypatch):
    from oracle.configs import TINY_D128 as cfg, NEW_TOKEN_IDS_TINY, StubTokenizer
    from tests import mock_ops
    from tests.test_host_logic_cpu import cpu_model, new_cache
    mock_ops.install(monkeypatch)
    model = cpu_model(cfg)
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    gi, lens, ropes = model.prepare_prompts([0], [0], ["a small red cube"], tok, NEW_TOKEN_IDS_TINY)
    cache = model.forward_cache_update_text(new_cache(cfg), **gi)
    si = model.prepare_start_tokens(lens, ropes, NEW_TOKEN_IDS_TINY)
    outs = []
    for seed in (5, 5, 6):
        torch.manual_seed(seed)
        outs.append(model.generate_text(past_key_values=copy.deepcopy(cache), max_length=8, do_sample=True, temperature=1.5, end_token_id=None, use_graph=False, **si))
        sess = model._last_decode_session
        assert sess.sampler is not None and sess.sampler[0] == "gumbel" and sess.sampler[1] == 1.5
    assert torch.equal(outs[0], outs[1]) and not torch.equal(outs[0], outs[2])
    assert (outs[0] >= 0).all() and (outs[0] < cfg["llm"]["vocab_size"]).all()
    # the torch sampler stays selectable
    monkeypatch.setenv("BAGEL_DECODE_SAMPLER", "torch")
    torch.manual_seed(5)
    t = model.generate_text(past_key_values=copy.deepcopy(cache), max_length=4, do_sample=True, temperature=1.5, end_token_id=None, use_graph=False, **si)
    assert model._last_decode_session.sampler is None and t.shape == (4, 1)
    with pytest.raises(ValueError):
        from bagel_amd.modeling.bagel.decode import DecodeSession
        DecodeSession(model.language_model.engine(), None, None, None, [0], torch.zeros(1, dtype=torch.long), torch.zeros(1, dtype=torch.long), 1, sampler=("gumbel", 0.0, 1))
