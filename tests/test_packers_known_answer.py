"""SURVEY.md Appendix D known-answer example (generated from the reference) for the oracle packers."""
import torch

from oracle import packers as P


class Tok:
    def encode(self, s):
        return [10, 11, 12, 13][: len(s)]


IDS = dict(bos_token_id=0, eos_token_id=1, start_of_image=2, end_of_image=3)


def test_appendix_d():
    gi, newlens, newrope = P.prepare_prompts([0, 0], [0, 0], ["ab", "c"], Tok(), IDS)
    assert gi["text_token_lens"].tolist() == [4, 3]
    assert gi["packed_text_ids"].tolist() == [0, 10, 11, 1, 0, 10, 1]
    assert gi["packed_text_position_ids"].tolist() == [0, 1, 2, 3, 0, 1, 2]
    assert gi["packed_text_indexes"].tolist() == [0, 1, 2, 3, 4, 5, 6]
    assert gi["packed_key_value_indexes"].tolist() == [] and gi["key_values_lens"].tolist() == [0, 0]
    assert (newlens, newrope) == ([4, 3], [4, 3])
    li = P.prepare_vae_latent([4, 3], [4, 3], [(32, 32), (16, 32)], IDS, 16, 64, 64)
    assert li["packed_text_ids"].tolist() == [2, 3, 2, 3]
    assert li["packed_text_indexes"].tolist() == [0, 5, 6, 9]
    assert li["packed_vae_token_indexes"].tolist() == [1, 2, 3, 4, 7, 8]
    assert li["packed_vae_position_ids"].tolist() == [0, 1, 64, 65, 0, 1]
    assert li["packed_init_noises"].shape == (6, 64) and li["packed_init_noises"].dtype == torch.float32
    assert li["packed_seqlens"].tolist() == [6, 4]
    assert li["packed_position_ids"].tolist() == [4] * 6 + [3] * 4
    assert li["key_values_lens"].tolist() == [4, 3]
    assert li["packed_indexes"].tolist() == [4, 5, 6, 7, 8, 9, 13, 14, 15, 16]
    assert li["packed_key_value_indexes"].tolist() == [0, 1, 2, 3, 10, 11, 12]
    ci = P.prepare_vae_latent_cfg([0, 0], [0, 0], [(32, 32), (16, 32)], 16)
    assert ci["cfg_packed_position_ids"].tolist() == [0] * 10
    assert ci["cfg_packed_query_indexes"].tolist() == list(range(10))
    assert ci["cfg_packed_key_value_indexes"].tolist() == []
    si = P.prepare_start_tokens([4, 3], [4, 3], IDS)
    assert si["packed_start_tokens"].tolist() == [0, 0]
    assert si["packed_query_position_ids"].tolist() == [4, 3]
    assert si["packed_key_value_indexes"].tolist() == list(range(7))
    from oracle.bagel_oracle import flow_schedule
    ts, dts = flow_schedule(5, 3.0)
    assert torch.allclose(ts, torch.tensor([1.0, 0.9, 0.75, 0.5]))
    assert torch.allclose(dts, torch.tensor([0.1, 0.15, 0.25, 0.5]))
