"""CPU restatement of BAGEL's host-side sequence packers (integer work: BIT-EXACT contract).

TEST INFRASTRUCTURE ONLY.  Plain-python loops, one function per reference method, checked against
the unmodified reference by oracle/make_golden.py (fixtures in tests/golden/packers_*.pt) and
against SURVEY.md Appendix D's known-answer example in tests/test_packers.py.

Merged-KV layout produced by every packer: ``[ctx_0 | query_0 | ctx_1 | query_1 | ...]``.
"""
import torch


def position_ids_extrapolate(img_h, img_w, patch, max_side):
    """get_flattened_position_ids_extrapolate, data/data_utils.py:53-58."""
    nh, nw = img_h // patch, img_w // patch
    return (torch.arange(0, nh)[:, None] * max_side + torch.arange(0, nw)).flatten()


def position_ids_interpolate(img_h, img_w, patch, max_side):
    """get_flattened_position_ids_interpolate, data/data_utils.py:61-69."""
    nh, nw = img_h // patch, img_w // patch
    bounds = torch.arange(1 / max_side, 1.0, 1 / max_side)
    fh = torch.arange(0, 1 - 1e-6, 1 / nh)
    fw = torch.arange(0, 1 - 1e-6, 1 / nw)
    bh = torch.bucketize(fh, bounds, right=True)
    bw = torch.bucketize(fw, bounds, right=True)
    return (bh[:, None] * max_side + bw).flatten()


def patchify(image, p):
    """data/data_utils.py:43-50: (C,H,W) -> (H/p*W/p, p*p*C), inner order (p_h, p_w, c)."""
    c, h, w = image.shape
    assert h % p == 0 and w % p == 0
    return torch.einsum("chpwq->hwpqc", image.reshape(c, h // p, p, w // p, p)).reshape(-1, p * p * c)


def prepare_prompts(curr_kvlens, curr_rope, prompts, tokenizer, new_token_ids):
    """Bagel.prepare_prompts, bagel.py:232-264."""
    ids, pos, lens, idx, kvidx = [], [], [], [], []
    curr = 0
    newlens, newrope = [], []
    for prompt, kvlen, rope in zip(prompts, curr_kvlens, curr_rope):
        kvidx.extend(range(curr, curr + kvlen))
        curr += kvlen
        t = [new_token_ids["bos_token_id"]] + list(tokenizer.encode(prompt)) + [new_token_ids["eos_token_id"]]
        lens.append(len(t))
        ids.extend(t)
        pos.extend(range(rope, rope + len(t)))
        idx.extend(range(curr, curr + len(t)))
        newlens.append(kvlen + len(t))
        newrope.append(rope + len(t))
        curr += len(t)
    gi = dict(
        text_token_lens=torch.tensor(lens, dtype=torch.int),
        packed_text_ids=torch.tensor(ids, dtype=torch.long),
        packed_text_position_ids=torch.tensor(pos, dtype=torch.long),
        packed_text_indexes=torch.tensor(idx, dtype=torch.long),
        packed_key_value_indexes=torch.tensor(kvidx, dtype=torch.long),
        key_values_lens=torch.tensor(curr_kvlens, dtype=torch.int),
    )
    return gi, newlens, newrope


def _image_block(curr, qcurr, n_tokens, kvlen, rope, st):
    """Shared body of the three image packers: <start> n tokens <end> after kvlen cached rows."""
    st["kvidx"].extend(range(curr, curr + kvlen))
    curr += kvlen
    st["text_idx"].append(qcurr)
    st["idx"].append(curr)
    curr += 1
    qcurr += 1
    st["tok_idx"].extend(range(qcurr, qcurr + n_tokens))
    st["idx"].extend(range(curr, curr + n_tokens))
    curr += n_tokens
    qcurr += n_tokens
    st["text_idx"].append(qcurr)
    st["idx"].append(curr)
    curr += 1
    qcurr += 1
    st["pos"].extend([rope] * (n_tokens + 2))
    st["seqlens"].append(n_tokens + 2)
    return curr, qcurr


def _new_state():
    return dict(kvidx=[], text_idx=[], idx=[], tok_idx=[], pos=[], seqlens=[])


def prepare_vit_images(curr_kvlens, curr_rope, images, transforms, new_token_ids, vit_patch, vit_max_side,
                       pos_fn=position_ids_extrapolate):
    """Bagel.prepare_vit_images, bagel.py:299-359."""
    st = _new_state()
    text_ids, vit_lens, vit_tokens, vit_pos = [], [], [], []
    curr = qcurr = 0
    newlens, newrope = [], []
    for image, kvlen, rope in zip(images, curr_kvlens, curr_rope):
        t = transforms(image)
        tokens = patchify(t, vit_patch)
        n = tokens.shape[0]
        text_ids += [new_token_ids["start_of_image"], new_token_ids["end_of_image"]]
        vit_pos.append(pos_fn(t.size(1), t.size(2), vit_patch, vit_max_side))
        vit_tokens.append(tokens)
        vit_lens.append(n)
        curr, qcurr = _image_block(curr, qcurr, n, kvlen, rope, st)
        newlens.append(kvlen + n + 2)
        newrope.append(rope + 1)
    gi = dict(
        packed_text_ids=torch.tensor(text_ids, dtype=torch.long),
        packed_text_indexes=torch.tensor(st["text_idx"], dtype=torch.long),
        vit_token_seqlens=torch.tensor(vit_lens, dtype=torch.int),
        packed_vit_tokens=torch.cat(vit_tokens, 0),
        packed_vit_position_ids=torch.cat(vit_pos, 0),
        packed_vit_token_indexes=torch.tensor(st["tok_idx"], dtype=torch.long),
        packed_position_ids=torch.tensor(st["pos"], dtype=torch.long),
        packed_seqlens=torch.tensor(st["seqlens"], dtype=torch.int),
        packed_indexes=torch.tensor(st["idx"], dtype=torch.long),
        packed_key_value_indexes=torch.tensor(st["kvidx"], dtype=torch.long),
        key_values_lens=torch.tensor(curr_kvlens, dtype=torch.int),
    )
    return gi, newlens, newrope


def prepare_vae_images(curr_kvlens, curr_rope, images, transforms, new_token_ids, latent_downsample,
                       max_latent_size, timestep=0, pos_fn=position_ids_extrapolate):
    """Bagel.prepare_vae_images, bagel.py:417-488."""
    st = _new_state()
    text_ids, shapes, vpos, tensors = [], [], [], []
    curr = qcurr = 0
    newlens, newrope = [], []
    for image, kvlen, rope in zip(images, curr_kvlens, curr_rope):
        t = transforms(image)
        tensors.append(t)
        vpos.append(pos_fn(t.size(1), t.size(2), latent_downsample, max_latent_size))
        H, W = t.shape[1:]
        h, w = H // latent_downsample, W // latent_downsample
        shapes.append((h, w))
        text_ids += [new_token_ids["start_of_image"], new_token_ids["end_of_image"]]
        curr, qcurr = _image_block(curr, qcurr, h * w, kvlen, rope, st)
        newlens.append(kvlen + h * w + 2)
        newrope.append(rope + 1)
    sizes = [tuple(t.shape) for t in tensors]
    mx = [max(s) for s in zip(*sizes)]
    padded = torch.zeros(size=(len(tensors), *mx))
    for i, t in enumerate(tensors):
        padded[i, :, : t.shape[1], : t.shape[2]] = t
    gi = dict(
        padded_images=padded,
        patchified_vae_latent_shapes=shapes,
        packed_vae_position_ids=torch.cat(vpos, 0),
        packed_timesteps=torch.tensor([timestep]),
        packed_vae_token_indexes=torch.tensor(st["tok_idx"], dtype=torch.long),
        packed_text_ids=torch.tensor(text_ids, dtype=torch.long),
        packed_text_indexes=torch.tensor(st["text_idx"], dtype=torch.long),
        packed_position_ids=torch.tensor(st["pos"], dtype=torch.long),
        packed_seqlens=torch.tensor(st["seqlens"], dtype=torch.int),
        packed_indexes=torch.tensor(st["idx"], dtype=torch.long),
        packed_key_value_indexes=torch.tensor(st["kvidx"], dtype=torch.long),
        key_values_lens=torch.tensor(curr_kvlens, dtype=torch.int),
    )
    return gi, newlens, newrope


def prepare_vae_latent(curr_kvlens, curr_rope, image_sizes, new_token_ids, latent_downsample, max_latent_size,
                       patch_latent_dim, pos_fn=position_ids_extrapolate):
    """Bagel.prepare_vae_latent, bagel.py:552-608.  Draws torch.randn per image, in sample order,
    from the global CPU generator (bagel.py:578-580)."""
    st = _new_state()
    text_ids, vpos, noises = [], [], []
    curr = qcurr = 0
    for (H, W), kvlen, rope in zip(image_sizes, curr_kvlens, curr_rope):
        vpos.append(pos_fn(H, W, latent_downsample, max_latent_size))
        h, w = H // latent_downsample, W // latent_downsample
        noises.append(torch.randn(h * w, patch_latent_dim))
        text_ids += [new_token_ids["start_of_image"], new_token_ids["end_of_image"]]
        curr, qcurr = _image_block(curr, qcurr, h * w, kvlen, rope, st)
    return dict(
        packed_text_ids=torch.tensor(text_ids, dtype=torch.long),
        packed_text_indexes=torch.tensor(st["text_idx"], dtype=torch.long),
        packed_init_noises=torch.cat(noises, 0),
        packed_vae_position_ids=torch.cat(vpos, 0),
        packed_vae_token_indexes=torch.tensor(st["tok_idx"], dtype=torch.long),
        packed_seqlens=torch.tensor(st["seqlens"], dtype=torch.int),
        packed_position_ids=torch.tensor(st["pos"], dtype=torch.long),
        key_values_lens=torch.tensor(curr_kvlens, dtype=torch.int),
        packed_indexes=torch.tensor(st["idx"], dtype=torch.long),
        packed_key_value_indexes=torch.tensor(st["kvidx"], dtype=torch.long),
    )


def prepare_vae_latent_cfg(curr_kvlens, curr_rope, image_sizes, latent_downsample):
    """Bagel.prepare_vae_latent_cfg, bagel.py:610-641."""
    st = _new_state()
    curr = qcurr = 0
    for (H, W), kvlen, rope in zip(image_sizes, curr_kvlens, curr_rope):
        h, w = H // latent_downsample, W // latent_downsample
        curr, qcurr = _image_block(curr, qcurr, h * w, kvlen, rope, st)
    return dict(
        cfg_packed_position_ids=torch.tensor(st["pos"], dtype=torch.long),
        cfg_key_values_lens=torch.tensor(curr_kvlens, dtype=torch.int),
        cfg_packed_query_indexes=torch.tensor(st["idx"], dtype=torch.long),
        cfg_packed_key_value_indexes=torch.tensor(st["kvidx"], dtype=torch.long),
    )


def prepare_start_tokens(curr_kvlens, curr_rope, new_token_ids):
    """Bagel.prepare_start_tokens, bagel.py:909-927."""
    start, kvidx, pos = [], [], []
    curr = 0
    for kvlen, rope in zip(curr_kvlens, curr_rope):
        kvidx.extend(range(curr, curr + kvlen))
        start.append(new_token_ids["bos_token_id"])
        pos.append(rope)
        curr += kvlen
    return dict(
        packed_start_tokens=torch.tensor(start, dtype=torch.long),
        packed_query_position_ids=torch.tensor(pos, dtype=torch.long),
        key_values_lens=torch.tensor(curr_kvlens, dtype=torch.int),
        packed_key_value_indexes=torch.tensor(kvidx, dtype=torch.long),
    )
