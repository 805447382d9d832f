"""Host-side tensor utilities on the hot path (integer work, bit-exact vs data/data_utils.py of the reference)."""
import torch


def patchify(image, patch_size):
    """(C,H,W) -> (H/p * W/p, p*p*C) with inner order (p_h, p_w, c)  [data_utils.py:43-50]."""
    c, h, w = image.shape
    p = patch_size
    if h % p or w % p:
        raise AssertionError("image size must be a multiple of the patch size")
    return image.reshape(c, h // p, p, w // p, p).permute(1, 3, 2, 4, 0).reshape((h // p) * (w // p), p * p * c)


def get_flattened_position_ids_extrapolate(img_h, img_w, patch_size, max_num_patches_per_side):
    """id = row * max_side + col  [data_utils.py:53-58]."""
    rows = torch.arange(img_h // patch_size)
    cols = torch.arange(img_w // patch_size)
    return (rows.unsqueeze(1) * max_num_patches_per_side + cols.unsqueeze(0)).reshape(-1)


def get_flattened_position_ids_interpolate(img_h, img_w, patch_size, max_num_patches_per_side):
    """Bucketised fractional coordinates  [data_utils.py:61-69]."""
    nh, nw = img_h // patch_size, img_w // patch_size
    edges = torch.arange(1 / max_num_patches_per_side, 1.0, 1 / max_num_patches_per_side)
    bh = torch.bucketize(torch.arange(0, 1 - 1e-6, 1 / nh), edges, right=True)
    bw = torch.bucketize(torch.arange(0, 1 - 1e-6, 1 / nw), edges, right=True)
    return (bh.unsqueeze(1) * max_num_patches_per_side + bw.unsqueeze(0)).reshape(-1)


def pil_img2rgb(image):
    """RGBA / palette-transparency images are composited on white  [data_utils.py:118-127]."""
    from PIL import Image
    if image.mode == "RGBA" or image.info.get("transparency", None) is not None:
        image = image.convert("RGBA")
        canvas = Image.new(mode="RGB", size=image.size, color=(255, 255, 255))
        canvas.paste(image, mask=image.split()[3])
        return canvas
    return image.convert("RGB")


def add_special_tokens(tokenizer):
    """Resolve the four chat / vision marker ids, adding the tokens if the vocabulary lacks them [data_utils.py:130-165]."""
    known = []
    for v in tokenizer.special_tokens_map.values():
        known += [v] if isinstance(v, str) else list(v)
    names = ["<|im_start|>", "<|im_end|>", "<|vision_start|>", "<|vision_end|>"]
    num_new = tokenizer.add_tokens([n for n in names if n not in known])
    ids = [tokenizer.convert_tokens_to_ids(n) for n in names]
    new_token_ids = dict(bos_token_id=ids[0], eos_token_id=ids[1], start_of_image=ids[2], end_of_image=ids[3])
    return tokenizer, new_token_ids, num_new
