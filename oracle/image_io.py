"""CPU restatement of the image pre/post-processing around BAGEL's forward path (the ORACLE for SURVEY.md 8f.3).

TEST INFRASTRUCTURE ONLY -- never imported by ``bagel_amd`` (the product).

What the reference does (data/transforms.py:15-115, inferencer.py:174-185):
  * ``MaxLongEdgeMinShortEdgeResize.forward``: pick (new_w, new_h) from the long/short-edge limits, the stride and the
    pixel budget, then ``torchvision.transforms.functional.resize(img, (new_h, new_w), BICUBIC, antialias=True)``.  For a
    PIL image torchvision dispatches to ``PIL.Image.resize((w, h), BICUBIC)`` -- the arithmetic therefore lives in the
    third-party dependency **Pillow** (``src/libImaging/Resample.c``; the reference pins no version, this image has
    12.2.0): a separable convolution with the Keys bicubic kernel (a = -0.5), support scaled by the down-sampling factor
    (antialias), coefficients normalised in double precision and quantised to 22-bit fixed point, horizontal pass first,
    8-bit intermediate, results rounded by adding 2^21 before the shift and clamped to [0, 255].
  * ``ToTensor`` + ``Normalize(0.5, 0.5)``: ``((u8 / 255) - 0.5) / 0.5`` in fp32, CHW.
  * ``decode_image``: ``((x * 0.5 + 0.5).clamp(0, 1) * 255).to(uint8)`` -- a TRUNCATING cast -- HWC.

Parity status: PINNED -- ``tests/test_image_io_cpu.py`` checks ``resize_bicubic_u8`` byte for byte against
``PIL.Image.resize`` (the dependency itself, importable on every box of this image) over up- and down-scaling cases,
and the size rule against values computed by running the reference's own class body (pure Python arithmetic).
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2      # Resample.c: 8-bit pixels, 2 guard bits


def target_size(width, height, max_size, min_size, stride, max_pixels, img_num=1):
    """(new_width, new_height) of MaxLongEdgeMinShortEdgeResize.forward (data/transforms.py:47-88)."""
    def divisible(v):
        return max(stride, int(round(v / stride) * stride))

    def scaled(w, h, s):
        return divisible(round(w * s)), divisible(round(h * s))

    scale = min(max_size / max(width, height), 1.0)
    scale = max(scale, min_size / min(width, height))
    nw, nh = scaled(width, height, scale)
    if nw * nh > max_pixels / img_num:
        nw, nh = scaled(nw, nh, max_pixels / img_num / (nw * nh))
    if max(nw, nh) > max_size:
        nw, nh = scaled(nw, nh, max_size / max(nw, nh))
    return nw, nh


def _bicubic(x):
    """Keys kernel, a = -0.5 (Resample.c ``bicubic_filter``), evaluated in float64 like the C doubles."""
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_coeffs(in_size, out_size):
    """``precompute_coeffs`` + ``normalize_coeffs_8bpc`` for the box (0, in_size): -> (ksize, bounds int32 [out, 2] as
    (first tap, tap count), kk int32 [out, ksize] fixed-point weights)."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _pass(src, bounds, kk, axis):
    """One separable pass over ``axis`` of an (H, W, C) uint8 array: fixed-point sum, +2^21, >> 22, clamp."""
    src = np.moveaxis(src, axis, 0).astype(np.int64)            # [n_in, ...]
    out = np.empty((bounds.shape[0],) + src.shape[1:], dtype=np.uint8)
    for o in range(bounds.shape[0]):
        first, n = int(bounds[o, 0]), int(bounds[o, 1])
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for t in range(n):
            acc += src[first + t] * int(kk[o, t])
        out[o] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bicubic_u8(img, out_h, out_w):
    """``ImagingResample`` (8 bits per channel) of an (H, W, C) uint8 array to (out_h, out_w, C): horizontal pass first
    (only over the source rows the vertical pass will read), 8-bit intermediate, then the vertical pass."""
    img = np.ascontiguousarray(img)
    in_h, in_w = img.shape[:2]
    if (in_h, in_w) == (out_h, out_w):
        return img.copy()
    _, bh, kh = resample_coeffs(in_w, out_w)
    _, bv, kv = resample_coeffs(in_h, out_h)
    need_h, need_v = in_w != out_w, in_h != out_h
    cur = img
    if need_h:
        first = int(bv[0, 0])
        last = int(bv[-1, 0] + bv[-1, 1])
        bv = bv.copy()
        bv[:, 0] -= first
        cur = _pass(cur[first:last], bh, kh, axis=1)
    if need_v:
        cur = _pass(cur, bv, kv, axis=0)
    return cur


def to_tensor_normalize(u8_hwc, mean=0.5, std=0.5):
    """ToTensor + Normalize (data/transforms.py:109-115): fp32 CHW, ((x / 255) - mean) / std, every op in fp32."""
    x = np.ascontiguousarray(u8_hwc).astype(np.float32) / np.float32(255)
    x = (x - np.float32(mean)) / np.float32(std)
    return np.ascontiguousarray(np.moveaxis(x, 2, 0))


def image_to_u8(chw_f32):
    """decode_image (inferencer.py:182-183): (x * 0.5 + 0.5).clamp(0, 1) * 255 -> uint8 (truncation), HWC."""
    x = np.asarray(chw_f32, dtype=np.float32)
    y = np.clip(x * np.float32(0.5) + np.float32(0.5), np.float32(0), np.float32(1)) * np.float32(255)
    return np.ascontiguousarray(np.moveaxis(y, 0, 2)).astype(np.uint8)
