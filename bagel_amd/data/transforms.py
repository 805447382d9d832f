"""Image pre-processing of the reference's ``data/transforms.py`` (MaxLongEdgeMinShortEdgeResize :15-88,
ImageTransform :91-115) with the pixel work on the MI355X.

Same constructor arguments, attributes (``stride``, ``resize_transform``) and call signatures, so
``InterleaveInferencer(model, vae, tokenizer, ImageTransform(1024, 512, 16), ImageTransform(980, 224, 14), ids)`` is the
reference recipe (app.py:137-138).  What runs where:
  * the size rule and the resampling TAPS are host arithmetic (a few hundred doubles per axis, computed exactly as
    Pillow's ``precompute_coeffs`` / ``normalize_coeffs_8bpc`` do and cached per (in, out) size);
  * the two resampling passes, ToTensor and Normalize are HIP kernels (``csrc/image.hip``) on the uint8 image uploaded
    once -- the reference resizes on the CPU (torchvision -> Pillow) and converts on the CPU.
Results are bit-identical to the reference's (integers for the resize; the same fp32 op sequence for the normalisation).
"""
import math

import numpy as np
import torch

from .. import ops

_PRECISION_BITS = 22      # Pillow Resample.c: 32 - 8 - 2


def _keys_bicubic(x):
    """Keys cubic convolution kernel, a = -0.5 (vectorised over a float64 array)."""
    x = np.abs(x)
    near = ((1.5 * x) - 2.5) * x * x + 1.0            # ((a+2)x - (a+3)) x^2 + 1
    far = (((x - 5.0) * x + 8.0) * x - 4.0) * -0.5
    return np.where(x < 1.0, near, np.where(x < 2.0, far, 0.0))


def bicubic_taps(in_size, out_size):
    """-> (bounds int32 [out, 2] = (first tap, tap count), kk int32 [out, ksize]): antialiased bicubic taps in 22-bit
    fixed point for resampling ``in_size`` samples to ``out_size``."""
    scale = float(in_size) / out_size
    fscale = max(scale, 1.0)
    support = 2.0 * fscale
    ksize = int(math.ceil(support)) * 2 + 1
    centers = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    first = np.maximum((centers - support + 0.5).astype(np.int64), 0)          # C (int) cast of a non-negative double
    first = np.where(centers - support + 0.5 < 0, 0, first)
    last = np.minimum((centers + support + 0.5).astype(np.int64), in_size)
    count = last - first
    taps = np.arange(ksize, dtype=np.float64)[None, :]
    w = _keys_bicubic((taps + first[:, None] - centers[:, None] + 0.5) * (1.0 / fscale))
    w = np.where(taps < count[:, None], w, 0.0)
    total = np.zeros(out_size, dtype=np.float64)
    for t in range(ksize):                       # left-to-right accumulation, like the C loop
        total = total + w[:, t]
    w = np.where(total[:, None] != 0.0, w / total[:, None], w)
    fixed = np.where(w < 0, -0.5 + w * (1 << _PRECISION_BITS), 0.5 + w * (1 << _PRECISION_BITS)).astype(np.int64)   # truncation toward zero
    fixed = np.where(taps < count[:, None], fixed, 0)
    return np.stack([first, count], 1).astype(np.int32), fixed.astype(np.int32)


class _TapCache:
    def __init__(self):
        self._c = {}

    def get(self, in_size, out_size, device):
        key = (in_size, out_size, str(device))
        v = self._c.get(key)
        if v is None:
            b, k = bicubic_taps(in_size, out_size)
            v = (b, torch.from_numpy(b).to(device), torch.from_numpy(k).to(device))
            if len(self._c) > 64:
                self._c.pop(next(iter(self._c)))
            self._c[key] = v
        return v


_TAPS = _TapCache()


def resize_bicubic_u8(img, out_h, out_w):
    """(H, W, C) uint8 GPU tensor -> (out_h, out_w, C): Pillow's two-pass 8-bit bicubic (horizontal pass first, only over
    the source rows the vertical pass reads; 8-bit intermediate)."""
    in_h, in_w, C = img.shape
    if (in_h, in_w) == (out_h, out_w):
        return img.clone()
    img = img.contiguous()
    dev = img.device
    cur = img
    bv_host, bv, kv = _TAPS.get(in_h, out_h, dev)
    if in_w != out_w:
        _, bh, kh = _TAPS.get(in_w, out_w, dev)
        first = int(bv_host[0, 0]) if in_h != out_h else 0
        last = int(bv_host[-1, 0] + bv_host[-1, 1]) if in_h != out_h else in_h
        tmp = torch.empty((last - first, out_w, C), dtype=torch.uint8, device=dev)
        ops.resample_u8(cur[first:last], tmp, bh, kh, vertical=False)
        cur = tmp
        if in_h != out_h and first:
            bv = bv.clone()
            bv[:, 0] -= first
    if in_h != out_h:
        out = torch.empty((out_h, out_w, C), dtype=torch.uint8, device=dev)
        ops.resample_u8(cur, out, bv, kv, vertical=True)
        cur = out
    return cur


def _to_u8_hwc(img, device):
    """PIL image / ndarray / tensor -> (H, W, C) uint8 tensor on ``device`` (+ whether the input was a PIL image)."""
    try:
        from PIL import Image
        if isinstance(img, Image.Image):
            if img.mode != "RGB":
                img = img.convert("RGB")
            a = np.array(img, dtype=np.uint8, copy=True)
            return torch.from_numpy(a).to(device), True
    except ImportError:
        pass
    t = torch.as_tensor(img)
    if t.dtype != torch.uint8 or t.dim() != 3:
        raise TypeError("expected a PIL image or an (H, W, C) uint8 array")
    return t.to(device), False


class MaxLongEdgeMinShortEdgeResize:
    """Resize so that the long edge <= max_size, the short edge >= min_size, both sides divisible by ``stride`` and the
    area within ``max_pixels`` (data/transforms.py:15-88); bicubic with antialiasing (the reference's default)."""

    def __init__(self, max_size, min_size, stride, max_pixels, interpolation="bicubic", antialias=True, device="cuda"):
        if str(getattr(interpolation, "value", interpolation)).lower() != "bicubic" or not antialias:
            raise NotImplementedError("only the reference's default (BICUBIC, antialias=True) is implemented")
        self.max_size, self.min_size, self.stride, self.max_pixels = max_size, min_size, stride, max_pixels
        self.interpolation, self.antialias = interpolation, antialias
        self.device = device

    def _make_divisible(self, value, stride):
        return max(stride, int(round(value / stride) * stride))

    def _apply_scale(self, width, height, scale):
        return (self._make_divisible(round(width * scale), self.stride), self._make_divisible(round(height * scale), self.stride))

    def target_size(self, width, height, img_num=1):
        """(new_width, new_height)."""
        scale = min(self.max_size / max(width, height), 1.0)
        scale = max(scale, self.min_size / min(width, height))
        nw, nh = self._apply_scale(width, height, scale)
        if nw * nh > self.max_pixels / img_num:
            nw, nh = self._apply_scale(nw, nh, self.max_pixels / img_num / (nw * nh))
        if max(nw, nh) > self.max_size:
            nw, nh = self._apply_scale(nw, nh, self.max_size / max(nw, nh))
        return nw, nh

    def resize_u8(self, u8_hwc, img_num=1):
        h, w = u8_hwc.shape[:2]
        nw, nh = self.target_size(w, h, img_num)
        return resize_bicubic_u8(u8_hwc, nh, nw)

    def forward(self, img, img_num=1):
        """PIL image -> PIL image (as the reference); (H, W, C) uint8 array/tensor -> uint8 tensor on the device."""
        u8, was_pil = _to_u8_hwc(img, self.device)
        out = self.resize_u8(u8, img_num)
        if was_pil:
            from PIL import Image
            return Image.fromarray(out.cpu().numpy(), "RGB")
        return out

    __call__ = forward


class ImageTransform:
    """resize -> ToTensor -> Normalize (data/transforms.py:91-115).  Returns a (3, H, W) fp32 tensor ON THE GPU."""

    def __init__(self, max_image_size, min_image_size, image_stride, max_pixels=14 * 14 * 9 * 1024,
                 image_mean=[0.5, 0.5, 0.5], image_std=[0.5, 0.5, 0.5], device="cuda"):
        self.stride = image_stride
        self.device = device
        self.resize_transform = MaxLongEdgeMinShortEdgeResize(max_size=max_image_size, min_size=min_image_size,
                                                              stride=image_stride, max_pixels=max_pixels, device=device)
        self.image_mean, self.image_std = list(image_mean), list(image_std)

    def __call__(self, img, img_num=1):
        u8, _ = _to_u8_hwc(img, self.device)
        u8 = self.resize_transform.resize_u8(u8, img_num)
        return ops.u8_to_chw_f32(u8, self.image_mean, self.image_std)
