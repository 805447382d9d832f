"""Golden vectors for the image pre-processing row from the UNMODIFIED reference ``data/transforms.py``.

TEST INFRASTRUCTURE ONLY.  Run in the build container (needs /root/reference):  python -m oracle.make_golden_image

``data/transforms.py`` imports cv2 and torchvision, neither of which is installed here (nor needed by the classes on
this path).  Stand-ins that restate the published behaviour of the three torchvision entry points the reference calls
on a PIL image are registered before the import (like the flash-attn stand-in of SURVEY.md 8c.1):
  * ``functional.resize(PIL, (h, w), BICUBIC, antialias)``  ->  ``img.resize((w, h), PIL.Image.BICUBIC)``  (torchvision
    ``_functional_pil.resize``; Pillow always antialiases),
  * ``ToTensor()``  ->  uint8 HWC -> CHW float32 ``.div(255)``,
  * ``Normalize(mean, std, inplace=True)``  ->  ``t.sub_(mean[:, None, None]).div_(std[:, None, None])``.
The resampling arithmetic itself is Pillow's (importable on every box); the reference's own code contributes the size
rule (``MaxLongEdgeMinShortEdgeResize``) and the call sequence.  The oracle (oracle/image_io.py) must agree BIT FOR BIT
before the fixture is written.
"""
import enum
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import image_io as IO      # noqa: E402
from oracle import ref_env             # noqa: E402


def _install_standins():
    from PIL import Image

    class InterpolationMode(enum.Enum):
        NEAREST = "nearest"
        BILINEAR = "bilinear"
        BICUBIC = "bicubic"

    pil_modes = {InterpolationMode.NEAREST: Image.NEAREST, InterpolationMode.BILINEAR: Image.BILINEAR,
                 InterpolationMode.BICUBIC: Image.BICUBIC}

    def resize(img, size, interpolation=InterpolationMode.BILINEAR, max_size=None, antialias=True):
        assert isinstance(img, Image.Image), "stand-in covers the PIL dispatch only"
        return img.resize(tuple(size[::-1]), pil_modes[interpolation])

    class ToTensor:
        def __call__(self, pic):
            a = torch.from_numpy(np.array(pic, np.uint8, copy=True))
            a = a.view(pic.size[1], pic.size[0], len(pic.getbands())).permute(2, 0, 1).contiguous()
            return a.to(dtype=torch.float32).div(255)

    class Normalize:
        def __init__(self, mean, std, inplace=False):
            self.mean, self.std, self.inplace = mean, std, inplace

        def __call__(self, t):
            if not self.inplace:
                t = t.clone()
            mean = torch.as_tensor(self.mean, dtype=t.dtype)
            std = torch.as_tensor(self.std, dtype=t.dtype)
            return t.sub_(mean.view(-1, 1, 1)).div_(std.view(-1, 1, 1))

    import importlib.machinery
    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")
    fn = types.ModuleType("torchvision.transforms.functional")
    for m in (tv, tr, fn):
        m.__spec__ = importlib.machinery.ModuleSpec(m.__name__, None)
    fn.resize = resize
    tr.functional, tr.InterpolationMode, tr.ToTensor, tr.Normalize = fn, InterpolationMode, ToTensor, Normalize
    tv.transforms = tr
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tr, "torchvision.transforms.functional": fn,
                        "cv2": types.ModuleType("cv2")})


def main():
    from PIL import Image
    ref_env.activate()          # imports transformers, which probes for torchvision: the stand-ins come afterwards
    _install_standins()
    from data.transforms import ImageTransform, MaxLongEdgeMinShortEdgeResize    # the reference's classes, unmodified
    rng = np.random.default_rng(5)
    out = {"size_rule": [], "transform": []}
    # (1) the size rule over a grid of shapes / limits, incl. the app.py settings (vae 1024/512/16, vit 980/224/14)
    for (mx, mn, st, mp) in ((1024, 512, 16, 14 * 14 * 9 * 1024), (980, 224, 14, 14 * 14 * 9 * 1024), (518, 224, 14, 200_000),
                             (64, 32, 16, 3000)):
        r = MaxLongEdgeMinShortEdgeResize(max_size=mx, min_size=mn, stride=st, max_pixels=mp)
        for (w, h) in ((640, 480), (4000, 3000), (100, 1000), (1024, 1024), (37, 53), (1920, 1080), (300, 301), (5000, 200)):
            for n in (1, 2):
                probe = Image.new("RGB", (w, h))
                got = r(probe, img_num=n).size
                assert IO.target_size(w, h, mx, mn, st, mp, n) == got, ("size rule", w, h, mx, mn, st, mp, n)
                out["size_rule"].append(((w, h, mx, mn, st, mp, n), got))
    # (2) the whole transform on seeded images (small limits so the fixture stays small): down- and up-scaling
    for (w, h), (mx, mn, st) in (((150, 97), (64, 32, 16)), ((40, 23), (112, 56, 14)), ((96, 96), (96, 96, 16)),
                                 ((33, 120), (70, 28, 14))):
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        t = ImageTransform(mx, mn, st, max_pixels=14 * 14 * 9 * 1024)
        ref = t(Image.fromarray(a, "RGB"))
        nw, nh = IO.target_size(w, h, mx, mn, st, 14 * 14 * 9 * 1024)
        mine = torch.from_numpy(IO.to_tensor_normalize(IO.resize_bicubic_u8(a, nh, nw)))
        assert ref.dtype == torch.float32 and tuple(ref.shape) == (3, nh, nw)
        assert torch.equal(ref, mine), ("transform", w, h)
        out["transform"].append(dict(image=torch.from_numpy(a), limits=(mx, mn, st), out=ref))
    path = os.path.join(ROOT, "tests", "golden", "image_io.pt")
    torch.save(out, path)
    print(f"[golden] {path} ({os.path.getsize(path) / 1024:.0f} KiB): {len(out['size_rule'])} size-rule cases, "
          f"{len(out['transform'])} transforms; oracle == reference bit-exact")


if __name__ == "__main__":
    main()
