"""Wall time of the real VAE's decode / encode of one 1024^2 image on the MI355X: fp32 (exact-fp32 MFMA) vs the bf16-autocast engine.
    python tools/vae_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bagel_amd.modeling.autoencoder import AutoEncoder, AutoEncoderParams  # noqa: E402
from oracle.configs import VAE_FULL  # noqa: E402
from oracle.weights import load_synth  # noqa: E402


def main():
    ae = AutoEncoder(AutoEncoderParams(**VAE_FULL["vae"]))
    load_synth(ae, 0)
    ae = ae.to("cuda").eval()
    g = torch.Generator().manual_seed(1)
    z = torch.randn(1, 16, 128, 128, generator=g).cuda()
    x = (torch.rand(1, 3, 1024, 1024, generator=g) * 2 - 1).cuda()
    noise = torch.randn(1, 16, 128, 128, generator=g)
    out = {}
    for prec in ("fp32", "bf16"):
        for what, fn in (("decode", lambda: ae.decode(z, precision=prec)), ("encode", lambda: ae.encode(x, sample_noise=noise, precision=prec))):
            fn(); fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                r = fn()
            torch.cuda.synchronize()
            out[(prec, what)] = (time.perf_counter() - t0) / 3 * 1e3
            assert torch.isfinite(r.float()).all()
    d32, d16 = ae.decode(z, precision="fp32"), ae.decode(z, precision="bf16")
    rel = float((d16.float() - d32).norm() / d32.norm())
    print(f"VAE 1024^2 (ch=128): decode fp32 {out[('fp32', 'decode')]:.1f} ms -> bf16 {out[('bf16', 'decode')]:.1f} ms; encode fp32 {out[('fp32', 'encode')]:.1f} ms -> "
          f"bf16 {out[('bf16', 'encode')]:.1f} ms; bf16 vs fp32 decode rel-L2 {rel:.2e}", flush=True)


if __name__ == "__main__":
    main()
