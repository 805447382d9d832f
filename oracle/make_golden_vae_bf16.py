"""Generate tests/golden/vae_full_bf16.pt: the REAL VAE (autoencoder.py:340-351: ch 128, ch_mult [1, 2, 4, 4], 2 res blocks, z 16) as the
reference's InterleaveInferencer runs it -- INSIDE ``torch.autocast(bf16)`` (inferencer.py:233 -> decode_image :174-185; the VAE-encode of
an edit request: forward_cache_update_vae under the same region, bagel.py:491-550).

TEST INFRASTRUCTURE ONLY.  Run in the build container (needs /root/reference):  python -m oracle.make_golden_vae_bf16

Two semantics, two goldens in one file (oracle/bagel_oracle.py VAE_AUTOCAST):
  * ``*_cpu``: the UNMODIFIED reference under ``torch.autocast("cpu", dtype=bfloat16)`` on this container.  The oracle in "cpu" mode must
    reproduce it BIT FOR BIT (asserted here): that pins the restatement's structure and every conv / SDPA / residual cast point.
  * ``*_cuda``: the oracle in "cuda" mode = the same restatement with the ONE policy difference of the device the reference actually runs
    on (group_norm is in CUDA autocast's fp32 list: statistics and result in fp32, swish on the fp32 tensor).  The MI355X bf16 VAE is
    tested against this one; its distance from the "cpu" golden is recorded (the GroupNorm rounding point, nothing else).
Scenario: decode of a 32 x 32 latent (256^2 image) and encode of a 256^2 image with a recorded reparameterisation draw."""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import bagel_oracle as O          # noqa: E402
from oracle import make_golden as MG          # noqa: E402
from oracle.configs import VAE_FULL           # noqa: E402


def scenario():
    """Runs the UNMODIFIED reference VAE under cpu autocast beside the oracle ON THIS HOST and raises unless the "cpu" policy reproduces it bit for bit
    (the host-independent pin: tests/test_reference_crosscheck.py calls this; nothing is written).  -> the fixture dict."""
    cfg = VAE_FULL
    MG.ref_env.activate()
    from modeling.autoencoder import AutoEncoder, AutoEncoderParams
    from oracle.weights import load_synth
    vae = AutoEncoder(AutoEncoderParams(**cfg["vae"]))
    load_synth(vae, MG.WEIGHT_SEED)
    vae = vae.eval()
    VW = {k: v for k, v in vae.state_dict().items()}
    g = torch.Generator().manual_seed(23)
    x = torch.rand(1, 3, 256, 256, generator=g) * 2 - 1
    z = torch.randn(1, cfg["vae"]["z_channels"], 32, 32, generator=g)
    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())  # noqa: E731
    with torch.no_grad():
        with torch.autocast("cpu", dtype=torch.bfloat16):
            dec = vae.decode(z)
            torch.manual_seed(47)
            enc = vae.encode(x)
        assert dec.dtype == torch.bfloat16 and enc.dtype == torch.bfloat16
        torch.manual_seed(47)
        noise = torch.randn(1, cfg["vae"]["z_channels"], 32, 32, dtype=torch.bfloat16)       # randn_like of the bf16 moments
        out = dict(x=x, z=z, enc_noise=noise.float(), decoded_cpu=dec, encoded_cpu=enc)
        O.VAE_AUTOCAST = "cpu"
        try:
            MG.same(dec, O.vae_decode(VW, cfg["vae"], z), "vae.decode under cpu autocast (oracle vs the unmodified reference)")
            MG.same(enc, O.vae_encode(VW, cfg["vae"], x, noise), "vae.encode under cpu autocast (oracle vs the unmodified reference)")
            O.VAE_AUTOCAST = "cuda"
            out["decoded_cuda"] = O.vae_decode(VW, cfg["vae"], z)
            out["encoded_cuda"] = O.vae_encode(VW, cfg["vae"], x, noise)
        finally:
            O.VAE_AUTOCAST = None
        dec32, enc32 = O.vae_decode(VW, cfg["vae"], z), O.vae_encode(VW, cfg["vae"], x, noise.float())
    out["distance"] = dict(decode_cuda_vs_cpu=rel(out["decoded_cuda"], dec), encode_cuda_vs_cpu=rel(out["encoded_cuda"], enc),
                           decode_cuda_vs_fp32=rel(out["decoded_cuda"], dec32), encode_cuda_vs_fp32=rel(out["encoded_cuda"], enc32),
                           decode_cpu_vs_fp32=rel(dec, dec32), encode_cpu_vs_fp32=rel(enc, enc32))
    out["host"] = dict(torch=torch.__version__)
    return out


def main():
    t0 = time.time()
    out = scenario()
    print("rel-L2 distances:", {k: f"{v:.3e}" for k, v in out["distance"].items()})
    path = os.path.join(MG.GOLD, "vae_full_bf16.pt")
    torch.save(out, path)
    print(f"wrote {path} ({os.path.getsize(path) / 1e6:.1f} MB) in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
