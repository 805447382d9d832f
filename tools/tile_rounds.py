"""Tile quantisation of the denoise forward on the 256-CU chip (no GPU needed): for every projection of a MoT layer, the
256x256 tiles the persistent GEMM walks, the rounds of its 256 workgroups they make, and what is paid for (whole rounds) --
sequential CFG forwards (today's default), stream-batched, and stream-batched with the marker rows on the dense side path
(DESIGN.md 3.7).  Also the live-row occupancy of the attention kernel's 256-row query tiles.

    python tools/tile_rounds.py [--batch 4] [--res 1024] [--streams 2]"""
import argparse
import math

H, I, NQ, NKV, D = 3584, 18944, 28, 4, 128
CUS = 256


def rounds(m_gen, m_und, n_cols, tile_n=256):
    tiles = (math.ceil(m_gen / 256) + (math.ceil(m_und / 256) if m_und else 0)) * math.ceil(n_cols / tile_n)
    return tiles, tiles / CUS, math.ceil(tiles / CUS)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--streams", type=int, default=2)
    a = ap.parse_args()
    n_img = (a.res // 16) ** 2
    gen, und = a.batch * n_img, a.batch * 2
    shapes = [("qkv", (NQ + 2 * NKV) * D, H), ("o", H, NQ * D), ("gate+up", 2 * I, H), ("down", H, I)]
    print(f"{a.batch} samples x {n_img} latent rows (+2 marker rows each), {a.streams} CFG streams per step; K-weighted cost = rounds x K")
    print(f"{'GEMM':8} {'mode':28} {'tiles':>7} {'rounds':>8} {'paid':>5} {'fill':>6}")
    total = {}
    for name, n, k in shapes:
        modes = [("sequential (x streams)", rounds(gen, und, n), a.streams),
                 ("batched", rounds(a.streams * gen, a.streams * und, n), 1),
                 ("batched + marker side path", rounds(a.streams * gen, 0, n), 1)]
        for mode, (t, r, p), mult in modes:
            print(f"{name:8} {mode:28} {t * mult:7d} {r * mult:8.2f} {p * mult:5d} {r / p:6.1%}")
            total[mode] = total.get(mode, 0.0) + p * mult * k
    base = total["sequential (x streams)"]
    for mode, v in total.items():
        print(f"layer GEMM cost, {mode:28}: {v / base:6.3f} x")
    lq = n_img + 2
    qt = math.ceil(lq / 256)
    for mode, samples in (("sequential (per forward)", a.batch), ("batched", a.batch * a.streams)):
        wgs = samples * NQ * qt
        full = samples * NQ * (qt - 1 if lq - (qt - 1) * 256 < 32 else qt)
        print(f"attention workgroups (one per CU at a time), {mode:26}: {wgs:5d} = {wgs / CUS:6.2f} rounds (paid {math.ceil(wgs / CUS)}); "
              f"full-cost ones {full} = {full / CUS:.2f} rounds")
    print(f"attention: {qt} query tiles of 256 rows per (sample, head); the last one has {lq - (qt - 1) * 256} live rows "
          f"-> {a.batch * a.streams * NQ * qt} workgroups per step-forward pair, {a.batch * a.streams * NQ} of them nearly empty "
          f"({1 / qt:.1%} of the tile count)")


if __name__ == "__main__":
    main()
