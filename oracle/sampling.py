"""CPU restatement of the device-side next-token sampler (bagel_amd/csrc/elementwise.hip ``sample_gumbel_kernel``, ``bagel_sample_gumbel_bf16``).

TEST INFRASTRUCTURE ONLY (imported by tests/): the product never imports it.

The reference samples with ``probs = softmax(pred_logits / temperature); curr_tokens = multinomial(probs, 1)`` (bagel.py:980-983) from torch's generator.  The
product draws from the SAME categorical distribution by the Gumbel-max identity, with its own counter-based generator so that the draw can live inside the captured
decode step:

    token[b] = argmax_i ( bf16(logit[b, i] / T) + g_i ),   g_i = -log(-log(u_i)),   u_i = ((x_i >> 9) + 0.5) * 2^-23  (23 bits + 0.5 is exact in fp32: u in [2^-24, 1 - 2^-24], the Gumbel value is always finite),
    (x_{4q}, .., x_{4q+3}) = Philox4x32-10(counter = (q, b, step, 0), key = (seed & 0xffffffff, seed >> 32))

(ties -> lowest index).  Philox4x32-10 is Salmon et al., "Parallel random numbers: as easy as 1, 2, 3" (SC'11), the generator curand / torch use as well; the
constants below are the paper's.  Parity is therefore "unpinned" against the reference's RNG STREAM by construction (no two devices share one: the reference says so
itself at bagel.py:980) and pinned against its DISTRIBUTION: tests compare the empirical frequencies with softmax(logits / T)."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over uint32 arrays c0..c3; scalar keys."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32).copy() for c in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xffffffff, int(k1) & 0xffffffff
    for _ in range(10):
        p0 = M0 * c0.astype(np.uint64)
        p1 = M1 * c2.astype(np.uint64)
        n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ np.uint32(k0)
        n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ np.uint32(k1)
        c0, c1, c2, c3 = n0, p1.astype(np.uint32), n2, p0.astype(np.uint32)
        k0, k1 = (k0 + W0) & 0xffffffff, (k1 + W1) & 0xffffffff
    return c0, c1, c2, c3


def bf16_round(x):
    """fp32 -> nearest-even bf16 -> fp32 (finite inputs)."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32)
    u = (u + np.uint32(0x7fff) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xffff0000)
    return u.view(np.float32)


def gumbel_of(x_u32):
    """uint32 draws -> Gumbel(0, 1) values, the device's ``gumbel_of``: 23 bits + 0.5 (exact in fp32), never 0 or 1, so never +-inf."""
    x = np.asarray(x_u32, dtype=np.uint32)
    u = ((x >> np.uint32(9)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -23)
    return -np.log(-np.log(u, dtype=np.float32), dtype=np.float32)


def gumbel_keys(logits_f32, temperature, seed, step):
    """[rows, cols] fp32 (bf16-valued) logits -> the perturbed scores the device maximises."""
    logits_f32 = np.asarray(logits_f32, dtype=np.float32)
    rows, cols = logits_f32.shape
    nq = (cols + 3) // 4
    out = np.empty((rows, nq * 4), dtype=np.float32)
    q = np.arange(nq, dtype=np.uint32)
    for b in range(rows):
        x = philox4x32_10(q, np.full(nq, b, np.uint32), np.full(nq, step, np.uint32), np.zeros(nq, np.uint32), seed & 0xffffffff, (seed >> 32) & 0xffffffff)
        xs = np.stack(x, 1).reshape(-1)
        out[b] = gumbel_of(xs)
    z = bf16_round(logits_f32 / np.float32(temperature))
    return z + out[:, :cols]


def sample_gumbel(logits_f32, temperature, seed, step=0):
    """-> int64 [rows]."""
    return np.argmax(gumbel_keys(logits_f32, temperature, seed, step), axis=1).astype(np.int64)
