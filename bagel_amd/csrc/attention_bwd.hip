// Reverse of the block-masked packed attention of the training forward (PackedAttentionMoT.forward_train, qwen2_navit.py:406-497, with
// the causal / full / noise split mask of data/data_utils.py:72-103) for gfx950, head_dim 64 or 128, GQA.  The reference differentiates
// flash-attn / flex-attention through autograd; this is the hand-written counterpart of the forward in attention.hip, same conventions:
// products on v_mfma_f32_32x32x16_bf16 with the TRANSPOSED score tile (rows = keys of the A operand, columns = the lane's query), so a
// lane owns all scores of one query (or one key) and the bf16 P / dS values go from the accumulator registers straight into the B
// operand of the next product; the MFMA row -> key assignment of every 16 rows is permuted (quads 1 <-> 2) so that a lane's 8
// contraction slots are 8 consecutive keys and the other operand is ONE 16-byte LDS read of a transposed tile.
//
// Two kernels, both deterministic (no atomics; a row's gradient is produced by exactly one wave):
//   attn_bwd_dq_kernel   workgroup = one 128-query item x one q head, wave = 32 queries.
//                        pass 1: S^T = K Q^T over the item's key tiles -> row log-sum-exp (online max / sum); delta = rowsum(dO o).
//                        pass 2: P^T = exp(scale S^T - lse), dP^T = V dO^T, dS^T = P^T (dP^T - delta), dQ^T += K^T dS^T (x scale at the store).
//   attn_bwd_dkv_kernel  workgroup = one 128-key item x one kv head, wave = 32 keys; loops over the group's q heads and over the 64-query
//                        tiles of the rows that see these keys:  P = exp(scale Q K^T - lse),  dV^T += dO^T P,  dS = P (dO V^T - delta),
//                        dK^T += Q^T dS.
// The contraction over keys (dq) / queries (dkv) reads K^T / Q^T / dO^T from transposed HBM images (bagel_transpose_bf16) -- the same
// trick as the forward's V^T image: 2 bytes per element per pass is noise next to the 4 + 4 products of the reverse.
// The mask is evaluated from per-item scalars (sample start, split start / end, causal flag) and one 64-bit noise-key word per key
// tile: full / noise splits see the clean prefix of their sample and themselves, causal splits additionally only keys <= the query,
// and keys of a noise split are hidden from every later split.
//
// Algorithmic FLOPs per (query, key) pair and head: 2 D (scores for lse) + 6 D (dq pass) + 8 D (dkv pass) = 16 D, against 4 D of the
// forward and 10 D of a fused flash backward: the price of two atomics-free kernels and of recomputing the row statistics instead of
// changing the forward kernel's interface (the training forward hands the statistics over: lse_from_forward).  Tiles are staged
// global -> registers -> LDS (two barriers per tile; not yet the LDS-DMA ring of the forward): in the dkv kernel (one wave per SIMD,
// its accumulators fill the register file) with the loads of step t + 1 issued before the products of step t, in the dq kernel with
// two workgroups per CU covering each other's latencies; tiles that every row of the item sees completely
// take a mask-free path; exponentials in base 2 with scale * log2(e) and the row's log-sum-exp folded into ONE fma per score (the lse workspace
// holds log2 values; the build runs with -ffp-contract=off, so the fma is spelled out: one VALU instruction less per score in kernels whose lone
// wave is issue-bound).
#include "common.h"
#include <stdlib.h>

struct AttnBwdParams {
    const bf16_t* q; long ldq;
    const bf16_t* k; long ldk;
    const bf16_t* v; long ldv;
    const bf16_t* o; long ldo;
    const bf16_t* d_o; long lddo;
    const bf16_t* qt; const bf16_t* dot; const bf16_t* kt; long ld_t;
    bf16_t* dq; long lddq;
    bf16_t* dk; long lddk;
    bf16_t* dv; long lddv;
    const int* q_items; const int* k_items;
    const unsigned long long* noise_bits;
    float* lse; float* delta;            // [nq][rows]
    int rows, nq, nkv, have_lse, n_q_items, n_k_items;
    float scale;
};

// Timing-only ablations (tools/ab_attn_bwd.sh builds one library per bit set; results are wrong by construction, never in the product
// build): 1 no lse pass, 2 no exp / mask arithmetic, 4 tile loads only once, 8 no barriers, 16 no LDS stores, 32 no dQ / dK / dV products,
// 64 no score products.
#ifndef AB_ABL
#define AB_ABL 0
#endif
#define AB_TP 72                          // element pitch of the transposed 64-column tiles (144 bytes: 16-byte aligned, banks spread)

// XCD-aware work mapping (workgroups are dealt to the 8 XCDs round-robin by their linear id): every kv head's work lands on the same
// XCD(s), so the K / V / K^T (Q / dO / Q^T / dO^T) tiles that the items and q heads of one kv head re-read stay in that XCD's 4 MB L2
// instead of being streamed through all eight.  nkv < 8: 8 / nkv XCDs per kv head share its items; nkv >= 8: kv heads g, g + 8, ... share
// XCD g.  `per` = the units of one kv head in one item (its q heads for the dq kernel, 1 for the dkv kernel).
// Returns false for the padding workgroups of the rounded-up grid.
__device__ __forceinline__ bool ab_map(int n_items, int nkv, int per, int& item, int& hkv, int& sub) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    if (nkv < 8 && 8 % nkv == 0) {
        const int xpg = 8 / nkv;                           // XCDs per kv head
        hkv = xcd / xpg;
        const int j = slot * xpg + (xcd % xpg);
        item = j / per; sub = j % per;
        return item < n_items;
    }
    if (nkv < 8) {                                         // 3, 5, 6, 7 kv heads: no even split of the XCDs -- plain round-robin
        const int u = blockIdx.x;
        hkv = u % nkv;
        item = (u / nkv) / per; sub = (u / nkv) % per;
        return item < n_items;
    }
    const int hpx = (nkv + 7) / 8;                         // kv heads per XCD
    const int j = slot;
    const int hl = j % hpx;                                // which of this XCD's kv heads
    hkv = xcd + 8 * hl;
    const int r = j / hpx;
    item = r / per; sub = r % per;
    return item < n_items && hkv < nkv;
}
__host__ static inline int ab_grid(int n_items, int nkv, int per) {
    if (nkv < 8 && 8 % nkv == 0) { const int xpg = 8 / nkv; return 8 * ((n_items * per + xpg - 1) / xpg); }
    if (nkv < 8) return (n_items * per * nkv + 7) / 8 * 8;
    return 8 * (n_items * per * ((nkv + 7) / 8));
}

__device__ __forceinline__ int ab_perm(int m) { return (m & ~12) | ((m & 4) << 1) | ((m & 8) >> 1); }

// 64 rows x D columns of a row-major operand: global -> registers (rows at or beyond rows_total repeat the last row: finite values that
// the mask turns into exact zeros -- no branch around the load), registers -> LDS [64][D + 8].
// The two halves are separate so that the loads of tile t + 1 are in flight while tile t is being multiplied.
template <int D>
struct AbRows { u32x4_t r[(64 * (D / 8)) / 256]; };
template <int D>
__device__ __forceinline__ void ab_load_rows(AbRows<D>& x, const bf16_t* src, long ld, int col0, int row0, int rows_total, int tid) {
    constexpr int CH = D / 8;
#pragma unroll
    for (int it = 0; it < (64 * CH) / 256; ++it) {
        const int idx = tid + it * 256;
        const int r = idx / CH, c = idx % CH;
        x.r[it] = *(const u32x4_t*)(src + (long)min(row0 + r, rows_total - 1) * ld + col0 + c * 8);
    }
}
template <int D>
__device__ __forceinline__ void ab_store_rows(bf16_t* lds, const AbRows<D>& x, int tid) {
    constexpr int CH = D / 8, RP = D + 8;
#pragma unroll
    for (int it = 0; it < (64 * CH) / 256; ++it) {
        const int idx = tid + it * 256;
        *(u32x4_t*)(lds + (idx / CH) * RP + (idx % CH) * 8) = x.r[it];
    }
}

// D rows x 64 columns of a transposed image -> registers -> LDS [D][AB_TP]
template <int D>
struct AbT { u32x4_t r[(D * 8) / 256]; };
template <int D>
__device__ __forceinline__ void ab_load_t(AbT<D>& x, const bf16_t* srct, long ld_t, int drow0, int col0, int tid) {
#pragma unroll
    for (int it = 0; it < (D * 8) / 256; ++it) {
        const int idx = tid + it * 256;
        x.r[it] = *(const u32x4_t*)(srct + (long)(drow0 + (idx >> 3)) * ld_t + col0 + (idx & 7) * 8);
    }
}
template <int D>
__device__ __forceinline__ void ab_store_t(bf16_t* lds, const AbT<D>& x, int tid) {
#pragma unroll
    for (int it = 0; it < (D * 8) / 256; ++it) {
        const int idx = tid + it * 256;
        *(u32x4_t*)(lds + (idx >> 3) * AB_TP + (idx & 7) * 8) = x.r[it];
    }
}

__device__ __forceinline__ bf16x8_t ab_pack8(const float* x) {
    u32x4_t v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = pack2bf(x[2 * e], x[2 * e + 1]);
    return __builtin_bit_cast(bf16x8_t, v);
}

// fragment of 8 consecutive d-columns of one row of a [rows, heads * D] operand in HBM (zero for an invalid row)
__device__ __forceinline__ bf16x8_t ab_row_frag(const bf16_t* base, long ld, int row, bool valid, int col) {
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (valid) v = *(const u32x4_t*)(base + (long)row * ld + col);
    return __builtin_bit_cast(bf16x8_t, v);
}

// C-layout (rows d = 32 db + (i & 3) + 8 (i >> 2) + 4 h, column = the lane's row of the output) -> out[row][col0 + d], 8-byte stores
template <int DB>
__device__ __forceinline__ void ab_store_acc(bf16_t* out, long ld, int row, int col0, const f32x16_t* acc, int h, float f = 1.0f) {
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u32x2_t v;
            v[0] = pack2bf(acc[db][4 * j] * f, acc[db][4 * j + 1] * f);
            v[1] = pack2bf(acc[db][4 * j + 2] * f, acc[db][4 * j + 3] * f);
            *(u32x2_t*)(out + (long)row * ld + col0 + 32 * db + 8 * j + 4 * h) = v;
        }
}

// Two waves per SIMD (two workgroups per CU: 223 registers, 2 x 52 KB of LDS): one workgroup's exponentials and tile staging run under the
// other's products -- measured 4.28 -> 2.86 ms at the probe's shapes against the one-wave form with a register prefetch of the next tile
// (profiles/r03_attn_bwd_ablations.log), so this kernel stages each tile global -> registers -> LDS right where it needs it.
template <int D>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const AttnBwdParams p) {
    constexpr int KS = D / 16, DB = D / 32, RP = D + 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* Ks = (bf16_t*)smem;                        // [64][RP]
    bf16_t* Vs = Ks + 64 * RP;                         // [64][RP]
    bf16_t* Kts = Vs + 64 * RP;                        // [D][AB_TP]
    int item_, hkv, sub_;
    if (!ab_map(p.n_q_items, p.nkv, p.nq / p.nkv, item_, hkv, sub_)) return;
    const int* it = p.q_items + (long)item_ * 8;
    const int row0 = it[0], nrows = it[1], kstart = it[2], sstart = it[3], send = it[4], causal = it[5], t0 = it[6], t1 = it[7];
    const int hq = hkv * (p.nq / p.nkv) + sub_;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, h = lane >> 5, pm = ab_perm(m);
    const int qloc = 32 * wave + m;
    const bool qvalid = qloc < nrows;
    const bool wave_on = 32 * wave < nrows;
    const int qrow = row0 + qloc;
    const float c2 = p.scale * 1.4426950408889634f;    // scores go through exp2: scale * log2(e) folded into one multiply

    bf16x8_t qf[KS], dof[KS];
    float delta = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int col = hq * D + 16 * ks + 8 * h;
        qf[ks] = ab_row_frag(p.q, p.ldq, qrow, qvalid, col);
        dof[ks] = ab_row_frag(p.d_o, p.lddo, qrow, qvalid, col);
        const u32x4_t ov = __builtin_bit_cast(u32x4_t, ab_row_frag(p.o, p.ldo, qrow, qvalid, col));
        const u32x4_t dv = __builtin_bit_cast(u32x4_t, dof[ks]);
#pragma unroll
        for (int e = 0; e < 4; ++e) delta += lo2f(ov[e]) * lo2f(dv[e]) + hi2f(ov[e]) * hi2f(dv[e]);
    }
    delta += __shfl_xor(delta, 32, 64);                 // the two halves of the wave hold the two halves of every 16 columns

    // a tile of hidden (noise) context keys is skipped; a tile that every query of the item sees completely needs no mask
    auto skip = [&](int t) { return p.noise_bits[t] == ~0ull && 64 * t + 64 <= sstart; };
    auto next_tile = [&](int t) { ++t; while (t < t1 && skip(t)) ++t; return t; };
    auto plain = [&](int t, unsigned long long nb) {
        const int c0 = 64 * t, c1 = c0 + 64;
        const bool in_ctx = c0 >= kstart && c1 <= sstart && nb == 0ull;
        const bool in_own = c0 >= sstart && c1 <= send && (!causal || c1 - 1 <= row0);
        return (in_ctx || in_own) && nrows == 128;
    };
    // branch-free mask of the lane's 16 scores of key block kb (bit i <-> accumulator register i <-> key 32 kb + 16 (i >> 3) + 8 h + (i & 7))
    auto mask16 = [&](int t, unsigned long long nb, int kb) {
        const unsigned w = (unsigned)(nb >> (32 * kb)) >> (8 * h);          // noise bits of the lane's keys at positions 16 a + e
        unsigned ok = 0u;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = 64 * t + 32 * kb + 16 * (i >> 3) + 8 * h + (i & 7);
            const unsigned nbit = (w >> (16 * (i >> 3) + (i & 7))) & 1u;
            const unsigned ctx = (unsigned)(c < sstart) & (nbit ^ 1u);
            const unsigned own = (unsigned)(c >= sstart) & (unsigned)(c < send) & ((unsigned)(causal == 0) | (unsigned)(c <= qrow));
            ok |= ((unsigned)(c >= kstart) & (ctx | own)) << i;
        }
        return qvalid ? ok : 0u;
    };
    const int tfirst = next_tile(t0 - 1);

    // ---------------- pass 1: row log-sum-exp (base 2) ----------------
    float mrun = -INFINITY, lrun = 0.f;
    if (!(AB_ABL & 1) && !p.have_lse) {
        AbRows<D> rk;
        for (int t = tfirst; t < t1;) {
            if (!(AB_ABL & 4) || t == tfirst) ab_load_rows<D>(rk, p.k, p.ldk, hkv * D, 64 * t, p.rows, tid);
            if (!(AB_ABL & 8)) __syncthreads();
            if (!(AB_ABL & 16)) ab_store_rows<D>(Ks, rk, tid);
            if (!(AB_ABL & 8)) __syncthreads();
            const int tn = next_tile(t);
            if (wave_on) {
                const unsigned long long nb = p.noise_bits[t];
                const bool pl = plain(t, nb);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    f32x16_t s;
#pragma unroll
                    for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
                        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8_t*)(Ks + (32 * kb + pm) * RP + 16 * ks + 8 * h), qf[ks], s, 0, 0, 0);
                    const unsigned ok = pl ? 0xffffu : mask16(t, nb, kb);
                    float mx = -INFINITY;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        s[i] = ((ok >> i) & 1u) ? s[i] * c2 : -INFINITY;
                        mx = fmaxf(mx, s[i]);
                    }
                    const float mnew = fmaxf(mrun, mx);
                    if (mnew > -INFINITY) {
                        float sum = 0.f;
#pragma unroll
                        for (int i = 0; i < 16; ++i) sum += __builtin_amdgcn_exp2f(s[i] - mnew);
                        lrun = lrun * __builtin_amdgcn_exp2f(mrun - mnew) + sum;
                        mrun = mnew;
                    }
                }
            }
            t = tn;
        }
    }
    float lse2 = 0.f;                                   // log2 of the softmax denominator, in the scaled base-2 domain
    {
        const float mo = __shfl_xor(mrun, 32, 64), lo = __shfl_xor(lrun, 32, 64);
        const float mm = fmaxf(mrun, mo);
        if (mm > -INFINITY) {
            const float ll = (mrun > -INFINITY ? lrun * exp2f(mrun - mm) : 0.f) + (mo > -INFINITY ? lo * exp2f(mo - mm) : 0.f);
            lse2 = mm + log2f(ll);
        }
    }
    if (p.have_lse) lse2 = qvalid ? p.lse[(long)hq * p.rows + qrow] : 0.f;      // left by the forward (bagel_attn_varlen_ranges_lse_bf16)
    if (qvalid && h == 0) {
        if (!p.have_lse) p.lse[(long)hq * p.rows + qrow] = lse2;
        p.delta[(long)hq * p.rows + qrow] = delta;
    }

    // ---------------- pass 2: dQ ----------------
    f32x16_t dq[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) dq[db][i] = 0.f;
    {
        AbRows<D> rk, rv;
        AbT<D> rt;
        auto load = [&](int t) {
            ab_load_rows<D>(rk, p.k, p.ldk, hkv * D, 64 * t, p.rows, tid);
            ab_load_rows<D>(rv, p.v, p.ldv, hkv * D, 64 * t, p.rows, tid);
            ab_load_t<D>(rt, p.kt, p.ld_t, hkv * D, 64 * t, tid);
        };
        for (int t = tfirst; t < t1;) {
            if (!(AB_ABL & 4) || t == tfirst) load(t);
            if (!(AB_ABL & 8)) __syncthreads();
            if (!(AB_ABL & 16)) {
                ab_store_rows<D>(Ks, rk, tid);
                ab_store_rows<D>(Vs, rv, tid);
                ab_store_t<D>(Kts, rt, tid);
            }
            if (!(AB_ABL & 8)) __syncthreads();
            const int tn = next_tile(t);
            if (wave_on) {
                const unsigned long long nb = p.noise_bits[t];
                const bool pl = plain(t, nb);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    f32x16_t s, dp;
#pragma unroll
                    for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        if (AB_ABL & 64) { s[ks] = delta * (float)t; dp[ks] = lse2; continue; }
                        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8_t*)(Ks + (32 * kb + pm) * RP + 16 * ks + 8 * h), qf[ks], s, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8_t*)(Vs + (32 * kb + pm) * RP + 16 * ks + 8 * h), dof[ks], dp, 0, 0, 0);
                    }
                    float ds[16];                            // dS / scale: the softmax scale multiplies the dQ accumulators once, at the store
                    if (pl) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            if (AB_ABL & 2) { ds[i] = s[i] + dp[i]; continue; }
                            ds[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[i], c2, -lse2)) * (dp[i] - delta);
                        }
                    } else {
                        const unsigned ok = mask16(t, nb, kb);
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            if (AB_ABL & 2) { ds[i] = s[i] + dp[i]; continue; }
                            const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[i], c2, -lse2));
                            ds[i] = (((ok >> i) & 1u) ? e : 0.f) * (dp[i] - delta);
                        }
                    }
                    const bf16x8_t dsf0 = ab_pack8(ds), dsf1 = ab_pack8(ds + 8);
#pragma unroll
                    for (int db = 0; db < DB; ++db) {
                        if (AB_ABL & 32) { dq[db][0] += ds[db] + ds[db + 8]; continue; }
                        dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8_t*)(Kts + (32 * db + m) * AB_TP + 32 * kb + 8 * h), dsf0, dq[db], 0, 0, 0);
                        dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8_t*)(Kts + (32 * db + m) * AB_TP + 32 * kb + 16 + 8 * h), dsf1, dq[db], 0, 0, 0);
                    }
                }
            }
            t = tn;
        }
    }
    if (qvalid) ab_store_acc<DB>(p.dq, p.lddq, qrow, hq * D, dq, h, p.scale);
}

template <int D>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const AttnBwdParams p) {
    constexpr int KS = D / 16, DB = D / 32, RP = D + 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* Qs = (bf16_t*)smem;                        // [64][RP]
    bf16_t* dOs = Qs + 64 * RP;                        // [64][RP]
    bf16_t* Qts = dOs + 64 * RP;                       // [D][AB_TP]
    bf16_t* dOts = Qts + D * AB_TP;                    // [D][AB_TP]
    float* lse_s = (float*)(dOts + D * AB_TP);         // [64]
    float* delta_s = lse_s + 64;                       // [64]
    int item_, hkv, sub_;
    if (!ab_map(p.n_k_items, p.nkv, 1, item_, hkv, sub_)) return;
    const int* it = p.k_items + (long)item_ * 8;
    const int key0 = it[0], nkeys = it[1], qbeg = it[2], qend = it[3], send = it[4], causal = it[5];
    const int G = p.nq / p.nkv;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, h = lane >> 5, pm = ab_perm(m);
    const int kloc = 32 * wave + m;
    const bool kvalid = kloc < nkeys;
    const bool wave_on = 32 * wave < nkeys;
    const int krow = key0 + kloc;
    const float c2 = p.scale * 1.4426950408889634f;

    bf16x8_t kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        kf[ks] = ab_row_frag(p.k, p.ldk, krow, kvalid, hkv * D + 16 * ks + 8 * h);
        vf[ks] = ab_row_frag(p.v, p.ldv, krow, kvalid, hkv * D + 16 * ks + 8 * h);
    }
    f32x16_t dk[DB], dv[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dk[db][i] = 0.f; dv[db][i] = 0.f; }

    // the (q head, 64-query tile) pairs of this item as one flat sequence, so that the loads of the next pair fly under the current products
    const int q00 = qbeg & ~63;
    const int ntile = qend > q00 ? (qend - q00 + 63) >> 6 : 0;
    const int nstep = ntile * G;
    AbRows<D> rq, rdo;
    AbT<D> rqt, rdot;
    float rstat = 0.f;
    auto load = [&](int step) {
        const int hq = hkv * G + step / ntile, q0 = q00 + 64 * (step % ntile);
        ab_load_rows<D>(rq, p.q, p.ldq, hq * D, q0, p.rows, tid);
        ab_load_rows<D>(rdo, p.d_o, p.lddo, hq * D, q0, p.rows, tid);
        ab_load_t<D>(rqt, p.qt, p.ld_t, hq * D, q0, tid);
        ab_load_t<D>(rdot, p.dot, p.ld_t, hq * D, q0, tid);
        if (tid < 128) {
            const int r = q0 + (tid & 63);
            const float* src = tid < 64 ? p.lse : p.delta;
            rstat = r < p.rows ? src[(long)hq * p.rows + r] : 0.f;
        }
    };
    if (nstep > 0) load(0);
    for (int step = 0; step < nstep; ++step) {
        const int q0 = q00 + 64 * (step % ntile);
        if (!(AB_ABL & 8)) __syncthreads();
        if (!(AB_ABL & 16)) {
            ab_store_rows<D>(Qs, rq, tid);
            ab_store_rows<D>(dOs, rdo, tid);
            ab_store_t<D>(Qts, rqt, tid);
            ab_store_t<D>(dOts, rdot, tid);
            if (tid < 128) lse_s[tid] = rstat;          // lse_s and delta_s are adjacent: [0, 64) lse, [64, 128) delta
        }
        if (!(AB_ABL & 8)) __syncthreads();
        if (step + 1 < nstep && !(AB_ABL & 4)) load(step + 1);
        if (!wave_on) continue;
        // every query of the tile sees every key of the item: rows of later splits, or of the own full / noise split
        const bool pl = nkeys == 128 && q0 >= qbeg && q0 + 64 <= qend && (q0 >= send || !causal);
#pragma nounroll                                          // one 32-query block at a time: both unrolled would not fit the register file
        for (int qb = 0; qb < 2; ++qb) {
            f32x16_t s, dp;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
            // ALL fragment reads of a product group go out before its first MFMA (sched_barrier keeps hipcc from sinking each read next to its
            // use: it emitted read, read, wait, MFMA, MFMA).  Round 4: 4.96 -> 4.75 ms at the probe's shapes -- the reads were NOT the main loss;
            // what else was tried on this kernel and did not pay is in profiles/r04_attn_bwd_experiments.log (role-split waves, role-split workgroups)
            bf16x8_t fa[KS], fb[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                fa[ks] = *(const bf16x8_t*)(Qs + (32 * qb + pm) * RP + 16 * ks + 8 * h);
                fb[ks] = *(const bf16x8_t*)(dOs + (32 * qb + pm) * RP + 16 * ks + 8 * h);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (AB_ABL & 64) { s[ks] = (float)step; dp[ks] = (float)q0; continue; }
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks], kf[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks], vf[ks], dp, 0, 0, 0);
            }
            // the transposed fragments of the gradient products land while the softmax arithmetic runs
            bf16x8_t ft[DB][2], fu[DB][2];
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const bf16_t* a0 = dOts + (32 * db + m) * AB_TP + 32 * qb + 8 * h;
                const bf16_t* b0 = Qts + (32 * db + m) * AB_TP + 32 * qb + 8 * h;
                ft[db][0] = *(const bf16x8_t*)a0; ft[db][1] = *(const bf16x8_t*)(a0 + 16);
                fu[db][0] = *(const bf16x8_t*)b0; fu[db][1] = *(const bf16x8_t*)(b0 + 16);
            }
            __builtin_amdgcn_sched_barrier(0);
            float pr[16], ds[16], lv[16], dl[16];
#pragma unroll
            for (int a = 0; a < 2; ++a)                      // the lane's 16 queries are two runs of 8: four 16-byte LDS reads each
#pragma unroll
                for (int e4 = 0; e4 < 2; ++e4) {
                    const f32x4_t x = *(const f32x4_t*)(lse_s + 32 * qb + 16 * a + 8 * h + 4 * e4);
                    const f32x4_t y = *(const f32x4_t*)(delta_s + 32 * qb + 16 * a + 8 * h + 4 * e4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { lv[8 * a + 4 * e4 + e] = x[e]; dl[8 * a + 4 * e4 + e] = y[e]; }
                }
            if (pl) {                                        // dS / scale: the softmax scale multiplies the dK accumulators once, at the store
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (AB_ABL & 2) { pr[i] = s[i]; ds[i] = dp[i]; continue; }
                    pr[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[i], c2, -lv[i]));
                    ds[i] = pr[i] * (dp[i] - dl[i]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int ql = 32 * qb + 16 * (i >> 3) + 8 * h + (i & 7);
                    const int qrow = q0 + ql;
                    const unsigned ok = (unsigned)kvalid & (unsigned)(qrow >= qbeg) & (unsigned)(qrow < qend) &
                                        ((unsigned)(qrow >= send) | (unsigned)(causal == 0) | (unsigned)(krow <= qrow));
                    if (AB_ABL & 2) { pr[i] = s[i]; ds[i] = dp[i]; continue; }
                    const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[i], c2, -lv[i]));
                    pr[i] = ok ? e : 0.f;
                    ds[i] = pr[i] * (dp[i] - dl[i]);
                }
            }
            const bf16x8_t pf0 = ab_pack8(pr), pf1 = ab_pack8(pr + 8), dsf0 = ab_pack8(ds), dsf1 = ab_pack8(ds + 8);
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                if (AB_ABL & 32) { dv[db][0] += pr[db] + pr[db + 8]; dk[db][0] += ds[db] + ds[db + 8]; continue; }
                dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ft[db][0], pf0, dv[db], 0, 0, 0);
                dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fu[db][0], dsf0, dk[db], 0, 0, 0);
                dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ft[db][1], pf1, dv[db], 0, 0, 0);
                dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fu[db][1], dsf1, dk[db], 0, 0, 0);
            }
        }
    }
    if (kvalid) {
        ab_store_acc<DB>(p.dk, p.lddk, krow, hkv * D, dk, h, p.scale);
        ab_store_acc<DB>(p.dv, p.lddv, krow, hkv * D, dv, h);
    }
}

template <int D>
static int attn_bwd_launch(const AttnBwdParams& p, int n_q_items, int n_k_items, hipStream_t stream) {
    constexpr int smem_dq = (2 * 64 * (D + 8) + D * AB_TP) * 2;
    constexpr int smem_dkv = (2 * 64 * (D + 8) + 2 * D * AB_TP) * 2 + 128 * 4;
    if (int rc = bagel_enable_lds((const void*)attn_bwd_dq_kernel<D>, smem_dq, "attn_bwd_dq_kernel")) return rc;
    if (int rc = bagel_enable_lds((const void*)attn_bwd_dkv_kernel<D>, smem_dkv, "attn_bwd_dkv_kernel")) return rc;
#if AB_ABL || defined(BAGEL_ENABLE_ABLATIONS)
    if (const char* only = getenv("BAGEL_ABWD_ONLY")) { if (only[1] == 'q') n_k_items = 0; else n_q_items = 0; }    // "dq" | "dkv"
#endif
    if (n_q_items > 0) {
        hipLaunchKernelGGL((attn_bwd_dq_kernel<D>), dim3(ab_grid(n_q_items, p.nkv, p.nq / p.nkv)), dim3(256), smem_dq, stream, p);
        if (int rc = bagel_check_launch("attn_bwd_dq_kernel")) return rc;
    }
    if (n_k_items > 0) {
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<D>), dim3(ab_grid(n_k_items, p.nkv, 1)), dim3(256), smem_dkv, stream, p);
        if (int rc = bagel_check_launch("attn_bwd_dkv_kernel")) return rc;
    }
    return BAGEL_OK;
}

extern "C" int bagel_attn_bwd_blockmask_bf16(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* o,
                                             int64_t ldo, const void* d_o, int64_t lddo, const void* qt, const void* dot, const void* kt,
                                             int64_t ld_t, void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv,
                                             const int32_t* q_items, int32_t n_q_items, const int32_t* k_items, int32_t n_k_items,
                                             const uint64_t* noise_bits, float* lse_delta, int32_t lse_from_forward, int32_t rows, int32_t nq,
                                             int32_t nkv, int32_t head_dim, float softmax_scale, hipStream_t stream) {
    BAGEL_REQUIRE(q && k && v && o && d_o && qt && dot && kt && dq && dk && dv && noise_bits && lse_delta, "attn_bwd: null pointer");
    BAGEL_REQUIRE((n_q_items == 0 || q_items) && (n_k_items == 0 || k_items), "attn_bwd: item tables missing");
    BAGEL_REQUIRE(nq > 0 && nkv > 0 && nq % nkv == 0, "attn_bwd: nq must be a multiple of nkv");
    BAGEL_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0 && lddo % 8 == 0 && ld_t % 8 == 0 && lddq % 4 == 0 && lddk % 4 == 0 && lddv % 4 == 0,
                  "attn_bwd: leading dimensions must be multiples of 8");
    BAGEL_REQUIRE(ld_t >= ((int64_t)rows + 63) / 64 * 64, "attn_bwd: the transposed images must hold ceil64(rows) columns");
    for (const void* ptr : {q, k, v, o, d_o, qt, dot, kt})
        BAGEL_REQUIRE(((uintptr_t)ptr % 16) == 0, "attn_bwd: 16-byte aligned operands expected");
    for (const void* ptr : {(const void*)dq, (const void*)dk, (const void*)dv})
        BAGEL_REQUIRE(((uintptr_t)ptr % 8) == 0, "attn_bwd: 8-byte aligned gradient buffers expected");
    if (rows <= 0) return BAGEL_OK;
    AttnBwdParams p;
    p.q = (const bf16_t*)q; p.ldq = ldq; p.k = (const bf16_t*)k; p.ldk = ldk; p.v = (const bf16_t*)v; p.ldv = ldv;
    p.o = (const bf16_t*)o; p.ldo = ldo; p.d_o = (const bf16_t*)d_o; p.lddo = lddo;
    p.qt = (const bf16_t*)qt; p.dot = (const bf16_t*)dot; p.kt = (const bf16_t*)kt; p.ld_t = ld_t;
    p.dq = (bf16_t*)dq; p.lddq = lddq; p.dk = (bf16_t*)dk; p.lddk = lddk; p.dv = (bf16_t*)dv; p.lddv = lddv;
    p.q_items = q_items; p.k_items = k_items; p.noise_bits = (const unsigned long long*)noise_bits;
    p.lse = lse_delta; p.delta = lse_delta + (long)nq * rows;
    p.rows = rows; p.nq = nq; p.nkv = nkv; p.have_lse = lse_from_forward; p.scale = softmax_scale;
    p.n_q_items = n_q_items; p.n_k_items = n_k_items;
    if (head_dim == 128) return attn_bwd_launch<128>(p, n_q_items, n_k_items, stream);
    if (head_dim == 64) return attn_bwd_launch<64>(p, n_q_items, n_k_items, stream);
    return bagel_set_error(BAGEL_ERR_UNSUPPORTED, "attn_bwd: head_dim %d not in {64,128} (pad the head)", head_dim);
}
