// Packed (NaViT / varlen) flash-style attention forward for gfx950, head_dim 64 or 128, GQA.
//
// Replaces flash_attn.flash_attn_varlen_func at its three call sites (qwen2_navit.py:361-370, 579-588;
// siglip_navit.py:232-241): per sample  softmax(q k^T * scale [+ bottom-right causal]) v, fp32 softmax,
// bf16 in/out.  Differences from the reference call that are deliberate MI355X design:
//   * K/V come from TWO segments per sample -- the immutable context cache and the freshly projected rows of
//     this forward -- so the per-layer "merge" copy of qwen2_navit.py:563-570 never happens;
//   * V is consumed TRANSPOSED ([kv_head][d][key], written once by bagel_v_transpose): the PV product contracts
//     over keys, so a [d][key] image lets both operands be fetched with conflict-free ds_read_b128.
//
// Kernel shape: one workgroup = 8 waves (two per SIMD) = 256 query rows of one (sample, head); each wave owns 32 rows.
//   S^T = K Q^T   with mfma_f32_32x32x16_bf16 (A = K tile from LDS, B = Q fragments held in registers), so every
//                 lane holds 32 of the 64 scores of ONE query row: the row max / row sum need a single
//                 cross-lane exchange (lane ^ 32).
//   O^T = V^T P^T with the SAME instruction: P (bf16, v_cvt_pk_bf16_f32) is used straight from the score registers
//                 as the B operand.  The MFMA row -> key assignment inside each 32-key block is permuted (quads
//                 1<->2 of every 16) when K fragments are read, which makes each lane's 8 k-slots 8 CONSECUTIVE
//                 keys, so the V^T fragment is one ds_read_b128.
//   K / V^T tiles (64 keys) stream HBM -> LDS by global_load_lds_dwordx4 (wave-uniform base + lane-constant offset) into a
//   3-slot ring, XOR-swizzled on the source side; ONE barrier per tile.
//   Every wave is SOFTWARE-PIPELINED over the KV tiles (round 2): while the matrix pipe runs S^T(t+1) = K(t+1) Q^T (phase A) the
//   VALU turns S^T(t) into P(t) (exp2, row sums, bf16 pack), and while it runs O^T += V^T(t) P^T(t) (phase B) the VALU takes the
//   row max of S^T(t+1).  Scores live in two named register sets that swap roles per tile (the tile loop is unrolled by two).
//   The issue order is written out by hand, one chunk per MFMA (see `step`).
//   Online softmax in base 2 with the scale folded into the exponent's fma; the running max is only raised (and O
//   rescaled) when some row's tile max exceeds it by more than 2^8 ("deferred max": P <= 256, exact in fp32/bf16
//   range; the final O/l normalisation makes the result independent of which max was used).  The decision for tile t+1 is
//   taken after P(t) V(t) has been issued -- O and l are rescaled behind those MFMAs, P(t+1) is exponentiated against the new max:
//   every term of O and l is scaled exactly once (cdna_hip_programming.md T13).
//   Block order is XCD-aware: the (sample, kv-head) pairs are dealt round-robin to the 8 XCDs (blockIdx % 8), and all
//   q-heads x q-tiles of one pair run back to back on that XCD, so its K/V (2.1 MB at 4098 keys) stays in the
//   XCD's 4 MiB L2 while ~119 workgroups stream it.
//
// Measured history of this kernel at the denoise shape (4 x 4098 rows, 28/4 heads, D 128; profiles/r02_attn_*.{log,txt}):
//   hipcc's own order, no pipelining 787 TFLOP/s -> LDS reads pipelined by sched_group_barrier 850-876 -> this form 867-904.
//   PMC of the previous form: matrix pipe busy 42 %, waves parked on waitcnt/barrier 27 %, issue-stalled 35 %, issuing 38 %.
//   Tried and removed: role alternation between the wave halves (four variants, 768-837); a one-wave-per-SIMD form with 64 query
//   rows per wave, O / Q / S in the accumulation registers and inline-asm MFMAs (correct, 155 VGPR + 256 AGPR, 787 TFLOP/s: a
//   single wave per SIMD issues every one of its ~590 instructions per tile itself -- 60 % of its cycles -- and nothing hides its
//   22 % of waitcnt/barrier time); timing-only ablations of this form: no barrier +5 %, no DMA +8 %, neither +11 %; same-box A/B
//   builds (tools/ab_attn.sh): fragment prefetch distance 3 / 5 / 6 instead of 4 and a static s_setprio 1 for waves 4-7: all within
//   1 % of 895 TFLOP/s (prefetch distance 5: -4 %).  A second one-wave-per-SIMD form (both score sets in VGPRs, separate 2-slot K / V^T
//   rings with compile-time LDS offsets, 438 instead of 590 instructions per tile, zero spills in the loop): correct, 877 TFLOP/s;
//   without barrier and DMA 987.  PMC (profiles/r02_attn_pmc_64row_v2.txt): the lone wave is issuing 55 % of its cycles -- ~5.8 cycles
//   per instruction (v_exp_f32 is quarter rate) = ~2 500 cycles per tile against 2 048 of matrix time -- so it cannot be matrix-bound
//   even with zero stalls, and it stalls another 45 %.  Removed as well; the two-waves-per-SIMD kernel below stays the only one.
//   Four waves per SIMD (two workgroups per CU) is not reachable from here: the wave's state is 64 O^T accumulators + two 32-register
//   score sets + 32 Q registers + fragment windows = 256 unified registers (the second __launch_bounds__ argument is waves per SIMD,
//   not workgroups per CU), so shrinking the LDS ring to 64 KB (separate 2-slot K and V^T rings: written, compiled, dropped) buys no
//   occupancy; 128 registers would need 16-row waves, i.e. twice the LDS fragment traffic per FLOP.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

struct AttnParams {
    const bf16_t* q; long ldq;
    const bf16_t* k_new; long ldk_new;
    const bf16_t* vt_new; long ldvt_new;
    const bf16_t* k_ctx; long ldk_ctx;
    const bf16_t* vt_ctx; long ldvt_ctx;
    bf16_t* out; long ldo;
    const int* cu_q;
    const int* cu_ctx;
    const int* q_end;        // optional: explicit end rows (ranges need not tile the buffers; prefixes may overlap)
    const int* ctx_end;
    const int* vt_new_col;
    const int* vt_ctx_col;
    int nq, nkv, causal;
    int batch, nqt;          // samples, 256-row query tiles per sample (from max_lq)
    int qsplit;              // interleaved query-tile sets per (sample, KV head) pair: 8 / gcd(pairs, 8), see the kernel
    float scale_log2;
    float* lse; long ld_lse; // optional: lse[h * ld_lse + row] = log2 of the row's softmax denominator in the scaled base-2 domain
                             // (P = exp2(scale_log2 * s - lse)), what the training backward needs from the forward (attention_bwd.hip)
};

// 16 bytes per lane HBM -> LDS (destination = wave-uniform LDS byte address + lane*16).  Issued from inline asm on
// purpose: hipcc cannot prove that the ring slot being filled is not the slot being read and would otherwise drain
// the DMA (s_waitcnt vmcnt(0)) in front of the first ds_read of every tile; completion is tracked by this file's own
// counted s_waitcnt vmcnt(N) + s_barrier (cdna_hip_programming.md 5.7).  M0 is saved/restored around the instruction.
__device__ __forceinline__ void glds16a(const void* gsrc, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst_uniform)
                 : "memory");
}
// the same with the address split into a wave-uniform 64-bit base (SGPR pair) and a 32-bit per-lane byte offset
__device__ __forceinline__ void glds16s(unsigned voff, const void* sbase_uniform, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase_uniform), "s"(lds_dst_uniform)
                 : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)p;
}

#define ATTN_DEFER_LOG2 8.0f

template <int D>
__global__ __launch_bounds__(512, 2) void attn_fwd_kernel(const AttnParams p) {
    constexpr int NQB = 1;                  // 32-row q-blocks per wave (the loops below are written for any number)
    constexpr int KS = D / 16;              // k-steps of the QK^T contraction
    constexpr int DB = D / 32;              // 32-row blocks of O^T
    constexpr int KROW = D * 2;             // bytes per K row in LDS
    constexpr int KT_BYTES = 64 * KROW;
    constexpr int VT_BYTES = D * 128;
    constexpr int STAGE = KT_BYTES + VT_BYTES;
    constexpr int NW = 8 / NQB;              // waves per workgroup; a wave owns NQB 32-row q-blocks
    constexpr int NLK = KT_BYTES / 1024 / NW;   // LDS-DMA pieces per wave for K   (4 @128)
    constexpr int NLV = VT_BYTES / 1024 / NW;   // ... for V^T
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- XCD-aware work mapping ----
    // All workgroups of one (sample, KV head) pair share its K / V^T through one XCD's L2, so pairs are dealt to the 8 XCDs (block id & 7).
    // When the pairs do not fill the XCDs evenly (batch-1 prefill: 4 pairs -> 4 idle XCDs; the 3-stream edit forward: 12 pairs -> 2/2/2/2/1/1/1/1)
    // every pair is split into p.qsplit interleaved sets of query tiles ("virtual pairs") so that their number is a multiple of 8.
    const int grp = p.nq / p.nkv;
    const int nqs = (p.nqt + p.qsplit - 1) / p.qsplit;       // query tiles per virtual pair
    const int nbpp = grp * nqs;
    const int nvp = p.batch * p.nkv * p.qsplit;
    const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
    const int vp = (kk / nbpp) * 8 + xcd;
    if (vp >= nvp) return;
    const int w = kk % nbpp;
    const int pair = vp / p.qsplit;
    const int b = pair / p.nkv, g = pair % p.nkv;
    const int qts = (w / grp) * p.qsplit + vp % p.qsplit;    // this workgroup's query tile, ascending
    if (qts >= p.nqt) return;
    // causal: the last query tile of a sample walks the most key tiles -- hand the heavy tiles out first (longest-processing-time order)
    const int h = g * grp + w % grp, qt = p.causal ? p.nqt - 1 - qts : qts;

    const int q0 = p.cu_q[b];
    const int Lq = (p.q_end ? p.q_end[b] : p.cu_q[b + 1]) - q0;
    if (qt * 256 >= Lq) return;
    const int c0 = p.cu_ctx ? p.cu_ctx[b] : 0;
    const int C = p.cu_ctx ? (p.ctx_end ? p.ctx_end[b] : p.cu_ctx[b + 1]) - c0 : 0;
    const int vcol_new = p.vt_new_col[b];
    const int vcol_ctx = C > 0 ? p.vt_ctx_col[b] : 0;

    const int qi = lane & 31, hi = lane >> 5;
    const int wrow0 = qt * 256 + wave * (32 * NQB);               // first row of this wave inside the sample
    const bool live = wrow0 < Lq;                          // wave-uniform
    int qrow[NQB];
    bf16x8_t qf[NQB][KS];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        qrow[qb] = wrow0 + 32 * qb + qi;                   // may exceed Lq-1: padding row
        const int rc = qrow[qb] < Lq ? qrow[qb] : Lq - 1;
        const bf16_t* qp = p.q + (long)(q0 + rc) * p.ldq + (long)h * D + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[qb][ks] = *(const bf16x8_t*)(qp + 16 * ks);
    }

    // ---- tile schedule: ctx tiles then new tiles ----
    const int nt_ctx = (C + 63) >> 6;
    const int qlast = min(Lq, qt * 256 + 256) - 1;
    const int new_needed = p.causal ? (qlast + 1) : Lq;
    const int nt_new = (new_needed + 63) >> 6;
    const int T = nt_ctx + nt_new;

    // ---- LDS-DMA of one 64-key tile: every address = wave-uniform base (SGPR pair) + a lane-constant 32-bit offset ----
    // piece i of this wave (global piece j = wave + NW*i): K rows RS*j + lane/(16|8), V^T rows 8*j + lane/8; the XOR swizzle of the
    // 16-byte chunk depends on the row only mod 16 (resp. (d>>1) mod 8), which the piece stride preserves -- so one lane offset per
    // operand and segment serves all pieces and all tiles, and the per-tile work is scalar arithmetic.
    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    constexpr int RS = D == 128 ? 4 : 8;          // K rows per LDS-DMA piece
    const int rowk = RS * wave + (D == 128 ? (lane >> 4) : (lane >> 3));
    const int gchk = D == 128 ? ((lane & 15) ^ (rowk & 15)) : ((lane & 7) ^ ((rowk >> 1) & 7));
    const int dv = 8 * wave + (lane >> 3);
    const int gchv = (lane & 7) ^ ((dv >> 1) & 7);
    const unsigned kcol = (unsigned)(g * D + gchk * 8) * 2u;
    const unsigned koff_ctx = (unsigned)rowk * (unsigned)p.ldk_ctx * 2u + kcol, koff_new = (unsigned)rowk * (unsigned)p.ldk_new * 2u + kcol;
    const unsigned voff_ctx = ((unsigned)(g * D + dv) * (unsigned)p.ldvt_ctx + gchv * 8) * 2u;
    const unsigned voff_new = ((unsigned)(g * D + dv) * (unsigned)p.ldvt_new + gchv * 8) * 2u;
    auto issue = [&](int stage, int t) {
        const unsigned sb = smem_base + __builtin_amdgcn_readfirstlane(stage) * STAGE + wave * 1024u;
        const bool is_ctx = t < nt_ctx;
        const int ti = is_ctx ? t : t - nt_ctx;
        const int seglen = is_ctx ? C : Lq;
        const long ldk = is_ctx ? p.ldk_ctx : p.ldk_new;
        const long ldvt = is_ctx ? p.ldvt_ctx : p.ldvt_new;
        const char* kseg = (const char*)(is_ctx ? p.k_ctx + (long)c0 * p.ldk_ctx : p.k_new + (long)q0 * p.ldk_new);
        const char* vseg = (const char*)(is_ctx ? p.vt_ctx : p.vt_new) + (long)((is_ctx ? vcol_ctx : vcol_new) + ti * 64) * 2;
        const unsigned vo = is_ctx ? voff_ctx : voff_new;
        if (ti * 64 + 64 <= seglen) {
            const char* kt = kseg + (long)ti * 64 * ldk * 2;
            const unsigned ko = is_ctx ? koff_ctx : koff_new;
#pragma unroll
            for (int i = 0; i < NLK; ++i) glds16s(ko, kt + (long)i * (RS * NW) * ldk * 2, sb + i * (NW * 1024u));
        } else {
            // the segment's last tile: rows past its end re-read the last key (masked in the scores); per-lane offsets
#pragma unroll
            for (int i = 0; i < NLK; ++i) {
                int key = ti * 64 + rowk + RS * NW * i;
                key = key < seglen ? key : seglen - 1;
                glds16s((unsigned)key * (unsigned)ldk * 2u + kcol, kseg, sb + i * (NW * 1024u));
            }
        }
#pragma unroll
        for (int i = 0; i < NLV; ++i) glds16s(vo, vseg + (long)i * (8 * NW) * ldvt * 2, sb + KT_BYTES + i * (NW * 1024u));
    };

    // ---- per-lane constants for fragment reads ----
    // MFMA row i = lane&31 reads key pi(i): swap quads 1<->2 inside each 16
    const int quad = (qi >> 2) & 3;
    const int pkey = (qi & 16) | ((((quad & 1) << 1) | (quad >> 1)) << 2) | (qi & 3);
    int kswz;
    if (D == 128) { kswz = pkey & 15; }
    else          { kswz = (pkey >> 1) & 7; }
    const int koff0 = pkey * KROW;
    const int vswz = (qi >> 1) & 7;
    const int voff = KT_BYTES + qi * 128;
    int kch[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kch[ks] = koff0 + (((2 * ks + hi) ^ kswz) << 4);
    int vch[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) vch[j] = voff + (((2 * j + hi) ^ vswz) << 4);

    f32x16_t o[NQB][DB];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
        for (int i = 0; i < DB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qb][i][r] = 0.f;
    float m_run[NQB], l_run[NQB], m_use[NQB];                          // m_run in log2 units
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) { m_run[qb] = -INFINITY; l_run[qb] = 0.f; m_use[qb] = 0.f; }

    if (T > 0) issue(0, 0);
    if (T > 1) issue(1, 1);
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" ::"v"(qf[qb][ks]));

    constexpr int NK = 2 * KS;                  // K fragments of a tile  (index KS*kb + ks)
    constexpr int NV = DB * 4;                  // V^T fragments of a tile (index 4*db + j)
    auto slot = [&](int t) { return (const char*)smem + (t % 3) * STAGE; };
    auto kfrag = [&](const char* sb, int idx) { return *(const bf16x8_t*)(sb + (idx / KS) * 32 * KROW + kch[idx % KS]); };
    auto vfrag = [&](const char* sb, int idx) { return *(const bf16x8_t*)(sb + (idx >> 2) * 4096 + vch[idx & 3]); };

    // mask (tile tails, causal) of the scores of tile t, then the row max and the deferred-max decision -> m_run / m_use (and the
    // rescale of O, l -- the caller guarantees every MFMA into O has been issued before)
    auto mask_tile = [&](f32x16_t (&s)[NQB][2], int t) {
        const bool is_ctx = t < nt_ctx;
        const int ti = is_ctx ? t : t - nt_ctx;
        const int seglen = is_ctx ? C : Lq;
        const int kbase = ti * 64;
        const bool need_mask = (kbase + 64 > seglen) || (p.causal && !is_ctx && (kbase + 63 > wrow0));
        if (need_mask) {
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kbase + 32 * kb + 16 * (r >> 3) + 8 * hi + (r & 7);
                        const bool ok = key < seglen && (!p.causal || is_ctx || key <= qrow[qb]);
                        s[qb][kb][r] = ok ? s[qb][kb][r] : -INFINITY;
                    }
        }
    };
    auto row_max = [&](f32x16_t (&s)[NQB][2], float (&mx)[NQB]) {
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            float a = fmaxf(s[qb][0][0], s[qb][1][0]), c = fmaxf(s[qb][0][1], s[qb][1][1]);
#pragma unroll
            for (int r = 2; r < 16; r += 2) {
                a = fmaxf(fmaxf(a, s[qb][0][r]), s[qb][1][r]);
                c = fmaxf(fmaxf(c, s[qb][0][r + 1]), s[qb][1][r + 1]);
            }
            a = fmaxf(a, c);
            mx[qb] = fmaxf(a, __shfl_xor(a, 32, 64)) * p.scale_log2;
        }
    };
    auto raise_max = [&](const float (&mx)[NQB]) {
        bool grow = false;
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) grow = grow || (mx[qb] > m_run[qb] + ATTN_DEFER_LOG2);
        if (__any(grow)) {
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) {
                const float m_new = fmaxf(m_run[qb], mx[qb]);
                const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - (m_new == -INFINITY ? 0.f : m_new));
                m_run[qb] = m_new;
                l_run[qb] *= alpha;
#pragma unroll
                for (int i = 0; i < DB; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[qb][i][r] *= alpha;
                m_use[qb] = m_new == -INFINITY ? 0.f : m_new;
            }
        }
    };

    // One pipeline step: phase A = [S^T(t+1) -> sn] || [exp2 / pack of sc -> pf] ; phase B = [O^T += V^T(t) P^T(t)] || [row sum of
    // P(t), row max of sn].  LAST: there is no tile t+1.
    auto step = [&](f32x16_t (&sc)[NQB][2], f32x16_t (&sn)[NQB][2], int t, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        if (!LAST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of tile t+1 have landed
        asm volatile("s_barrier" ::: "memory");          // ... everyone's; and everyone is done with tile t-1
        if (t + 2 < T) issue((t + 2) % 3, t + 2);
        if (!live) return;
        const char* sbv = slot(t);
        const char* sbk = slot(t + 1);
        // Issue order is written out by hand: one CHUNK per MFMA = the MFMA, the fragment read WIN chunks ahead, and a 1/16 slice of the
        // softmax work; a sched_barrier(0) closes every chunk, so hipcc orders instructions inside a chunk only.  Consecutive MFMAs hit
        // different accumulators (a VALU slot between two MFMAs of one accumulation chain costs ~43 cycles: MI355X_MICROARCH.md).
        constexpr int WIN = 4;        // 3 .. 6 measure the same (894-899 TFLOP/s, same-box A/B): LDS latency is not what this kernel waits for
        constexpr int NMA = NQB * NK, NMB = NQB * NV;          // MFMAs of phase A / phase B
        constexpr int SPA = 16 * NQB / NMA, SPB = 16 * NQB / NMB;   // softmax / row-max slices per chunk (1 at D = 128, 2 at D = 64)
        unsigned pw[NQB][16];                                  // P(t) as packed bf16 pairs: pw[qb][4 * f + e] = word e of fragment f
        float acc[NQB][4];
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[qb][i] = 0.f;
        // slice m of the softmax of tile t: scores (2 pi, 2 pi + 1) of q-block qbv -> exp2, partial row sums, one packed word
        auto softmax_slice = [&](int m) {
            const int qbv = m / 16, pi = m % 16;
            const int kb = pi / 8, r = (2 * pi) % 16;
            const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[qbv][kb][r], p.scale_log2, -m_use[qbv]));
            const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[qbv][kb][r + 1], p.scale_log2, -m_use[qbv]));
            acc[qbv][2 * (pi & 1)] += p0;
            acc[qbv][2 * (pi & 1) + 1] += p1;
            pw[qbv][pi] = pack2bf(p0, p1);
            // opaque re-definitions: these values have no consumer before phase B, and LLVM would otherwise sink the whole slice below
            // the mask branch, out of the chunk it is meant to fill
            asm volatile("" : "+v"(pw[qbv][pi]), "+v"(acc[qbv][2 * (pi & 1)]), "+v"(acc[qbv][2 * (pi & 1) + 1]));
        };
        auto kidx = [&](int j) { return (j & 1) * KS + (j >> 1); };          // K fragments alternate between the two key blocks
        auto vidx = [&](int f) { return 4 * (f % DB) + f / DB; };            // V^T fragments round-robin over the O^T blocks
        bf16x8_t vf[NV];
        // ---------------- phase A:  S^T(t+1) = K(t+1) Q^T  ||  P(t) = exp2(S^T(t) - m) ----------------
        __builtin_amdgcn_sched_barrier(0);
        if (!LAST) {
            bf16x8_t kf[NK];
#pragma unroll
            for (int j = 0; j < WIN; ++j) kf[j] = kfrag(sbk, kidx(j));
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sn[qb][kb][r] = 0.f;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < NMA; ++m) {
                const int j = m / NQB, qb = m % NQB, kb = j & 1, ks = j >> 1;
                sn[qb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[j], qf[qb][ks], sn[qb][kb], 0, 0, 0);
                if (qb == NQB - 1) {
                    if (j + WIN < NK) kf[j + WIN] = kfrag(sbk, kidx(j + WIN));
                    else if (j + WIN - NK < WIN) vf[j + WIN - NK] = vfrag(sbv, vidx(j + WIN - NK));     // the first V^T fragments of phase B
                }
#pragma unroll
                for (int u = 0; u < SPA; ++u) softmax_slice(m * SPA + u);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < WIN; ++j) vf[j] = vfrag(sbv, vidx(j));
#pragma unroll
            for (int m = 0; m < 16 * NQB; ++m) softmax_slice(m);
        }
        bf16x8_t pf[NQB][4];
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            l_run[qb] += (acc[qb][0] + acc[qb][1]) + (acc[qb][2] + acc[qb][3]);
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                u32x4_t v4 = {pw[qb][4 * f], pw[qb][4 * f + 1], pw[qb][4 * f + 2], pw[qb][4 * f + 3]};
                pf[qb][f] = __builtin_bit_cast(bf16x8_t, v4);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!LAST) mask_tile(sn, t + 1);
        // ---------------- phase B:  O^T += V^T(t) P^T(t)  ||  row max of S^T(t+1) ----------------
        __builtin_amdgcn_sched_barrier(0);
        float mx[NQB];
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) mx[qb] = -INFINITY;
#pragma unroll
        for (int m = 0; m < NMB; ++m) {
            const int f = m / NQB, qb = m % NQB, db = f % DB, jj = f / DB;
            o[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[f], pf[qb][jj], o[qb][db], 0, 0, 0);
            if (qb == NQB - 1 && f + WIN < NV) vf[f + WIN] = vfrag(sbv, vidx(f + WIN));
            if (!LAST) {
#pragma unroll
                for (int u = 0; u < SPB; ++u) {
                    const int c = m * SPB + u, qbv = c / 16, i = c % 16, kb = i / 8, r = (2 * i) % 16;
                    mx[qbv] = fmaxf(fmaxf(mx[qbv], sn[qbv][kb][r]), sn[qbv][kb][r + 1]);
                    asm volatile("" : "+v"(mx[qbv]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!LAST) {
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) mx[qb] = fmaxf(mx[qb], __shfl_xor(mx[qb], 32, 64)) * p.scale_log2;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!LAST) raise_max(mx);
    };

    // f32x16_t x[2][2] twice: the two score sets
    f32x16_t sA[NQB][2], sB[NQB][2];
    if (T > 0) {
        // prologue: tile 0 resident; S^T(0) -> sA with nothing to overlap, its mask / max / first m
        if (T > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLK + NLV) : "memory");
        else       asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (live) {
            const char* sbk = slot(0);
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sA[qb][kb][r] = 0.f;
#pragma unroll
            for (int j = 0; j < NK; ++j) {
                const bf16x8_t kf = kfrag(sbk, j);
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb)
                    sA[qb][j / KS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qb][j % KS], sA[qb][j / KS], 0, 0, 0);
            }
            mask_tile(sA, 0);
            float mx[NQB];
            row_max(sA, mx);
            raise_max(mx);
        }
        int t = 0;
        for (; t + 2 <= T - 1; t += 2) {
            step(sA, sB, t, std::false_type{});
            step(sB, sA, t + 1, std::false_type{});
        }
        if (t < T - 1) {
            step(sA, sB, t, std::false_type{});
            step(sB, sA, t + 1, std::true_type{});
        } else {
            step(sA, sB, t, std::true_type{});
        }
    }

    // ---- epilogue: O[q][d] = O^T / l ; lane owns d = 32*db + 8*u + 4*hi + (0..3) ----
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
        const float inv = 1.0f / l_tot;
        if (p.lse && hi == 0 && qrow[qb] < Lq) p.lse[(long)h * p.ld_lse + q0 + qrow[qb]] = m_use[qb] + log2f(l_tot);
        if (qrow[qb] < Lq) {
            bf16_t* op = p.out + (long)(q0 + qrow[qb]) * p.ldo + (long)h * D + 4 * hi;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    u32x2_t v = {pack2bf(o[qb][db][4 * u] * inv, o[qb][db][4 * u + 1] * inv),
                                 pack2bf(o[qb][db][4 * u + 2] * inv, o[qb][db][4 * u + 3] * inv)};
                    *(u32x2_t*)(op + 32 * db + 8 * u) = v;
                }
        }
    }
}

static int attn_launch(const void* q, int64_t ldq, const void* k_new, int64_t ldk_new, const void* vt_new,
                       int64_t ldvt_new, const void* k_ctx, int64_t ldk_ctx, const void* vt_ctx,
                       int64_t ldvt_ctx, void* out, int64_t ldo, const int32_t* cu_q, const int32_t* q_end, const int32_t* cu_ctx,
                       const int32_t* ctx_end, const int32_t* vt_new_col, const int32_t* vt_ctx_col, int32_t batch, int32_t max_lq,
                       int32_t nq, int32_t nkv, int32_t head_dim, int32_t causal, float softmax_scale, hipStream_t stream,
                       float* lse = nullptr, int64_t ld_lse = 0);

extern "C" int bagel_attn_varlen_bf16(const void* q, int64_t ldq, const void* k_new, int64_t ldk_new, const void* vt_new,
                                      int64_t ldvt_new, const void* k_ctx, int64_t ldk_ctx, const void* vt_ctx,
                                      int64_t ldvt_ctx, void* out, int64_t ldo, const int32_t* cu_q, const int32_t* cu_ctx,
                                      const int32_t* vt_new_col, const int32_t* vt_ctx_col, int32_t batch, int32_t max_lq,
                                      int32_t nq, int32_t nkv, int32_t head_dim, int32_t causal, float softmax_scale,
                                      hipStream_t stream) {
    return attn_launch(q, ldq, k_new, ldk_new, vt_new, ldvt_new, k_ctx, ldk_ctx, vt_ctx, ldvt_ctx, out, ldo, cu_q, nullptr, cu_ctx,
                       nullptr, vt_new_col, vt_ctx_col, batch, max_lq, nq, nkv, head_dim, causal, softmax_scale, stream);
}

// Same kernel, explicit [start, end) row ranges per sequence: query/new-key rows q_start[b]..q_end[b] need not tile the
// buffers and the context ranges ctx_start[b]..ctx_end[b] may overlap (prefixes of one stream) -- what the block mask of
// the training forward decomposes into (data/data_utils.py:72-103; bagel.py:155-166).
extern "C" int bagel_attn_varlen_ranges_bf16(const void* q, int64_t ldq, const void* k_new, int64_t ldk_new, const void* vt_new,
                                             int64_t ldvt_new, const void* k_ctx, int64_t ldk_ctx, const void* vt_ctx,
                                             int64_t ldvt_ctx, void* out, int64_t ldo, const int32_t* q_start,
                                             const int32_t* q_end, const int32_t* ctx_start, const int32_t* ctx_end,
                                             const int32_t* vt_new_col, const int32_t* vt_ctx_col, int32_t batch,
                                             int32_t max_lq, int32_t nq, int32_t nkv, int32_t head_dim, int32_t causal,
                                             float softmax_scale, hipStream_t stream) {
    BAGEL_REQUIRE(q_end && (!ctx_start || ctx_end), "attn_ranges: end arrays missing");
    return attn_launch(q, ldq, k_new, ldk_new, vt_new, ldvt_new, k_ctx, ldk_ctx, vt_ctx, ldvt_ctx, out, ldo, q_start, q_end, ctx_start,
                       ctx_end, vt_new_col, vt_ctx_col, batch, max_lq, nq, nkv, head_dim, causal, softmax_scale, stream);
}

// bagel_attn_varlen_ranges_bf16 that also leaves the row statistics the training backward needs: lse[h * ld_lse + row] = log2 of the
// softmax denominator of (row, q head h) in the scaled base-2 domain, fp32 (every query row must belong to exactly one range).
extern "C" int bagel_attn_varlen_ranges_lse_bf16(const void* q, int64_t ldq, const void* k_new, int64_t ldk_new, const void* vt_new,
                                                 int64_t ldvt_new, const void* k_ctx, int64_t ldk_ctx, const void* vt_ctx,
                                                 int64_t ldvt_ctx, void* out, int64_t ldo, const int32_t* q_start,
                                                 const int32_t* q_end, const int32_t* ctx_start, const int32_t* ctx_end,
                                                 const int32_t* vt_new_col, const int32_t* vt_ctx_col, int32_t batch,
                                                 int32_t max_lq, int32_t nq, int32_t nkv, int32_t head_dim, int32_t causal,
                                                 float softmax_scale, float* lse, int64_t ld_lse, hipStream_t stream) {
    BAGEL_REQUIRE(q_end && (!ctx_start || ctx_end), "attn_ranges: end arrays missing");
    BAGEL_REQUIRE(lse && ld_lse > 0, "attn_ranges_lse: lse buffer missing");
    return attn_launch(q, ldq, k_new, ldk_new, vt_new, ldvt_new, k_ctx, ldk_ctx, vt_ctx, ldvt_ctx, out, ldo, q_start, q_end, ctx_start,
                       ctx_end, vt_new_col, vt_ctx_col, batch, max_lq, nq, nkv, head_dim, causal, softmax_scale, stream, lse, ld_lse);
}

static int attn_launch(const void* q, int64_t ldq, const void* k_new, int64_t ldk_new, const void* vt_new,
                       int64_t ldvt_new, const void* k_ctx, int64_t ldk_ctx, const void* vt_ctx,
                       int64_t ldvt_ctx, void* out, int64_t ldo, const int32_t* cu_q, const int32_t* q_end, const int32_t* cu_ctx,
                       const int32_t* ctx_end, const int32_t* vt_new_col, const int32_t* vt_ctx_col, int32_t batch, int32_t max_lq,
                       int32_t nq, int32_t nkv, int32_t head_dim, int32_t causal, float softmax_scale, hipStream_t stream,
                       float* lse, int64_t ld_lse) {
    BAGEL_REQUIRE(q && k_new && vt_new && out && cu_q && vt_new_col, "attn: null pointer");
    BAGEL_REQUIRE(!cu_ctx || (k_ctx && vt_ctx && vt_ctx_col), "attn: context segment incomplete");
    BAGEL_REQUIRE(nq > 0 && nkv > 0 && nq % nkv == 0, "attn: bad head counts %d/%d", nq, nkv);
    BAGEL_REQUIRE(ldq % 8 == 0 && ldk_new % 8 == 0 && ldvt_new % 8 == 0 && ldo % 4 == 0 && ldk_ctx % 8 == 0 && ldvt_ctx % 8 == 0,
                  "attn: leading dims must keep 16-byte alignment");
    BAGEL_REQUIRE(softmax_scale > 0.f, "attn: softmax_scale must be positive");
    if (batch <= 0 || max_lq <= 0) return BAGEL_OK;
    AttnParams p;
    p.q = (const bf16_t*)q; p.ldq = ldq;
    p.k_new = (const bf16_t*)k_new; p.ldk_new = ldk_new;
    p.vt_new = (const bf16_t*)vt_new; p.ldvt_new = ldvt_new;
    p.k_ctx = (const bf16_t*)k_ctx; p.ldk_ctx = ldk_ctx;
    p.vt_ctx = (const bf16_t*)vt_ctx; p.ldvt_ctx = ldvt_ctx;
    p.out = (bf16_t*)out; p.ldo = ldo;
    p.cu_q = cu_q; p.cu_ctx = cu_ctx; p.q_end = q_end; p.ctx_end = ctx_end; p.vt_new_col = vt_new_col; p.vt_ctx_col = vt_ctx_col;
    p.nq = nq; p.nkv = nkv; p.causal = causal;
    p.batch = batch; p.nqt = ceil_div(max_lq, 256);
    p.scale_log2 = softmax_scale * 1.4426950408889634f;
    p.lse = lse; p.ld_lse = ld_lse;
    const int npairs = batch * nkv;
    int gcd8 = 8;
    while (npairs % gcd8) gcd8 >>= 1;
    p.qsplit = 8 / gcd8;
    while (p.qsplit > 1 && p.qsplit > p.nqt) p.qsplit >>= 1;
    const int nbpp = (nq / nkv) * ceil_div(p.nqt, p.qsplit);
    const int vp_per_xcd = ceil_div((long)npairs * p.qsplit, 8);
    const dim3 grid(8 * vp_per_xcd * nbpp), block(512);
    if (head_dim == 128) {
        constexpr int smem = 3 * (64 * 256 + 128 * 128);
        if (int rc = bagel_enable_lds((const void*)attn_fwd_kernel<128>, smem, "attn_fwd_kernel<128>")) return rc;
        hipLaunchKernelGGL((attn_fwd_kernel<128>), grid, block, smem, stream, p);
    } else if (head_dim == 64) {
        constexpr int smem = 3 * (64 * 128 + 64 * 128);
        hipLaunchKernelGGL((attn_fwd_kernel<64>), grid, block, smem, stream, p);
    } else {
        return bagel_set_error(BAGEL_ERR_UNSUPPORTED, "attn: head_dim %d not in {64,128} (pad the head)", head_dim);
    }
    return bagel_check_launch("attn_fwd_kernel");
}

// ---------------------------------------------------------------------------------------------------------
// V -> V^T :  src rows [cu[b], cu[b+1]) x [nkv][D]  ->  dst[(g*D + d) * ld_dst + col0[b] + (row - cu[b])]
// One block = 64 rows x one kv head; 16-byte loads, LDS transpose, 16-byte stores (8 consecutive keys of one d).
// Columns past the sample's end inside the last 64-block are zero-filled so the PV MFMA never sees stale NaNs.
// ---------------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void v_transpose_kernel(const bf16_t* __restrict__ src, long ld_src, bf16_t* __restrict__ dst,
                                                          long ld_dst, const int* __restrict__ cu, const int* __restrict__ col0,
                                                          int nkv) {
    __shared__ bf16_t tile[64][D + 2];
    const int b = blockIdx.z, g = blockIdx.y, rt = blockIdx.x;
    const int r0 = cu[b], L = cu[b + 1] - r0;
    if (rt * 64 >= L) return;
    const int tid = threadIdx.x;
    constexpr int CPR = D / 8;   // 16-byte chunks per row
    for (int i = tid; i < 64 * CPR; i += 256) {
        const int r = i / CPR, c = i % CPR;
        const int row = rt * 64 + r;
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (row < L) v = *(const u32x4_t*)(src + (long)(r0 + row) * ld_src + (long)g * D + c * 8);
        unsigned* t = (unsigned*)&tile[r][c * 8];   // (D+2)*2 bytes per row keeps 4-byte alignment
        t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; t[3] = v[3];
    }
    __syncthreads();
    for (int i = tid; i < D * 8; i += 256) {
        const int d = i >> 3, kc = i & 7;
        bf16_t e[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) e[k] = tile[kc * 8 + k][d];
        u32x4_t v = {(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16),
                     (unsigned)e[4] | ((unsigned)e[5] << 16), (unsigned)e[6] | ((unsigned)e[7] << 16)};
        *(u32x4_t*)(dst + ((long)g * D + d) * ld_dst + col0[b] + rt * 64 + kc * 8) = v;
    }
}

extern "C" int bagel_v_transpose_bf16(const void* v, int64_t ld_src, void* vt, int64_t ld_dst, const int32_t* cu_rows,
                                      const int32_t* col_start, int32_t batch, int32_t max_len, int32_t nkv, int32_t head_dim,
                                      hipStream_t stream) {
    BAGEL_REQUIRE(v && vt && cu_rows && col_start, "v_transpose: null pointer");
    BAGEL_REQUIRE(ld_src % 8 == 0 && ld_dst % 8 == 0, "v_transpose: leading dims must be multiples of 8");
    if (batch <= 0 || max_len <= 0) return BAGEL_OK;
    const dim3 grid(ceil_div(max_len, 64), nkv, batch), block(256);
    if (head_dim == 128)
        hipLaunchKernelGGL(v_transpose_kernel<128>, grid, block, 0, stream, (const bf16_t*)v, (long)ld_src, (bf16_t*)vt, (long)ld_dst, cu_rows, col_start, nkv);
    else if (head_dim == 64)
        hipLaunchKernelGGL(v_transpose_kernel<64>, grid, block, 0, stream, (const bf16_t*)v, (long)ld_src, (bf16_t*)vt, (long)ld_dst, cu_rows, col_start, nkv);
    else
        return bagel_set_error(BAGEL_ERR_UNSUPPORTED, "v_transpose: head_dim %d not in {64,128}", head_dim);
    return bagel_check_launch("v_transpose_kernel");
}
