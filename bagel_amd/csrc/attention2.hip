// Planned, PERSISTENT form of the packed flash-style attention forward of attention.hip (same arithmetic, same fragment layouts, same
// software pipeline per wave -- read that file's header first).  Replaces flash_attn.flash_attn_varlen_func at the engine's call sites
// (qwen2_navit.py:579-588; siglip_navit.py:232-241) when the caller holds the sequence lengths on the host, which the product's
// ForwardPlan always does.
//
// What changes against attention.hip (round-2 verdict, "Weak #2": non-persistent grid, the 17th query tile, nothing prefetched across
// block seams):
//   * WORK LIST.  The launch no longer derives (sample, head, query tile) from blockIdx arithmetic.  A host planner
//     (bagel_attn_plan, below: plain C++, no device code) cuts the problem into ITEMS -- one 256-row query tile of one query head over
//     a range of 64-key tiles -- described by 16 ints each, and hands every one of the chip's 256 workgroups ("workers") its own
//     contiguous run of items.  The kernel reads its items with scalar loads.
//   * TAIL ROWS.  A last query tile of <= 32 rows (the <|vision_end|> tail of a 4098-row denoise sample: 4098 = 16 x 256 + 2) used
//     to cost G = 7 full workgroups per KV head, each streaming every key for ONE live wave.  It is now ONE item in
//     "head-per-wave" form: wave w serves query head g*G + w on the same <= 32 rows, so the K / V^T tiles are streamed once for the
//     whole GQA group.  The same form serves short prompts (a 34-token text prefill is 4 items instead of 28).
//   * KEY SPLITS.  Items are dealt to the 32 workers of an XCD in rounds; the items of a last, partial round are split along the key
//     axis into as many sub-items as there are idle workers.  A sub-item writes un-normalised fp32 partials (O, running max, running
//     sum); attn2_combine_kernel merges them.  (3584 + 32 items of a stream-batched denoise forward = 14 full rounds + a round of
//     eighth-length sub-items instead of 15 rounds; the 1344 items of the 3-stream edit forward = 5.25 rounds instead of 6.)
//   * PERSISTENT workgroups with a CONTINUOUS K / V^T tile stream: the LDS-DMA cursor runs two tiles ahead of the compute cursor
//     ACROSS item seams, so the first two tiles of item i+1 are in flight while item i computes its last two tiles; the 3-slot ring,
//     the one barrier per tile and the counted waits are unchanged.
//   * XCD placement as before: every (sample, KV head) pair (or interleaved part of one) is worked on by the 32 workers of ONE XCD at
//     the same time, so its K / V^T is fetched from HBM once per XCD pass.
// Results: an item that is not key-split performs exactly attention.hip's operations in the same order -- bit-identical output
// (tests/test_attn2_gpu.py); key-split items differ by the fp32 re-association of the combine only.
#include "common.h"
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include <vector>

struct AttnItem {        // 16 ints = 64 bytes, written by the host planner
    int q_row0;          // absolute row (q / out buffers) of the item's first query row
    int nrows;           // live query rows: <= 256 (head-per-wave form: <= 32)
    int h;               // query head (head-per-wave: first head of the KV group)
    int g;               // KV head
    int flags;           // bit 0 head-per-wave, bit 1 causal, bit 2 partial output; bits 8..15 = G (waves that are live in head-per-wave form)
    int q_rel0;          // index of the first query row inside its sample's new segment (causal mask)
    int kn_row0;         // absolute row of new-segment key 0 in k_new
    int l_new;           // keys in the sample's new segment
    int kc_row0;         // absolute row of context key 0 in k_ctx
    int l_ctx;           // context keys
    int vtn_col0;        // V^T column of new-segment key 0
    int vtc_col0;        // V^T column of context key 0
    int t0, t1;          // 64-key tile range [t0, t1) over the sample's tile list [context tiles | new tiles]
    int part;            // partial slot (flags bit 2)
    int pad;
};
#define ATTN2_HPW 1
#define ATTN2_CAUSAL 2
#define ATTN2_PARTIAL 4

struct AttnComb {        // 8 ints: one key-split item = `nslots` consecutive partial slots -> one output tile
    int q_row0, nrows, h, flags, slot0, nslots, pad0, pad1;
};

struct Attn2Params {
    const bf16_t* q; long ldq;
    const bf16_t* k_new; long ldk_new;
    const bf16_t* vt_new; long ldvt_new;
    const bf16_t* k_ctx; long ldk_ctx;
    const bf16_t* vt_ctx; long ldvt_ctx;
    bf16_t* out; long ldo;
    float* part;                 // [slots][256 * (D + 2)] fp32
    float scale_log2;
};

__device__ __forceinline__ const char* a2_uniform(const char* ptr) {          // makes wave-uniformity provable to hipcc (an "s" operand)
    const unsigned long v = (unsigned long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(((unsigned long)hi << 32) | lo);
}
__device__ __forceinline__ void a2_glds16s(unsigned voff, const void* sbase_uniform, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase_uniform), "s"(lds_dst_uniform)
                 : "memory");
}

// Measured and removed in round 6 (profiles/r06_attn_split_ring.log, "DMA path A/B"): an incremental-pointer fast path for the regular tiles of the LDS-DMA stream
// (all pieces of a tile in one asm statement off ONE uniform base per operand, m0 saved once per tile, the cursor arithmetic reduced to two 64-bit adds): 1 018-1 024
// TFLOP/s against 1 035-1 037 for the general path below on the same box -- the second code path cost 114 spilled SGPRs -- while switching the DMA OFF altogether
// (A2_ABL = 8) bounds everything on that side at +5.8 %.  The softmax VALU (A2_ABL = 2: +26 %) is what this loop waits for, not its scalar address work.
#define ATTN2_DEFER_LOG2 8.0f

// Measured and removed in round 3 (profiles/r03_attn_planned_vs_tile.log; the machinery was a table of softmax slices / LDS-DMA pieces per
// chunk): moving 4-6 of the 16 softmax slices of a tile from phase A (7 VALU per MFMA) into phase B (1 VALU per MFMA), issuing the
// LDS-DMA pieces of tile t+2 between the first MFMAs of phase A instead of in front of them, and slice PAIRS with their four
// fma / exp2 / add chains interleaved: all within 1 % of the plain one-slice-per-MFMA schedule (1001 / 988 / 994 / 991 TFLOP/s on one
// box, 1045 / 1046 on another) -- neither the issue balance of the two phases, nor the DMA issue slots, nor VALU dependency stalls
// are what this loop waits for.  A fragment prefetch distance of 3 chunks instead of 4 measures the same and frees four registers.
#ifndef A2_WIN
#define A2_WIN 3
#endif
#ifndef A2_ABL
#define A2_ABL 0             // timing-only ablations (WRONG results; tools/ab_build.sh): bit 0 no barrier / vmcnt waits, bit 1 no softmax / row-max VALU,
#endif                       // bit 2 no LDS fragment reads inside the phases, bit 3 no LDS-DMA after the priming
#ifndef A2_WIDE_STORE
#define A2_WIDE_STORE 1      // epilogue: half-wave exchange -> 16-byte stores (0 = the 8-byte form, same bytes; tools/ab_build.sh A/B)
#endif
#ifndef A2_SLOTS
#define A2_SLOTS 4          // LDS ring slots (3 = fetch at the top of a step, two tiles ahead; 4 = fetch inside phase B, three tiles ahead)
#endif

// value of lane ^ 32 combined with the own value, without the LDS round trip of __shfl_xor (ds_bpermute + 6 address VALU + ~100 cycles
// on the serial tail of every tile): v_permlane32_swap exchanges the upper half of one register with the lower half of the other, so
// after swapping two copies of x, (r0, r1) = (x[l & 31], x[32 | (l & 31)]) in every lane.
__device__ __forceinline__ float a2_max_halves(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float a2_sum_halves(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// SPLIT (round 6, the round-5 verdict's "second workgroup per CU"): the K and the V^T tiles get a 2-slot ring EACH (2 x 16 KB + 2 x 16 KB = 64 KB instead
// of the unified 4-slot ring's 128 KB), so TWO workgroups are resident per CU -- four waves per SIMD: while one workgroup's waves sit at the tile
// barrier or in the softmax VALU, the other's MFMAs have the matrix pipe.  The tile stream is the same (tiles of all the worker's items in order, index x):
// K(x) is read by the step (or item prologue) that forms S(x), V^T(x) by step x, so after the barrier at the top of step x both K slot x % 2 (held K(x))
// and V^T slot (x + 1) % 2 (held V^T(x - 1)) are free: the fetch unit F(x) = {K(x + 2), V^T(x + 1)} is issued right there and has landed -- vmcnt(0), it is
// the newest thing in flight -- before the barrier of step x + 1: ONE step of flight (the unified ring gives three; the second workgroup covers the rest).
// Two cursors walk the item list, the K one a tile ahead of the V^T one.  Same arithmetic in the same order: bit-identical to the unified-ring kernel.
template <int D, bool SPLIT>
__global__ __launch_bounds__(512, 2) void attn2_kernel(const Attn2Params p, const int* __restrict__ worker_off,
                                                        const AttnItem* __restrict__ item_tab) {
    // (the two tables are separate noalias kernel arguments on purpose: only then does hipcc read them with SCALAR loads -- as members of
    // `p` they may alias the output stores, every descriptor field becomes a vector load + v_readfirstlane, and hipcc's vmcnt waits for
    // those loads drain the asm LDS-DMA in the middle of the tile loop)
    constexpr int KS = D / 16;              // k-steps of the QK^T contraction
    constexpr int DB = D / 32;              // 32-row blocks of O^T
    constexpr int KROW = D * 2;             // bytes per K row in LDS
    constexpr int KT_BYTES = 64 * KROW;
    constexpr int VT_BYTES = D * 128;
    constexpr int STAGE = KT_BYTES + VT_BYTES;
    constexpr int NS = A2_SLOTS;            // ring slots; the LDS-DMA cursor runs NS - 1 tiles ahead of the compute cursor
    constexpr int NW = 8;
    constexpr int NLK = KT_BYTES / 1024 / NW;   // LDS-DMA pieces per wave for K   (2 @128)
    constexpr int NLV = VT_BYTES / 1024 / NW;   // ... for V^T
    constexpr int PSLOT = 256 * (D + 2);        // floats per partial slot
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i_begin = worker_off[blockIdx.x], i_end = worker_off[blockIdx.x + 1];
    if (i_begin >= i_end) return;
    const int qi = lane & 31, hi = lane >> 5;

    // ---- LDS-DMA lane constants: every address = wave-uniform base (SGPR pair) + a lane-constant 32-bit offset (attention.hip) ----
    const unsigned smem_base = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)smem);
    constexpr int RS = D == 128 ? 4 : 8;          // K rows per LDS-DMA piece
    const int rowk = RS * wave + (D == 128 ? (lane >> 4) : (lane >> 3));
    const int gchk = D == 128 ? ((lane & 15) ^ (rowk & 15)) : ((lane & 7) ^ ((rowk >> 1) & 7));
    const int dv = 8 * wave + (lane >> 3);
    const int gchv = (lane & 7) ^ ((dv >> 1) & 7);
    const unsigned kcol = (unsigned)gchk * 16u;
    const unsigned koff_ctx = (unsigned)rowk * (unsigned)p.ldk_ctx * 2u + kcol, koff_new = (unsigned)rowk * (unsigned)p.ldk_new * 2u + kcol;
    const unsigned voff_ctx = ((unsigned)dv * (unsigned)p.ldvt_ctx + gchv * 8) * 2u;
    const unsigned voff_new = ((unsigned)dv * (unsigned)p.ldvt_new + gchv * 8) * 2u;

    // ---- SPLIT rings: one cursor per operand (see the kernel's header comment) ----
    struct Cur { int i, t, t1, ntc, lc, ln, x; const char *c, *n; };
    Cur kc = {i_begin, 0, 0, 0, 0, 0, 0, nullptr, nullptr}, vc = {i_begin, 0, 0, 0, 0, 0, 0, nullptr, nullptr};
    auto cur_load = [&](Cur& c, bool is_k) {
        const AttnItem* it = item_tab + c.i;
        c.lc = it->l_ctx; c.ln = it->l_new; c.ntc = (c.lc + 63) >> 6; c.t = it->t0; c.t1 = it->t1;
        const long g = it->g;
        if (is_k) {
            c.c = (const char*)(p.k_ctx + (long)it->kc_row0 * p.ldk_ctx + g * D);
            c.n = (const char*)(p.k_new + (long)it->kn_row0 * p.ldk_new + g * D);
        } else {
            c.c = (const char*)(p.vt_ctx + g * D * p.ldvt_ctx + it->vtc_col0);
            c.n = (const char*)(p.vt_new + g * D * p.ldvt_new + it->vtn_col0);
        }
    };
    auto cur_next = [&](Cur& c, bool is_k) {
        ++c.x;
        if (++c.t >= c.t1) {
            ++c.i;
            if (c.i < i_end) cur_load(c, is_k);
        }
    };
    // K(x) of the K cursor's tile -> K slot x % 2 (a segment's ragged last tile: rows past its end re-read the last key, masked in the scores)
    auto fetch_k = [&]() {
        if (kc.i >= i_end) return;
        const unsigned sb = __builtin_amdgcn_readfirstlane(smem_base + (kc.x & 1) * KT_BYTES + wave * 1024u);
        const bool is_ctx = kc.t < kc.ntc;
        const int ti = is_ctx ? kc.t : kc.t - kc.ntc;
        const int seglen = is_ctx ? kc.lc : kc.ln;
        const long ldk = is_ctx ? p.ldk_ctx : p.ldk_new;
        const char* kseg = a2_uniform(is_ctx ? kc.c : kc.n);
        if (ti * 64 + 64 <= seglen) {
            const char* kt = a2_uniform(kseg + (long)ti * 64 * ldk * 2);
#pragma unroll
            for (int i = 0; i < NLK; ++i) a2_glds16s(is_ctx ? koff_ctx : koff_new, a2_uniform(kt + (long)i * (RS * NW) * ldk * 2), sb + i * (NW * 1024u));
        } else {
#pragma unroll
            for (int i = 0; i < NLK; ++i) {
                int key = ti * 64 + rowk + RS * NW * i;
                key = key < seglen ? key : seglen - 1;
                a2_glds16s((unsigned)key * (unsigned)ldk * 2u + kcol, kseg, sb + i * (NW * 1024u));
            }
        }
        cur_next(kc, true);
    };
    auto fetch_v = [&]() {
        if (vc.i >= i_end) return;
        const unsigned sb = __builtin_amdgcn_readfirstlane(smem_base + 2 * KT_BYTES + (vc.x & 1) * VT_BYTES + wave * 1024u);
        const bool is_ctx = vc.t < vc.ntc;
        const int ti = is_ctx ? vc.t : vc.t - vc.ntc;
        const long ldvt = is_ctx ? p.ldvt_ctx : p.ldvt_new;
        const char* vseg = a2_uniform((is_ctx ? vc.c : vc.n) + (long)ti * 128);
#pragma unroll
        for (int i = 0; i < NLV; ++i) a2_glds16s(is_ctx ? voff_ctx : voff_new, a2_uniform(vseg + (long)i * (8 * NW) * ldvt * 2), sb + i * (NW * 1024u));
        cur_next(vc, false);
    };

    // ---- the DMA cursor: (item, tile) of the next 64-key tile to fetch, two tiles ahead of the compute cursor ----
    int d_i = i_begin, d_t = 0, d_t1 = 0, d_ntc = 0, d_lc = 0, d_ln = 0, d_slot = 0, issued = 0;
    const char *d_kc = nullptr, *d_kn = nullptr, *d_vc = nullptr, *d_vn = nullptr;
    auto d_load = [&](int idx) {
        const AttnItem* it = item_tab + idx;
        d_lc = it->l_ctx; d_ln = it->l_new; d_ntc = (d_lc + 63) >> 6; d_t = it->t0; d_t1 = it->t1;
        const long g = it->g;
        d_kc = (const char*)(p.k_ctx + (long)it->kc_row0 * p.ldk_ctx + g * D);
        d_kn = (const char*)(p.k_new + (long)it->kn_row0 * p.ldk_new + g * D);
        d_vc = (const char*)(p.vt_ctx + g * D * p.ldvt_ctx + it->vtc_col0);
        d_vn = (const char*)(p.vt_new + g * D * p.ldvt_new + it->vtn_col0);
    };
    // One 64-key tile = NLK + NLV LDS-DMA pieces per wave.  dma_prepare() resolves the tile under the cursor and advances the cursor; a
    // regular tile leaves per-piece (uniform base, LDS destination) in scalars for dma_piece(i) -- the lane offsets are the lane constants
    // koff_* / voff_* --, a segment's ragged LAST tile (per-lane clamped rows) is issued on the spot.  Returns 0 none / 1 deferred / 2 issued.
    constexpr int NP = NLK + NLV;
    const char* pc_base[NP];
    unsigned pc_dst[NP];
    bool pc_ctx = false;
    auto dma_piece = [&](int i) { a2_glds16s(i < NLK ? (pc_ctx ? koff_ctx : koff_new) : (pc_ctx ? voff_ctx : voff_new), pc_base[i], pc_dst[i]); };
    auto dma_next = [&]() -> int {
        if (d_i >= i_end) return 0;
        const unsigned sb = __builtin_amdgcn_readfirstlane(smem_base + d_slot * STAGE + wave * 1024u);
        const bool is_ctx = d_t < d_ntc;
        const int ti = is_ctx ? d_t : d_t - d_ntc;
        const int seglen = is_ctx ? d_lc : d_ln;
        const long ldk = is_ctx ? p.ldk_ctx : p.ldk_new;
        const long ldvt = is_ctx ? p.ldvt_ctx : p.ldvt_new;
        const char* kseg = a2_uniform(is_ctx ? d_kc : d_kn);
        const char* vseg = a2_uniform((is_ctx ? d_vc : d_vn) + (long)ti * 128);
        pc_ctx = is_ctx;
#pragma unroll
        for (int i = 0; i < NLK; ++i) pc_dst[i] = sb + i * (NW * 1024u);
#pragma unroll
        for (int i = 0; i < NLV; ++i) {
            pc_base[NLK + i] = a2_uniform(vseg + (long)i * (8 * NW) * ldvt * 2); pc_dst[NLK + i] = sb + KT_BYTES + i * (NW * 1024u);
        }
        int mode = 1;
        if (ti * 64 + 64 <= seglen) {
            const char* kt = a2_uniform(kseg + (long)ti * 64 * ldk * 2);
#pragma unroll
            for (int i = 0; i < NLK; ++i) pc_base[i] = a2_uniform(kt + (long)i * (RS * NW) * ldk * 2);
#pragma unroll
            for (int i = 0; i < NP; ++i) dma_piece(i);
            mode = 2;
        } else {
            // the segment's last tile: rows past its end re-read the last key (masked in the scores); per-lane offsets, issued here
#pragma unroll
            for (int i = 0; i < NLK; ++i) {
                int key = ti * 64 + rowk + RS * NW * i;
                key = key < seglen ? key : seglen - 1;
                a2_glds16s((unsigned)key * (unsigned)ldk * 2u + kcol, kseg, pc_dst[i]);
            }
#pragma unroll
            for (int i = NLK; i < NP; ++i) dma_piece(i);
            mode = 2;
        }
        d_slot = d_slot == NS - 1 ? 0 : d_slot + 1;
        ++issued;
        if (++d_t >= d_t1) {
            ++d_i;
            if (d_i < i_end) d_load(d_i);
        }
        return mode;
    };
    auto issue_next = [&]() -> bool { if ((A2_ABL & 8) && issued >= NS - 1) { if (d_i >= i_end) return false; ++issued; if (++d_t >= d_t1) { ++d_i; if (d_i < i_end) d_load(d_i); } return true; } return dma_next() != 0; };

    // ---- per-lane constants for fragment reads (attention.hip's layout; here as ONE address per operand) ----
    // K fragment ks of key block kb sits at  koff0 + kb * 32 * KROW + (((2 ks + hi) ^ kswz) << 4)  and  V^T fragment (db, j) at
    // voff + db * 4096 + (((2 j + hi) ^ vswz) << 4).  The swizzled chunk index is (hi ^ swz) ^ (2 ks), and koff0 / voff are multiples of the
    // row size, so the address is  kb0 ^ (ks << 5)  resp.  vb0 ^ (j << 5)  (+ an immediate): two registers instead of KS + 4 address
    // registers, and still one VALU per fragment read (the xor replaces the add of the ring-slot base, which goes into kb0 / vb0 per step).
    const int quad = (qi >> 2) & 3;
    const int pkey = (qi & 16) | ((((quad & 1) << 1) | (quad >> 1)) << 2) | (qi & 3);
    int kswz;
    if (D == 128) { kswz = pkey & 15; }
    else          { kswz = (pkey >> 1) & 7; }
    const int vswz = (qi >> 1) & 7;
    const unsigned kb0 = (unsigned)(pkey * KROW) | ((unsigned)(hi ^ kswz) << 4);
    const unsigned vb0 = (unsigned)((SPLIT ? 2 * KT_BYTES : KT_BYTES) + qi * 128) | ((unsigned)((hi ^ vswz) & 7) << 4);
    static_assert(STAGE % 256 == 0 && KT_BYTES % 128 == 0, "ring slots must keep the low address bits of the fragment offsets free");

    constexpr int NK = 2 * KS;                  // K fragments of a tile  (index KS*kb + ks)
    constexpr int NV = DB * 4;                  // V^T fragments of a tile (index 4*db + j)
    // kbs / vbs = kb0 / vb0 + the byte offset of the ring slot (a multiple of STAGE)
    auto kfrag = [&](unsigned kbs, int idx) { return *(const bf16x8_t*)(smem + (kbs ^ (unsigned)((idx % KS) << 5)) + (idx / KS) * 32 * KROW); };
    auto vfrag = [&](unsigned vbs, int idx) { return *(const bf16x8_t*)(smem + (vbs ^ (unsigned)((idx & 3) << 5)) + (idx >> 2) * 4096); };

    // ---- prime the tile stream: the first two tiles of this worker ----
    if (SPLIT) {
        cur_load(kc, true);
        cur_load(vc, false);
        fetch_k();                              // K(0): the first item's prologue
        fetch_k();                              // K(1), V^T(0): step 0
        fetch_v();
    } else {
        d_load(d_i);
#pragma unroll
        for (int i = 0; i < NS - 1; ++i) issue_next();
    }
    int u = 0;                                  // stream index of the compute cursor's tile (tiles are consumed in the order they are fetched)
    // counted wait until this wave's pieces of stream tile x have landed: the `issued - (x + 1)` tiles fetched after it may stay in
    // flight (LDS-DMA returns in order; anything else hipcc issued meanwhile -- the epilogue's stores -- only makes the wait longer)
    auto wait_for = [&](int x) {
        if (SPLIT) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }      // the newest fetch unit is the one the next barrier needs
        const int n = issued - (x + 1);
        if (n >= 2)      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NLK + NLV)) : "memory");
        else if (n == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLK + NLV) : "memory");
        else             asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    wait_for(0);

    int cs = 0;                                 // ring slot of the compute cursor's tile (the tile stream is consumed in order)
    for (int ci = i_begin; ci < i_end; ++ci) {
        const AttnItem* it = item_tab + ci;
        const int flags = it->flags;
        const bool hpw = flags & ATTN2_HPW, causal = flags & ATTN2_CAUSAL, partial = flags & ATTN2_PARTIAL;
        const int nrows = it->nrows, C = it->l_ctx, Lnew = it->l_new, t0 = it->t0;
        const int T = it->t1 - t0;
        const int q_row0 = it->q_row0, q_rel0 = it->q_rel0, part_slot = it->part;
        const int nt_ctx = (C + 63) >> 6;
        // this wave's rows and head: tile form = rows 32*wave.. of head h; head-per-wave form = rows 0.. of head h + wave
        const int wrow0 = hpw ? 0 : 32 * wave;
        const bool live = hpw ? wave < ((flags >> 8) & 255) : wrow0 < nrows;          // wave-uniform
        const int hw = (hpw && live) ? it->h + wave : it->h;      // (idle waves still load a Q fragment: keep them on a head that exists)
        const int qrel = q_rel0 + wrow0 + qi;                                     // row index inside the sample (causal mask)
        const bool row_ok = wrow0 + qi < nrows;
        bf16x8_t qf[KS];
        {
            const int rc = wrow0 + qi < nrows ? wrow0 + qi : nrows - 1;
            const bf16_t* qp = p.q + (long)(q_row0 + rc) * p.ldq + (long)hw * D + 8 * hi;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const bf16x8_t*)(qp + 16 * ks);
            // consume the loads HERE: hipcc then waits for them once, in front of the tile loop.  Left to itself it re-inserts counted
            // waits for them in front of every MFMA of the loop, and its count does not know the asm LDS-DMA: vmcnt(2) inside phase A
            // would stall every step on the tile that was just requested
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) asm volatile("" ::"v"(qf[ks]));
        }
        f32x16_t o[DB];
#pragma unroll
        for (int i = 0; i < DB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
        float m_run = -INFINITY, l_run = 0.f, m_use = 0.f;            // m_run in log2 units

        auto next_slot = [&](int s) { return s == NS - 1 ? 0 : s + 1; };

        auto mask_tile = [&](f32x16_t (&s)[2], int t) {
            const int tt = t0 + t;
            const bool is_ctx = tt < nt_ctx;
            const int ti = is_ctx ? tt : tt - nt_ctx;
            const int seglen = is_ctx ? C : Lnew;
            const int kbase = ti * 64;
            const bool need_mask = (kbase + 64 > seglen) || (causal && !is_ctx && (kbase + 63 > q_rel0 + wrow0));
            if (need_mask) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kbase + 32 * kb + 16 * (r >> 3) + 8 * hi + (r & 7);
                        const bool ok = key < seglen && (!causal || is_ctx || key <= qrel);
                        s[kb][r] = ok ? s[kb][r] : -INFINITY;
                    }
            }
        };
        auto row_max = [&](f32x16_t (&s)[2]) -> float {
            float a = fmaxf(s[0][0], s[1][0]), c = fmaxf(s[0][1], s[1][1]);
#pragma unroll
            for (int r = 2; r < 16; r += 2) {
                a = fmaxf(fmaxf(a, s[0][r]), s[1][r]);
                c = fmaxf(fmaxf(c, s[0][r + 1]), s[1][r + 1]);
            }
            a = fmaxf(a, c);
            return a2_max_halves(a) * p.scale_log2;
        };
        auto raise_max = [&](float mx) {
            const bool grow = mx > m_run + ATTN2_DEFER_LOG2;
            if (__any(grow)) {
                const float m_new = fmaxf(m_run, mx);
                const float alpha = __builtin_amdgcn_exp2f(m_run - (m_new == -INFINITY ? 0.f : m_new));
                m_run = m_new;
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < DB; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
                m_use = m_new == -INFINITY ? 0.f : m_new;
            }
        };

        // One pipeline step (attention.hip `step`): phase A = [S^T(t+1) -> sn] || [exp2 / pack of sc -> pf]; phase B = [O^T += V^T(t) P^T(t)] ||
        // [row max of sn].  LAST: there is no tile t+1 in THIS item (the stream's next tile belongs to the next item and is not waited for here).
        auto step = [&](f32x16_t (&sc)[2], f32x16_t (&sn)[2], int t, auto last_tag) {
            constexpr bool LAST = decltype(last_tag)::value;
            if (!(A2_ABL & 1)) {
            if (!LAST || SPLIT) wait_for(u + 1);             // this wave's pieces of tile t+1 have landed (SPLIT: F(u - 1) = {K(u + 1), V^T(u)}, needed by a LAST step too)
            asm volatile("s_barrier" ::: "memory");          // ... everyone's; and everyone is done with tile t-1, whose slot the NEXT fetch takes
            }
            unsigned sbv, sbk;
            if (SPLIT) {
                fetch_k();                                   // F(u) = {K(u + 2), V^T(u + 1)} into the slots step u - 1 read
                fetch_v();
                sbv = vb0 + (u & 1) * VT_BYTES;
                sbk = kb0 + ((u + 1) & 1) * KT_BYTES;
                ++u;
            } else {
                ++u;
                if (NS == 3 || !live) issue_next();          // 3 slots: fetch tile t+2 here; 4 slots: tile t+3 goes out inside phase B (below)
                sbv = vb0 + cs * STAGE;
                cs = next_slot(cs);
                sbk = kb0 + cs * STAGE;
            }
            if (!live) return;
            constexpr int WIN = A2_WIN;
            constexpr int NMA = NK, NMB = NV;
            constexpr int SLOTS = 16 / NMA;                 // schedule slots per chunk (1 at D = 128, 2 at D = 64)
            constexpr int SPB = 16 / NMB;
            unsigned pw[16];
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            auto softmax_slice = [&](int pi) {
                if (A2_ABL & 2) { pw[pi] = 0x3c003c00u; if (pi == 0) asm volatile("" ::"v"(sc[0][0]), "v"(sc[1][0])); return; }
                const int kb = pi / 8, r = (2 * pi) % 16;
                const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kb][r], p.scale_log2, -m_use));
                const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kb][r + 1], p.scale_log2, -m_use));
                acc[2 * (pi & 1)] += p0;
                acc[2 * (pi & 1) + 1] += p1;
                pw[pi] = pack2bf(p0, p1);
                asm volatile("" : "+v"(pw[pi]), "+v"(acc[2 * (pi & 1)]), "+v"(acc[2 * (pi & 1) + 1]));
            };
            auto kidx = [&](int j) { return (j & 1) * KS + (j >> 1); };
            auto vidx = [&](int f) { return 4 * (f % DB) + f / DB; };
            auto pfrag = [&](int f) {
                u32x4_t v4 = {pw[4 * f], pw[4 * f + 1], pw[4 * f + 2], pw[4 * f + 3]};
                return __builtin_bit_cast(bf16x8_t, v4);
            };
            bf16x8_t vf[NV];
            // ---------------- phase A ----------------
            __builtin_amdgcn_sched_barrier(0);
            if (!LAST) {
                bf16x8_t kf[NK];
#pragma unroll
                for (int j = 0; j < WIN; ++j) kf[j] = kfrag(sbk, kidx(j));
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sn[kb][r] = 0.f;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < NMA; ++m) {
                    const int kb = m & 1, ks = m >> 1;
                    sn[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[m], qf[ks], sn[kb], 0, 0, 0);
                    if (A2_ABL & 4) { if (m + WIN < NK) kf[m + WIN] = kf[m % WIN]; else if (m + WIN - NK < WIN) vf[m + WIN - NK] = kf[m % WIN]; }
                    else if (m + WIN < NK) kf[m + WIN] = kfrag(sbk, kidx(m + WIN));
                    else if (m + WIN - NK < WIN) vf[m + WIN - NK] = vfrag(sbv, vidx(m + WIN - NK));
#pragma unroll
                    for (int u = 0; u < SLOTS; ++u) softmax_slice(m * SLOTS + u);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int j = 0; j < WIN; ++j) vf[j] = vfrag(sbv, vidx(j));
#pragma unroll
                for (int slot = 0; slot < 16; ++slot) softmax_slice(slot);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!LAST) mask_tile(sn, t + 1);
            // ---------------- phase B ----------------
            __builtin_amdgcn_sched_barrier(0);
            float mx = -INFINITY;
#pragma unroll
            for (int m = 0; m < NMB; ++m) {
                const int db = m % DB, jj = m / DB;
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[m], pfrag(jj), o[db], 0, 0, 0);
                if (m + WIN < NV) vf[m + WIN] = (A2_ABL & 4) ? vf[m % WIN] : vfrag(sbv, vidx(m + WIN));
                if (!SPLIT && NS > 3 && m == NMB / 2) issue_next();   // the slot of tile t-1 is free since this step's barrier; phase B has the issue slack
#pragma unroll
                for (int u = 0; u < SPB; ++u) {
                    const int slot = m * SPB + u;
                    if (!LAST && (A2_ABL & 2)) {       // keep the S^T MFMAs alive (one element per accumulator), drop the rest of the row max
                        if (slot < 2) { mx = fmaxf(mx, sn[slot][0]); asm volatile("" : "+v"(mx)); }
                    } else if (!LAST) {
                        const int kb = slot / 8, r = (2 * slot) % 16;
                        mx = fmaxf(fmaxf(mx, sn[kb][r]), sn[kb][r + 1]);
                        asm volatile("" : "+v"(mx));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            l_run += (acc[0] + acc[1]) + (acc[2] + acc[3]);
            if (!LAST) mx = a2_max_halves(mx) * p.scale_log2;
            __builtin_amdgcn_sched_barrier(0);
            if (!LAST) raise_max(mx);
        };

        f32x16_t sA[2], sB[2];
        // prologue: this item's first tile has landed for every wave (each waited before its previous epilogue / after the priming)
        asm volatile("s_barrier" ::: "memory");
        if (live) {
            const unsigned sbk = SPLIT ? kb0 + (u & 1) * KT_BYTES : kb0 + cs * STAGE;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) sA[kb][r] = 0.f;
#pragma unroll
            for (int j = 0; j < NK; ++j) {
                const bf16x8_t kf = kfrag(sbk, j);
                sA[j / KS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[j % KS], sA[j / KS], 0, 0, 0);
            }
            mask_tile(sA, 0);
            raise_max(row_max(sA));
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
        // the next item's first tile must have landed before this wave meets the others at that item's prologue barrier; waiting HERE keeps
        // the epilogue's stores out of the count
        if (ci + 1 < i_end) wait_for(u);

        // ---- epilogue: lane owns d = 32*db + 8*u + 4*hi + (0..3) of query row qi ----
        if (live) {
            const float l_tot = a2_sum_halves(l_run);
            if (!partial) {
                const float inv = 1.0f / l_tot;
#if A2_WIDE_STORE
                // The store tail is store-ISSUE bound (cdna_hip_programming.md T21): a lane owns d = 32 db + 8 u + 4 hi + (0..3) of its row -- 8 bytes
                // per (db, u), 16 dwordx2 stores -- while lane ^ 32 owns the other half of the same 16 bytes.  One v_permlane32_swap per dword
                // trades group u + 1 of the lower half-wave against group u of the upper one: lanes < 32 end up with d = 8 (u + 1) .. + 7,
                // lanes >= 32 with d = 8 u .. + 7 -- 8 dwordx4 stores per lane, same bytes.  (Executed by the whole wave: the swap ignores no lane.)
                {
                    bf16_t* op = p.out + (long)(q_row0 + (row_ok ? wrow0 + qi : 0)) * p.ldo + (long)hw * D;
#pragma unroll
                    for (int db = 0; db < DB; ++db)
#pragma unroll
                        for (int u = 0; u < 4; u += 2) {
                            const unsigned a0 = pack2bf(o[db][4 * u] * inv, o[db][4 * u + 1] * inv), a1 = pack2bf(o[db][4 * u + 2] * inv, o[db][4 * u + 3] * inv);
                            const unsigned b0 = pack2bf(o[db][4 * u + 4] * inv, o[db][4 * u + 5] * inv), b1 = pack2bf(o[db][4 * u + 6] * inv, o[db][4 * u + 7] * inv);
                            // swap(d = B, s = A): d' = [B.lower | A.lower], s' = [B.upper | A.upper]  ->  lanes < 32: (d', s') = (B(l), B(l + 32)); lanes >= 32: (A(l - 32), A(l))
                            const auto r0 = __builtin_amdgcn_permlane32_swap(b0, a0, false, false);
                            const auto r1 = __builtin_amdgcn_permlane32_swap(b1, a1, false, false);
                            const u32x4_t v = {r0[0], r1[0], r0[1], r1[1]};
                            if (row_ok) *(u32x4_t*)(op + 32 * db + 8 * (hi ? u : u + 1)) = v;
                        }
                }
#else
                if (row_ok) {
                    bf16_t* op = p.out + (long)(q_row0 + wrow0 + qi) * p.ldo + (long)hw * D + 4 * hi;
#pragma unroll
                    for (int db = 0; db < DB; ++db)
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            u32x2_t v = {pack2bf(o[db][4 * u] * inv, o[db][4 * u + 1] * inv),
                                         pack2bf(o[db][4 * u + 2] * inv, o[db][4 * u + 3] * inv)};
                            *(u32x2_t*)(op + 32 * db + 8 * u) = v;
                        }
                }
#endif
            } else if (row_ok) {
                // un-normalised fp32 partials of this key range: O (relative to m_use), the max in use (log2 units) and the row sum
                float* pp = p.part + (long)part_slot * PSLOT;
                const int r = 32 * wave + qi;
                float* po = pp + (long)r * D + 4 * hi;
#pragma unroll
                for (int db = 0; db < DB; ++db)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        f32x4_t v = {o[db][4 * u], o[db][4 * u + 1], o[db][4 * u + 2], o[db][4 * u + 3]};
                        *(f32x4_t*)(po + 32 * db + 8 * u) = v;
                    }
                if (hi == 0) {
                    pp[256 * D + r] = m_run == -INFINITY ? -INFINITY : m_use;
                    pp[256 * D + 256 + r] = l_tot;
                }
            }
        }
    }
}

// Merge the partial slots of every key-split item: out = sum_s 2^(m_s - M) O_s / sum_s 2^(m_s - M) l_s, M = max_s m_s.
// One workgroup per (split item, 32-row block): thread = (row, eighth of the head dim), so the 8 threads of a row read 8 consecutive
// D/8-float pieces of each slot's O row (whole 128-byte lines) and write D/8 consecutive bf16.
template <int D>
__global__ __launch_bounds__(256) void attn2_combine_kernel(const AttnComb* __restrict__ comb, const float* __restrict__ part,
                                                            bf16_t* __restrict__ out, long ldo) {
    constexpr int PSLOT = 256 * (D + 2);
    constexpr int W = D / 8;                     // floats per thread
    const AttnComb c = comb[blockIdx.x >> 3];
    const int wave = blockIdx.x & 7, qi = threadIdx.x >> 3, part8 = threadIdx.x & 7;
    const int r = 32 * wave + qi;
    const bool hpw = c.flags & ATTN2_HPW;
    const bool ok = hpw ? (wave < ((c.flags >> 8) & 255) && qi < c.nrows) : r < c.nrows;
    if (!ok) return;
    const float* base = part + (long)c.slot0 * PSLOT;
    float M = -INFINITY;
    for (int s = 0; s < c.nslots; ++s) M = fmaxf(M, base[(long)s * PSLOT + 256 * D + r]);
    float den = 0.f;
    float a[W];
#pragma unroll
    for (int i = 0; i < W; ++i) a[i] = 0.f;
    for (int s = 0; s < c.nslots; ++s) {
        const float* ps = base + (long)s * PSLOT;
        const float m = ps[256 * D + r];
        const float w = m == -INFINITY ? 0.f : exp2f(m - M);
        den += w * ps[256 * D + 256 + r];
        const f32x4_t* src = (const f32x4_t*)(ps + (long)r * D + part8 * W);
#pragma unroll
        for (int i = 0; i < W / 4; ++i) {
            const f32x4_t v = src[i];
            a[4 * i] += v[0] * w; a[4 * i + 1] += v[1] * w; a[4 * i + 2] += v[2] * w; a[4 * i + 3] += v[3] * w;
        }
    }
    const float inv = 1.0f / den;
    const long row = hpw ? c.q_row0 + qi : c.q_row0 + r;
    const int h = hpw ? c.h + wave : c.h;
    bf16_t* op = out + row * ldo + (long)h * D + part8 * W;
#pragma unroll
    for (int i = 0; i < W / 4; ++i) {
        u32x2_t v = {pack2bf(a[4 * i] * inv, a[4 * i + 1] * inv), pack2bf(a[4 * i + 2] * inv, a[4 * i + 3] * inv)};
        *(u32x2_t*)(op + 4 * i) = v;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Host planner (no device code).  Layout of the plan buffer (int32):
//   [0] n_workers  [1] n_items  [2] n_comb  [3] n_slots  [4] offset of the item table  [5] offset of the combine table
//   [6] makespan (tile steps of the busiest worker)  [7] total tile steps
//   [8 .. 8 + n_workers]  worker_off      then (16-int aligned) items[n_items][16], comb[n_comb][8]
// Worker w is assumed to run on XCD w % 8 (the hardware deals workgroup ids round-robin; an assumption for speed only).
// ---------------------------------------------------------------------------------------------------------
namespace {
struct PItem { AttnItem it; int cost; };
inline int cdiv(int a, int b) { return (a + b - 1) / b; }
}

extern "C" int bagel_attn_plan(const int32_t* q_start, const int32_t* q_len, const int32_t* ctx_start, const int32_t* ctx_len,
                               const int32_t* vt_new_col, const int32_t* vt_ctx_col, int32_t batch, int32_t nq, int32_t nkv,
                               int32_t causal, int32_t n_workers, int32_t split_min_tiles, int32_t* plan, int64_t plan_ints) {
    BAGEL_REQUIRE(q_start && q_len && vt_new_col && plan, "attn_plan: null pointer");
    BAGEL_REQUIRE(nq > 0 && nkv > 0 && nq % nkv == 0, "attn_plan: bad head counts %d/%d", nq, nkv);
    BAGEL_REQUIRE(n_workers > 0 && n_workers % 8 == 0, "attn_plan: the worker count must be a multiple of the 8 XCDs");
    const int G = nq / nkv, wpx = n_workers / 8;
    if (split_min_tiles <= 0) split_min_tiles = 16;
    // ---- items of every (sample, KV head) pair, query tiles ascending ----
    std::vector<std::vector<PItem>> pairs;
    for (int b = 0; b < batch; ++b) {
        const int Lq = q_len[b], C = ctx_len ? ctx_len[b] : 0;
        if (Lq <= 0) continue;
        const int nt_ctx = cdiv(C, 64);
        const int nqt = cdiv(Lq, 256);
        for (int g = 0; g < nkv; ++g) {
            std::vector<PItem> v;
            for (int qt = 0; qt < nqt; ++qt) {
                const int nrows = std::min(256, Lq - 256 * qt);
                const int vis = causal ? std::min(Lq, 256 * qt + nrows) : Lq;        // new-segment keys any row of the tile may see
                const int T = nt_ctx + cdiv(vis, 64);
                const bool hpw = nrows <= 32 && G <= 8 && G > 1;
                AttnItem it = {};
                it.q_row0 = q_start[b] + 256 * qt; it.nrows = nrows; it.g = g;
                it.flags = (hpw ? ATTN2_HPW : 0) | (causal ? ATTN2_CAUSAL : 0) | (G << 8);
                it.q_rel0 = 256 * qt; it.kn_row0 = q_start[b]; it.l_new = Lq;
                it.kc_row0 = (C > 0 && ctx_start) ? ctx_start[b] : 0; it.l_ctx = C;
                it.vtn_col0 = vt_new_col[b]; it.vtc_col0 = (C > 0 && vt_ctx_col) ? vt_ctx_col[b] : 0;
                it.t0 = 0; it.t1 = T; it.part = -1;
                if (hpw) { it.h = g * G; v.push_back({it, T}); }
                else for (int j = 0; j < G; ++j) { it.h = g * G + j; v.push_back({it, T}); }
            }
            pairs.push_back(std::move(v));
        }
    }
    // ---- virtual pairs: when the pairs do not fill the 8 XCDs evenly, every pair is dealt into interleaved sets of items ----
    const int npairs = (int)pairs.size();
    std::vector<std::vector<PItem>> vps;
    if (npairs > 0) {
        int gcd8 = 8;
        while (npairs % gcd8) gcd8 >>= 1;
        const int qsplit = 8 / gcd8;
        for (auto& pr : pairs)
            for (int s = 0; s < qsplit; ++s) {
                std::vector<PItem> v;
                for (size_t j = s; j < pr.size(); j += qsplit) v.push_back(pr[j]);
                if (!v.empty()) vps.push_back(std::move(v));
            }
    }
    // ---- virtual pairs -> XCDs: heaviest first onto the least loaded XCD ----
    std::vector<long> vcost(vps.size(), 0);
    for (size_t i = 0; i < vps.size(); ++i) for (auto& x : vps[i]) vcost[i] += x.cost;
    std::vector<int> order(vps.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b2) { return vcost[a] > vcost[b2]; });
    std::vector<std::vector<int>> xcd_vps(8);
    long xload[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i : order) {
        int best = 0;
        for (int x = 1; x < 8; ++x) if (xload[x] < xload[best]) best = x;
        xcd_vps[best].push_back(i);
        xload[best] += vcost[i];
    }
    // ---- inside an XCD: longest-processing-time dealing of its item stream (virtual pair after virtual pair, heavy items first inside
    //      one) onto its `wpx` workers -- every item goes to the least loaded worker, which for equal costs is plain round-robin: the 32
    //      workers then sit on the same (sample, KV head) pair at the same time.  An item that would push its worker more than `slack`
    //      tile steps past the XCD's mean load is held back; the held-back items are cut along the key axis into contiguous tile ranges
    //      that fill the workers up to the mean (stream-K over the leftovers, chunks of >= min_chunk tiles), each range a sub-item
    //      that writes partials. ----
    std::vector<std::vector<AttnItem>> wl(n_workers);
    std::vector<long> wload(n_workers, 0);
    std::vector<AttnComb> combs;
    int n_slots = 0;
    const int min_chunk = std::max(split_min_tiles / 2, 1);
    for (int x = 0; x < 8; ++x) {
        std::vector<PItem> stream;
        for (int i : xcd_vps[x]) {
            std::vector<PItem> v = vps[i];
            std::stable_sort(v.begin(), v.end(), [](const PItem& a, const PItem& b2) { return a.cost > b2.cost; });
            stream.insert(stream.end(), v.begin(), v.end());
        }
        if (stream.empty()) continue;
        long tot = 0;
        for (auto& pi : stream) tot += pi.cost;
        const long mean = (tot + wpx - 1) / wpx;
        const long slack = 2;
        auto least = [&]() { int best = x; for (int j = 1; j < wpx; ++j) if (wload[j * 8 + x] < wload[best]) best = j * 8 + x; return best; };
        std::vector<PItem> held;
        for (auto& pi : stream) {
            const int w = least();
            const bool splittable = pi.cost >= split_min_tiles && pi.cost >= 2 * min_chunk;
            if (wload[w] + pi.cost > mean + slack && splittable) { held.push_back(pi); continue; }
            wl[w].push_back(pi.it);
            wload[w] += pi.cost;
        }
        // the held-back items as one stream of tiles: every cut gives the currently least loaded worker what it lacks to the mean
        for (auto& pi : held) {
            const int T = pi.it.t1 - pi.it.t0;
            std::vector<std::pair<int, int>> cuts;      // (worker, tiles)
            int left = T;
            while (left > 0) {
                const int w = least();
                long want = mean - wload[w];
                if (want < min_chunk) want = min_chunk;
                int n = (int)std::min<long>(want, left);
                if (left - n < min_chunk) n = left;                     // no crumbs
                cuts.push_back({w, n});
                wload[w] += n;
                left -= n;
            }
            if (cuts.size() == 1) { wl[cuts[0].first].push_back(pi.it); continue; }
            AttnComb c = {pi.it.q_row0, pi.it.nrows, pi.it.h, pi.it.flags, n_slots, (int)cuts.size(), 0, 0};
            combs.push_back(c);
            int at = pi.it.t0;
            for (auto& cw : cuts) {
                AttnItem sub = pi.it;
                sub.t0 = at; sub.t1 = at + cw.second; at += cw.second;
                sub.flags |= ATTN2_PARTIAL;
                sub.part = n_slots++;
                wl[cw.first].push_back(sub);
            }
        }
    }
    int n_items = 0;
    long total = 0, makespan = 0;
    for (int w = 0; w < n_workers; ++w) { n_items += (int)wl[w].size(); total += wload[w]; makespan = std::max(makespan, wload[w]); }
    const int off_items = ((8 + n_workers + 1) + 15) / 16 * 16;
    const int off_comb = off_items + 16 * n_items;
    const long need = (long)off_comb + 8L * (long)combs.size();
    BAGEL_REQUIRE(need <= plan_ints, "attn_plan: the plan needs %ld ints, the buffer holds %ld", need, (long)plan_ints);
    plan[0] = n_workers; plan[1] = n_items; plan[2] = (int)combs.size(); plan[3] = n_slots; plan[4] = off_items; plan[5] = off_comb;
    plan[6] = (int)makespan; plan[7] = (int)std::min<long>(total, 0x7fffffff);
    int at = 0;
    AttnItem* items = (AttnItem*)(plan + off_items);
    for (int w = 0; w < n_workers; ++w) {
        plan[8 + w] = at;
        for (auto& it : wl[w]) items[at++] = it;
    }
    plan[8 + n_workers] = at;
    for (int i = 8 + n_workers + 1; i < off_items; ++i) plan[i] = 0;
    AttnComb* cb = (AttnComb*)(plan + off_comb);
    for (size_t i = 0; i < combs.size(); ++i) cb[i] = combs[i];
    return BAGEL_OK;
}

extern "C" int bagel_attn_planned_bf16(const void* q, int64_t ldq, const void* k_new, int64_t ldk_new, const void* vt_new,
                                       int64_t ldvt_new, const void* k_ctx, int64_t ldk_ctx, const void* vt_ctx, int64_t ldvt_ctx,
                                       void* out, int64_t ldo, const int32_t* plan_dev, int32_t n_workers, int32_t n_comb,
                                       int32_t off_items, int32_t off_comb, void* partials, int32_t head_dim, float softmax_scale,
                                       hipStream_t stream) {
    BAGEL_REQUIRE(q && k_new && vt_new && out && plan_dev, "attn_planned: null pointer");
    BAGEL_REQUIRE(ldq % 8 == 0 && ldk_new % 8 == 0 && ldvt_new % 8 == 0 && ldo % 4 == 0 && ldk_ctx % 8 == 0 && ldvt_ctx % 8 == 0,
                  "attn_planned: leading dims must keep 16-byte alignment");
    BAGEL_REQUIRE(softmax_scale > 0.f, "attn_planned: softmax_scale must be positive");
    BAGEL_REQUIRE(n_comb == 0 || partials, "attn_planned: the plan has key-split items but no partials workspace was given");
    BAGEL_REQUIRE(n_workers > 0, "attn_planned: no workers");
    Attn2Params p;
    p.q = (const bf16_t*)q; p.ldq = ldq;
    p.k_new = (const bf16_t*)k_new; p.ldk_new = ldk_new;
    p.vt_new = (const bf16_t*)vt_new; p.ldvt_new = ldvt_new;
    p.k_ctx = (const bf16_t*)k_ctx; p.ldk_ctx = ldk_ctx;
    p.vt_ctx = (const bf16_t*)vt_ctx; p.ldvt_ctx = ldvt_ctx;
    p.out = (bf16_t*)out; p.ldo = ldo;
    const int* worker_off = plan_dev + 8;
    const AttnItem* items = (const AttnItem*)(plan_dev + off_items);
    p.part = (float*)partials;
    p.scale_log2 = softmax_scale * 1.4426950408889634f;
    const dim3 grid(n_workers), block(512);
    // a plan with MORE workers than the device has CUs asks for two resident workgroups per CU: the split-ring kernel (64 KB of LDS per workgroup)
    static int cus_of[64] = {0};
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return bagel_set_error(BAGEL_ERR_LAUNCH, "attn_planned: cannot query the device");
    if (dev >= 0 && dev < 64 && cus_of[dev] > 0) cus = cus_of[dev];
    else {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            return bagel_set_error(BAGEL_ERR_LAUNCH, "attn_planned: cannot query the device");
        if (dev >= 0 && dev < 64) cus_of[dev] = cus;
    }
    const bool split = n_workers > cus;
    if (head_dim == 128) {
        constexpr int smem = A2_SLOTS * (64 * 256 + 128 * 128), smem_split = 2 * (64 * 256 + 128 * 128);
        if (split) {
            if (int rc = bagel_enable_lds((const void*)attn2_kernel<128, true>, smem_split, "attn2_kernel<128, split>")) return rc;
            hipLaunchKernelGGL((attn2_kernel<128, true>), grid, block, smem_split, stream, p, worker_off, items);
        } else {
            if (int rc = bagel_enable_lds((const void*)attn2_kernel<128, false>, smem, "attn2_kernel<128>")) return rc;
            hipLaunchKernelGGL((attn2_kernel<128, false>), grid, block, smem, stream, p, worker_off, items);
        }
    } else if (head_dim == 64) {
        constexpr int smem = A2_SLOTS * (64 * 128 + 64 * 128), smem_split = 2 * (64 * 128 + 64 * 128);
        if (split) {
            if (int rc = bagel_enable_lds((const void*)attn2_kernel<64, true>, smem_split, "attn2_kernel<64, split>")) return rc;
            hipLaunchKernelGGL((attn2_kernel<64, true>), grid, block, smem_split, stream, p, worker_off, items);
        } else {
            if (int rc = bagel_enable_lds((const void*)attn2_kernel<64, false>, smem, "attn2_kernel<64>")) return rc;
            hipLaunchKernelGGL((attn2_kernel<64, false>), grid, block, smem, stream, p, worker_off, items);
        }
    } else {
        return bagel_set_error(BAGEL_ERR_UNSUPPORTED, "attn_planned: head_dim %d not in {64,128} (pad the head)", head_dim);
    }
    if (int rc = bagel_check_launch("attn2_kernel")) return rc;
    if (n_comb > 0) {
        const AttnComb* cb = (const AttnComb*)(plan_dev + off_comb);
        if (head_dim == 128) hipLaunchKernelGGL((attn2_combine_kernel<128>), dim3(8 * n_comb), dim3(256), 0, stream, cb, (const float*)partials, (bf16_t*)out, (long)ldo);
        else hipLaunchKernelGGL((attn2_combine_kernel<64>), dim3(8 * n_comb), dim3(256), 0, stream, cb, (const float*)partials, (bf16_t*)out, (long)ldo);
        return bagel_check_launch("attn2_combine_kernel");
    }
    return BAGEL_OK;
}

// TEST / TOOLING HOOK: resident workgroups per CU of the planned attention kernel (unified 4-slot ring, or the split 2 + 2 ring) as the runtime's occupancy
// calculator sees them -- registers, LDS and wave slots together (tools/attn2_probe.py --workers).
extern "C" int bagel_debug_attn_occupancy(int32_t head_dim, int32_t split) {
    int n = 0;
    hipError_t e;
    if (head_dim == 128) {
        constexpr int smem = A2_SLOTS * (64 * 256 + 128 * 128), smem_split = 2 * (64 * 256 + 128 * 128);
        if (int rc = bagel_enable_lds(split ? (const void*)attn2_kernel<128, true> : (const void*)attn2_kernel<128, false>, split ? smem_split : smem, "attn2_kernel<128>")) return rc;
        e = split ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn2_kernel<128, true>, 512, smem_split)
                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn2_kernel<128, false>, 512, smem);
    } else if (head_dim == 64) {
        constexpr int smem = A2_SLOTS * (64 * 128 + 64 * 128), smem_split = 2 * (64 * 128 + 64 * 128);
        if (int rc = bagel_enable_lds(split ? (const void*)attn2_kernel<64, true> : (const void*)attn2_kernel<64, false>, split ? smem_split : smem, "attn2_kernel<64>")) return rc;
        e = split ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn2_kernel<64, true>, 512, smem_split)
                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn2_kernel<64, false>, 512, smem);
    } else {
        return bagel_set_error(BAGEL_ERR_UNSUPPORTED, "attn_occupancy: head_dim %d not in {64,128}", head_dim);
    }
    if (e != hipSuccess) return bagel_set_error(BAGEL_ERR_LAUNCH, "attn_occupancy: %s", hipGetErrorString(e));
    return n;
}
