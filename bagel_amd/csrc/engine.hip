// Batch-1 text decode: a CHAIN of projections as ONE persistent launch on a loader / consumer weight-streaming engine.
//
// The launch form of a decoded token (decode.py, decode.hip) is 6 kernels per layer -- qkv, attention, combine, o, gate+up, down -- each
// of which ramps the chip up, streams its weights and drains: 35 of the 107 us of a 7B layer are launch floors and graph edges
// (DESIGN 3.5).  This kernel runs a list of up to 4 dependent projections  y_p = epilogue_p(W_p . norm_p(y_{p-1}))  -- for a decoder
// layer: o_proj(+residual) -> RMSNorm + gate/up (SwiGLU) -> down(+residual) -> RMSNorm + the NEXT layer's qkv (or the final norm +
// lm_head) -- replacing Qwen2MLP.forward / the F.linear calls of qwen2_navit.py:591-594,515-517 and modeling_qwen2.py:200-201 at Lq = 1
// (bagel.py:930-1000).  The attention stays its own two launches: that seam is an all-to-all over keys, the cut the guide prescribes.
//
// Structure (MI355X_MICROARCH.md rows ldsdma-fill / nt-weights / engine-vs-launches; cdna_hip_programming.md 5.6, Guideline 16):
//   * one workgroup per CU, all co-resident (grid == CU count); the last `nloaders` waves (2) are LOADERS, the others (6) CONSUMERS;
//   * every workgroup owns a fixed, contiguous share of the weight-row pairs of every phase.  The loaders walk that static list of units
//     (unit = one row PAIR of K <= 5120, or one K-quarter of a pair for the long rows of the down projection; loader j takes units j, j + 2, ..
//     and owns their ring slots) and stream them with `global_load_lds_dwordx4 ... nt` into a ring of LDS slots -- they never wait for an
//     activation, so while the consumers sit in the hand-off between two projections the ring fills with the next projection's weights (the
//     "prefetch credit" that is supposed to pay for the hop);
//   * consumer c takes units c, c + NC, c + 2 NC ... of the same list: waits for the slot's `full` word (LDS), pulls the unit into registers,
//     hands the slot back (`freed`), runs the lane-FMA body of gemv_kernel on the staged activation vector (same chunk -> lane map, same
//     accumulation order, same wave reduction, same epilogue roundings: the engine is BIT-IDENTICAL to the chain of bagel_gemv_bf16 launches,
//     tests/test_engine_gpu.py), and lane 0 writes the outputs write-through (`sc1`: agent-scope relaxed atomic stores);
//   * hand-off between phases (Guideline 16, form R1): when the last consumer of a workgroup has drained its stores it sets the
//     workgroup's flag word of that phase (agent-scope relaxed store).  Consumer 0 of every workgroup polls the n_wg flag words with ONE
//     wave (relaxed `sc1` loads, s_sleep between sweeps), then the activation vector is read with `sc1` loads (no acquire fence: the
//     payload was stored write-through and the loads bypass L1) and staged in LDS -- with the Qwen2RMSNorm of gemv_kernel fused in --
//     together with the workgroup's bias / residual dwords;
//   * every spin is bounded: on a timeout the workgroup records a code in `status` and all of its waves leave; the host checks it.
// MEASURED (profiles/r05_decode_engine.log): 94.4 us per 7B layer against 83.8 us for the four launches -- the launch form stays the default
// (BAGEL_DECODE_ENGINE=1 selects this kernel); the comments below keep what each step of the way bought.
//
// Flags live in caller memory that must be ZERO when the launch starts (one memset per token covers every layer's slice).
#include "common.h"
#include <stdlib.h>

#define EPI_NONE 0
#define EPI_GELU_TANH 1
#define EPI_SILU 2
#define EPI_SWIGLU16 3

#define ENG_MAXPH 4
#define ENG_MAX_SLOTS 8
#define ENG_MAX_DPAIRS 32                 // split-K pairs one workgroup may own (partials live in LDS)
#define ENG_SYNC_BYTES 2560
#define ENG_RB_PAIRS 64                   // row pairs per workgroup whose bias / residual dwords are staged in LDS at the hand-off
#define ENG_LDS_MAX (160 * 1024)
#define ENG_SPIN_LDS (1u << 23)           // bound of an LDS poll loop (~1 s)
#define ENG_SPIN_GLOBAL (1u << 19)        // bound of a flag sweep loop (~1-2 s)

struct EngPhase {
    const bf16_t* A;          // activation vector [K] (phase 0: written by an earlier launch; later phases: the previous phase's C)
    const bf16_t* W; long ldw;
    const bf16_t* bias;       // [N] or null
    const bf16_t* norm_w;     // [K] or null: Qwen2RMSNorm fused into the staging
    const bf16_t* R;          // [N] residual or null (may alias C)
    bf16_t* C;                // [N] ([N/2] for SwiGLU16)
    int N, K, epi;
    int kind;                 // 0: a unit = one row pair; 2: a unit = one K-quarter of a row pair (gemv_kernel<., 4>)
    int in_launch;            // A is produced by the previous phase of THIS launch -> wait for every workgroup's flag first
    int ngr;                  // 64-chunk groups per row = ceil(K / 512)
    int gq;                   // groups per K-quarter (kind 2)
};

struct EngParams {
    EngPhase ph[ENG_MAXPH];
    int nph;
    float eps;
    unsigned* flags;          // [nph][n_wg], zero at launch
    unsigned* status;         // [4]: first failure code (0 = ok), never cleared by the kernel
    int xs_bytes, slot_bytes, nslot, depth, nt, nloaders;
    int abl;                  // timing-only ablations (BAGEL_ENGINE_ABL, results are WRONG): 1 consumers skip the arithmetic, 2 loader issues no DMA,
                              // 4 no flag polling between phases (bits may be combined)
    unsigned long long* trace;   // diagnostic: [n_wg][ENG_MAXPH][16] event times (100 MHz ticks) or null
};

struct EngSync {              // at smem + xs_bytes + nslot * slot_bytes
    unsigned full[ENG_MAX_SLOTS];
    unsigned freed[ENG_MAX_SLOTS];
    unsigned abort_;
    unsigned go[ENG_MAXPH];
    unsigned staged[ENG_MAXPH];
    unsigned done[ENG_MAXPH];
    unsigned pair_cnt[ENG_MAX_DPAIRS];
    float part[ENG_MAX_DPAIRS][4][2];
    int pair0[ENG_MAXPH], npair[ENG_MAXPH], ubeg[ENG_MAXPH + 1];      // this workgroup's first pair of every phase; its units' position in the workgroup's sequence
    unsigned long long tr[ENG_MAXPH][14];           // trace events of this workgroup (see ENG_T_*), copied out by the waves that own them
    unsigned rb_bias[ENG_RB_PAIRS], rb_res[ENG_RB_PAIRS];   // this workgroup's bias / residual dwords of the current phase (rows 2 i, 2 i + 1 of its share)
};
// Phase / launch parameters are copied out of the kernel-argument segment ONCE and pinned in scalar registers: left to itself hipcc re-loads them
// with s_load inside the unit loops (8-10 dependent scalar loads per unit in the consumer, each waited for: with no DMA, no arithmetic and no flag
// polling the first version still took 83 us per layer -- 0.6 us of pure protocol per unit, profiles/r05_decode_engine.log ablation 7).
__device__ __forceinline__ void eng_pin(EngPhase& P) {
    asm volatile("" : "+s"(P.A), "+s"(P.W), "+s"(P.ldw), "+s"(P.bias), "+s"(P.norm_w), "+s"(P.R), "+s"(P.C));
    asm volatile("" : "+s"(P.N), "+s"(P.K), "+s"(P.epi), "+s"(P.kind), "+s"(P.in_launch), "+s"(P.ngr), "+s"(P.gq));
}
struct EngKnobs { int xs_bytes, slot_bytes, nslot, depth, abl; unsigned* flags; unsigned* status; bool tr; };
__device__ __forceinline__ EngKnobs eng_knobs(const EngParams& p) {
    EngKnobs k{p.xs_bytes, p.slot_bytes, p.nslot, p.depth, p.abl, p.flags, p.status, p.trace != nullptr};
    asm volatile("" : "+s"(k.xs_bytes), "+s"(k.slot_bytes), "+s"(k.nslot), "+s"(k.depth), "+s"(k.abl), "+s"(k.flags), "+s"(k.status));
    return k;
}
static_assert(sizeof(EngSync) <= ENG_SYNC_BYTES, "EngSync does not fit its LDS reservation");

// trace slots (per phase): loader first issue / last issue / ticks blocked on a free slot / ticks blocked in counted waits; consumer 0 stage begin /
// flags seen / staged; consumer c last unit done (7 + c, c < 3) and ticks it waited for full slots (10 + c); 13 the workgroup's flag store
#define ENG_T_LFIRST 0
#define ENG_T_LLAST 1
#define ENG_T_LFREE 2
#define ENG_T_LVM 3
#define ENG_T_SBEGIN 4
#define ENG_T_SPOLL 5
#define ENG_T_SDONE 6
#define ENG_T_UDONE 7
#define ENG_T_UWAIT 10
#define ENG_T_FLAG 13
__device__ __forceinline__ unsigned long long eng_now() { return __builtin_amdgcn_s_memrealtime(); }

#define ENG_WG __HIP_MEMORY_SCOPE_WORKGROUP
#define ENG_AG __HIP_MEMORY_SCOPE_AGENT

typedef __attribute__((address_space(1))) unsigned eng_gu32;
typedef __attribute__((address_space(1))) unsigned short eng_gu16;
typedef __attribute__((address_space(1))) unsigned long long eng_gu64;

__device__ __forceinline__ unsigned eng_ld_u32(const void* p) { return __hip_atomic_load((const eng_gu32*)p, __ATOMIC_RELAXED, ENG_AG); }
__device__ __forceinline__ u32x4_t eng_ld_chunk(const bf16_t* p) {          // 16 bytes past L1 (two sc1 8-byte loads)
    const unsigned long long a = __hip_atomic_load((const eng_gu64*)p, __ATOMIC_RELAXED, ENG_AG);
    const unsigned long long b = __hip_atomic_load((const eng_gu64*)p + 1, __ATOMIC_RELAXED, ENG_AG);
    return u32x4_t{(unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32)};
}
__device__ __forceinline__ void eng_st_u32(void* p, unsigned v) { __hip_atomic_store((eng_gu32*)p, v, __ATOMIC_RELAXED, ENG_AG); }
__device__ __forceinline__ void eng_st_u16(void* p, unsigned short v) { __hip_atomic_store((eng_gu16*)p, v, __ATOMIC_RELAXED, ENG_AG); }

// ---- failure path: record the first code, tell the other waves of this workgroup to leave -----------------------------------------
__device__ __forceinline__ void eng_fail(EngSync* sy, unsigned* status, unsigned code, int lane) {
    if (lane == 0) {
        unsigned expected = 0u;
        __hip_atomic_compare_exchange_strong(status, &expected, code | (blockIdx.x << 8), __ATOMIC_RELAXED, __ATOMIC_RELAXED, ENG_AG);
        __hip_atomic_store(&sy->abort_, 1u, __ATOMIC_RELAXED, ENG_WG);
    }
}
// wait until the LDS word reaches `want` (monotonic words); false = aborted / timed out
__device__ __forceinline__ bool eng_wait_lds(unsigned* word, unsigned want, EngSync* sy, unsigned* status, unsigned code, int lane) {
    for (unsigned spins = 0;; ++spins) {
        if (__hip_atomic_load(word, __ATOMIC_ACQUIRE, ENG_WG) >= want) return true;
        if (__hip_atomic_load(&sy->abort_, __ATOMIC_RELAXED, ENG_WG)) return false;
        if (spins > ENG_SPIN_LDS) { eng_fail(sy, status, code, lane); return false; }
        __builtin_amdgcn_s_sleep(1);
    }
}

// s_waitcnt vmcnt(m) with the largest m <= n out of a short ladder, for a wave-uniform run-time n (the LDS-DMA instructions are invisible to hipcc's
// own counting; gfx9 has no register form of the wait).  Waiting for FEWER outstanding instructions than allowed is always safe.  The ladder holds
// the sums that occur with 7- and 10-group rows (14 / 20 instructions per unit) exactly; a descending compare chain costs 2 scalar instructions per
// rung and the common values sit at its top.  (A 64-way switch came back from hipcc's structurizer as a ~100-instruction walk per wait.)
__device__ __forceinline__ void eng_wait_vmcnt(int n) {
#define ENG_VM(i) if (n >= i) { asm volatile("s_waitcnt vmcnt(" #i ")" ::: "memory"); return; }
    ENG_VM(60) ENG_VM(54) ENG_VM(48) ENG_VM(42) ENG_VM(40) ENG_VM(34) ENG_VM(28) ENG_VM(24) ENG_VM(20) ENG_VM(16) ENG_VM(14) ENG_VM(12) ENG_VM(8)
    ENG_VM(6) ENG_VM(4) ENG_VM(2)
#undef ENG_VM
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// N x 1 KB of one weight row -> LDS: lane l's 16 bytes of piece i go to lds_dst + 1024 i + 16 l.  Address = wave-uniform 64-bit base (SGPR pair) +
// the lane's 32-bit byte offset (one VGPR) + the instruction's immediate, and the immediate moves the LDS address as well: ONE base, ONE offset register
// and ONE M0 value serve four instructions.  M0 is NOT preserved: hipcc has no use for it in this kernel (gfx9 LDS instructions take no M0; checked on
// the ISA by tests/test_host_cpu.py::test_decode_engine_isa_invariants: no m0 outside these statements, no scratch).  History (profiles/r05_decode_engine.log): the first version issued every
// piece from a scalar loop (M0 save / set / restore, 64-bit lane address arithmetic, two branches per instruction): 72 ns per instruction, 2.4 TB/s.
template <int N, bool NT>
__device__ __forceinline__ void eng_dma(unsigned voff, const void* sbase_uniform, unsigned lds_dst_uniform) {
#define ENG_GL(off) "global_load_lds_dwordx4 %0, %1 offset:" #off
    if constexpr (NT) {
        if constexpr (N == 1) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t" ENG_GL(0) " nt" :: "v"(voff), "s"(sbase_uniform), "s"(lds_dst_uniform) : "memory");
        if constexpr (N == 2) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t" ENG_GL(0) " nt\n\t" ENG_GL(1024) " nt" :: "v"(voff), "s"(sbase_uniform), "s"(lds_dst_uniform) : "memory");
        if constexpr (N == 3) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t" ENG_GL(0) " nt\n\t" ENG_GL(1024) " nt\n\t" ENG_GL(2048) " nt" :: "v"(voff), "s"(sbase_uniform), "s"(lds_dst_uniform) : "memory");
        if constexpr (N == 4) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t" ENG_GL(0) " nt\n\t" ENG_GL(1024) " nt\n\t" ENG_GL(2048) " nt\n\t" ENG_GL(3072) " nt" :: "v"(voff), "s"(sbase_uniform), "s"(lds_dst_uniform) : "memory");
    } else {
        if constexpr (N == 1) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t" ENG_GL(0) :: "v"(voff), "s"(sbase_uniform), "s"(lds_dst_uniform) : "memory");
        if constexpr (N == 2) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t" ENG_GL(0) "\n\t" ENG_GL(1024) :: "v"(voff), "s"(sbase_uniform), "s"(lds_dst_uniform) : "memory");
        if constexpr (N == 3) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t" ENG_GL(0) "\n\t" ENG_GL(1024) "\n\t" ENG_GL(2048) :: "v"(voff), "s"(sbase_uniform), "s"(lds_dst_uniform) : "memory");
        if constexpr (N == 4) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t" ENG_GL(0) "\n\t" ENG_GL(1024) "\n\t" ENG_GL(2048) "\n\t" ENG_GL(3072) :: "v"(voff), "s"(sbase_uniform), "s"(lds_dst_uniform) : "memory");
    }
#undef ENG_GL
}
// `ng` whole groups of a row, starting at the wave-uniform address `src` -> the LDS image at dst, in runs of four; then (ragged > 0) one last group of
// `ragged` chunks whose lanes past the end re-read the row's last chunk (their activations are 0).  lane16 = 16 * lane.
template <bool NT>
__device__ __forceinline__ void eng_dma_row(const char* src, int ng, int ragged, unsigned dst, unsigned lane16) {
    unsigned voff = lane16;
    int g = 0;
    for (; g + 4 <= ng; g += 4) {
        eng_dma<4, NT>(voff, src, dst);
        voff += 4096u;
        dst += 4096u;
    }
    const int rest = ng - g;
    if (rest == 3) eng_dma<3, NT>(voff, src, dst);
    else if (rest == 2) eng_dma<2, NT>(voff, src, dst);
    else if (rest == 1) eng_dma<1, NT>(voff, src, dst);
    if (ragged > 0) {
        unsigned l = lane16 >> 4;
        if (l > (unsigned)(ragged - 1)) l = (unsigned)(ragged - 1);
        eng_dma<1, NT>((unsigned)ng * 1024u + l * 16u, src, dst + (unsigned)rest * 1024u);
    }
}

// ---- the static unit list of this workgroup: contiguous shares of the row pairs of every phase (tables in LDS: read back through
//      readfirstlane, so that every index derived from them stays in scalar registers and no kernel-argument access turns into a
//      vector-memory load that the DMA accounting below would not know about) --------------------------------------------------------
__device__ __forceinline__ void eng_share_init(const EngParams& p, EngSync* sy) {
    const long G = gridDim.x, b = blockIdx.x;
    int u = 0;
    for (int i = 0; i < ENG_MAXPH; ++i) {
        sy->ubeg[i] = u;
        int lo = 0, hi = 0;
        if (i < p.nph) {
            const long NP = p.ph[i].N >> 1;
            lo = (int)(b * NP / G);
            hi = (int)((b + 1) * NP / G);
            u += (hi - lo) * (p.ph[i].kind == 2 ? 4 : 1);
        }
        sy->pair0[i] = lo;
        sy->npair[i] = hi - lo;
    }
    sy->ubeg[ENG_MAXPH] = u;
}
__device__ __forceinline__ int eng_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// groups [g_lo, g_hi) of 64 chunks a unit covers
__device__ __forceinline__ void eng_unit_groups(const EngPhase& P, int lu, int& g_lo, int& g_hi) {
    if (P.kind == 2) {
        const int j = lu & 3;
        g_lo = j * P.gq;
        g_hi = (g_lo + P.gq < P.ngr) ? g_lo + P.gq : P.ngr;
        if (g_lo > g_hi) g_lo = g_hi;
    } else {
        g_lo = 0;
        g_hi = P.ngr;
    }
}
__device__ __forceinline__ void eng_unit_rows(const EngPhase& P, int pp, int& r0, int& r1) {
    if (P.epi == EPI_SWIGLU16) { r0 = ((pp >> 4) << 5) + (pp & 15); r1 = r0 + 16; }
    else { r0 = 2 * pp; r1 = r0 + 1; }
}

// =====================================================================================================================
// LOADER wave.  One short loop per unit: (1) wait until the ring slot has been handed back (unit i - nslot; it was published at least
// nslot - depth units ago, so this wait never depends on an unpublished unit), (2) issue the unit's DMAs, (3) once `depth` younger units are
// behind it, retire the oldest unit in flight: a counted `s_waitcnt vmcnt(n)` with n = the younger units' instructions, then its `full` word.
// The per-unit instruction counts of the units in flight travel in one 64-bit scalar (8 bits each).  `depth` is clamped on the host so that
// depth < nslot and depth x the largest unit stays below the 63 instructions a counted wait can express.
// (The first version kept a general "make room" loop with four exit conditions around every unit: ~100 scalar instructions and ~25 branches per
// unit on ONE wave = 0.5 us per unit, as long as the DMA issue itself; profiles/r05_decode_engine.log, ablation rows.)
// =====================================================================================================================
template <bool NT>
__device__ __forceinline__ void eng_loader(const EngParams& p, unsigned char* smem, EngSync* sy, int lane, int lj, int nl) {
    // loader lj of nl takes the units q = lj, lj + nl, ... of the workgroup's sequence (the ring has a multiple of nl slots, so it also owns the slots
    // of its units); `issued` / `published` count ITS units, each wave has its own vmcnt
    const EngKnobs k = eng_knobs(p);
    const int nph = eng_uni(p.nph);
    const unsigned ring_base = eng_uni((int)(unsigned)(unsigned long)(const __attribute__((address_space(3))) unsigned char*)smem) + (unsigned)k.xs_bytes;
    const unsigned lane16 = (unsigned)lane * 16u;
    int issued = 0, published = 0, inflight = 0;
    int q = lj;                                   // next unit of this loader in the workgroup's sequence
    int islot = lj % k.nslot, pslot = islot, pq = lj;
    const int slot_step = nl % k.nslot;
    unsigned long long counts = 0ull;            // instruction count of this loader's unit u at bits 8 (u & 7)
    __builtin_amdgcn_s_setprio(3);
    const bool tr = k.tr && lj == 0;
    unsigned long long t_free = 0ull, t_vm = 0ull;
    auto retire = [&]() {                          // oldest unit in flight: landed once only the younger units' DMAs are outstanding
        const int n_old = (int)((counts >> (8 * (published & 7))) & 0xffull);
        const unsigned long long t0 = tr ? eng_now() : 0ull;
        eng_wait_vmcnt(inflight - n_old);
        if (tr) t_vm += eng_now() - t0;
        __hip_atomic_store(&sy->full[pslot], (unsigned)(pq + 1), __ATOMIC_RELAXED, ENG_WG);
        inflight -= n_old;
        ++published;
        pq += nl;
        pslot += slot_step;
        if (pslot >= k.nslot) pslot -= k.nslot;
    };
    for (int ph = 0; ph < nph; ++ph) {
        EngPhase P = p.ph[ph];
        eng_pin(P);
        const int ubeg = eng_uni(sy->ubeg[ph]), uend = eng_uni(sy->ubeg[ph + 1]);
        const int pair0 = eng_uni(sy->pair0[ph]);
        const int nch = P.K >> 3;
        const int nfull = nch >> 6;               // whole 64-chunk groups of a row
        const bool split = P.kind == 2, swiglu = P.epi == EPI_SWIGLU16;
        const long row_bytes = P.ldw * 2;
        t_free = t_vm = 0ull;
        bool first = true;
        for (; q < uend; q += nl) {
            const int lu = q - ubeg;
            // ---- (1) the slot
            const unsigned long long t_room = tr ? eng_now() : 0ull;
            if (q >= k.nslot) {
                const unsigned want = (unsigned)(q - k.nslot + 1);
                for (unsigned spins = 0; __hip_atomic_load(&sy->freed[islot], __ATOMIC_RELAXED, ENG_WG) < want;) {
                    if (__hip_atomic_load(&sy->abort_, __ATOMIC_RELAXED, ENG_WG)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
                    if (++spins > ENG_SPIN_LDS) { eng_fail(sy, k.status, 0x10u + (unsigned)ph, lane); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            if (tr) {
                const unsigned long long now = eng_now();
                t_free += now - t_room;
                if (lane == 0) {
                    if (first) sy->tr[ph][ENG_T_LFIRST] = now;
                    sy->tr[ph][ENG_T_LLAST] = now; sy->tr[ph][ENG_T_LFREE] = t_free; sy->tr[ph][ENG_T_LVM] = t_vm;
                }
                first = false;
            }
            // ---- (2) issue: row r0's groups, then row r1's, 1 KB per instruction
            const int pp = pair0 + (split ? (lu >> 2) : lu);
            const int r0 = swiglu ? ((pp >> 4) << 5) + (pp & 15) : 2 * pp;
            int g_lo = 0, g_hi = P.ngr;
            if (split) {
                g_lo = (lu & 3) * P.gq;
                g_hi = (g_lo + P.gq < P.ngr) ? g_lo + P.gq : P.ngr;
                if (g_lo > g_hi) g_lo = g_hi;
            }
            const int ng = g_hi - g_lo;
            const int ng_full = (g_hi <= nfull ? g_hi : nfull) - g_lo;            // whole groups of this unit (>= 0: a ragged group is the row's last)
            const int ragged = g_hi > nfull ? nch - nfull * 64 : 0;
            const int n_new = (k.abl & 2) ? 0 : 2 * ng;
            const unsigned dst0 = ring_base + (unsigned)islot * (unsigned)k.slot_bytes;
            if (!(k.abl & 2)) {
                const char* src0 = (const char*)P.W + (long)r0 * row_bytes + (long)g_lo * 1024;
                const char* src1 = src0 + (swiglu ? 16 * row_bytes : row_bytes);
                eng_dma_row<NT>(src0, ng_full > 0 ? ng_full : 0, ragged, dst0, lane16);
                eng_dma_row<NT>(src1, ng_full > 0 ? ng_full : 0, ragged, dst0 + (unsigned)ng * 1024u, lane16);
            }
            counts = (counts & ~(0xffull << (8 * (issued & 7)))) | ((unsigned long long)n_new << (8 * (issued & 7)));
            inflight += n_new;
            ++issued;
            islot += slot_step;
            if (islot >= k.nslot) islot -= k.nslot;
            // ---- (3) retire the oldest unit once `depth` younger ones are behind it
            if (issued - published > k.depth) retire();
        }
    }
    while (published < issued) retire();
    if (tr && lane < 4 * ENG_MAXPH) {                 // (no DMA is outstanding any more: ordinary stores are safe here)
        const int ph = lane >> 2, k = lane & 3;
        p.trace[((long)blockIdx.x * ENG_MAXPH + ph) * 16 + k] = sy->tr[ph][k];
    }
}

// =====================================================================================================================
// CONSUMER waves
// =====================================================================================================================
// Stage the activation vector of phase `ph` in LDS.  Normalised inputs: consumer 0 alone, reproducing the thread -> chunk map and the
// summation tree of gemv_kernel's RMSNorm (256 virtual threads: chunks t and t + 256; wave sums of 64 consecutive threads; the four wave
// sums added left to right).  Plain inputs: every consumer copies its share.
__device__ __forceinline__ bool eng_stage(const EngPhase& P, const EngKnobs& k, float eps, int ph, unsigned char* smem, EngSync* sy, int cw, int nc, int lane) {
    const int nch = P.K >> 3;
    bf16_t* xs = (bf16_t*)smem;
    const int G = gridDim.x;
    const bool tr = k.tr;
    if (cw == 0) {
        if (tr && lane == 0) sy->tr[ph][ENG_T_SBEGIN] = eng_now();
        if (P.in_launch && !(k.abl & 4)) {                          // every workgroup has published the previous phase's outputs
            const unsigned* fl = k.flags + (long)(ph - 1) * G;
            for (unsigned spins = 0;; ++spins) {
                bool ok = true;
                for (int i = lane; i < G; i += 64) ok &= eng_ld_u32(fl + i) != 0u;
                if (__all(ok)) break;
                if (__hip_atomic_load(&sy->abort_, __ATOMIC_RELAXED, ENG_WG)) return false;
                if (spins > ENG_SPIN_GLOBAL) { eng_fail(sy, k.status, 0x20u + (unsigned)ph, lane); return false; }
                __builtin_amdgcn_s_sleep(4);
            }
        }
        if (tr && lane == 0) sy->tr[ph][ENG_T_SPOLL] = eng_now();
        // epilogue operands of this workgroup's row pairs -> LDS, so that no consumer has a load in its unit loop: gfx9 counts loads and stores on one
        // counter, and a wait for a residual dword would also wait for the previous unit's write-through store (~1.5 us each)
        unsigned rb_b = 0u, rb_r = 0u;
        const int rb_n = P.epi == EPI_SWIGLU16 ? 0 : eng_uni(sy->npair[ph]);
        const bool rb_on = rb_n <= ENG_RB_PAIRS && lane < rb_n;
        if (rb_on) {
            const int r0 = 2 * (eng_uni(sy->pair0[ph]) + lane);
            if (P.bias) rb_b = *(const unsigned*)(P.bias + r0);
            if (P.R) rb_r = eng_ld_u32(P.R + r0);
        }
        if (P.norm_w) {
            u32x4_t xr[4][2], gw[4][2];
            float red[4];
#pragma unroll
            for (int vw = 0; vw < 4; ++vw)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int c = vw * 64 + lane + 256 * i;
                    xr[vw][i] = c < nch ? eng_ld_chunk(P.A + (long)c * 8) : u32x4_t{0u, 0u, 0u, 0u};
                    gw[vw][i] = c < nch ? *(const u32x4_t*)(P.norm_w + (long)c * 8) : u32x4_t{0u, 0u, 0u, 0u};
                }
#pragma unroll
            for (int vw = 0; vw < 4; ++vw) {
                float ss = 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a = lo2f(xr[vw][i][e]), b = hi2f(xr[vw][i][e]);
                        ss += a * a + b * b;
                    }
                red[vw] = wave_sum(ss);
            }
            const float inv = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)P.K + eps);
#pragma unroll
            for (int vw = 0; vw < 4; ++vw)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int c = vw * 64 + lane + 256 * i;
                    if (c < nch) {
                        u32x4_t v = xr[vw][i];
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            v[e] = pack2bf(bfround(lo2f(v[e]) * inv) * lo2f(gw[vw][i][e]), bfround(hi2f(v[e]) * inv) * hi2f(gw[vw][i][e]));
                        *(u32x4_t*)(xs + (long)c * 8) = v;
                    }
                }
            if (rb_on) { sy->rb_bias[lane] = rb_b; sy->rb_res[lane] = rb_r; }
            if (lane == 0) __hip_atomic_fetch_add(&sy->staged[ph], (unsigned)nc, __ATOMIC_RELEASE, ENG_WG);
            if (tr && lane == 0) sy->tr[ph][ENG_T_SDONE] = eng_now();
            return true;
        }
        if (rb_on) { sy->rb_bias[lane] = rb_b; sy->rb_res[lane] = rb_r; }
        if (lane == 0) __hip_atomic_store(&sy->go[ph], 1u, __ATOMIC_RELEASE, ENG_WG);
    } else {
        if (P.norm_w) return eng_wait_lds(&sy->staged[ph], (unsigned)nc, sy, k.status, 0x30u + (unsigned)ph, lane);
        if (!eng_wait_lds(&sy->go[ph], 1u, sy, k.status, 0x40u + (unsigned)ph, lane)) return false;
    }
    // plain copy, split over the consumers (16 chunks per lane and batch in flight)
    for (int c0 = cw * 64 + lane; c0 < nch; c0 += nc * 64 * 16) {
        u32x4_t v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int c = c0 + j * nc * 64;
            if (c < nch) v[j] = eng_ld_chunk(P.A + (long)c * 8);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int c = c0 + j * nc * 64;
            if (c < nch) *(u32x4_t*)(xs + (long)c * 8) = v[j];
        }
    }
    if (lane == 0) __hip_atomic_fetch_add(&sy->staged[ph], 1u, __ATOMIC_RELEASE, ENG_WG);
    const bool ok = eng_wait_lds(&sy->staged[ph], (unsigned)nc, sy, k.status, 0x50u + (unsigned)ph, lane);
    if (tr && cw == 0 && lane == 0) sy->tr[ph][ENG_T_SDONE] = eng_now();
    return ok;
}

// epilogue of one row pair by lane 0: the rounding points of gemv_body (bias, activation, residual, SwiGLU16)
__device__ __forceinline__ void eng_epilogue(const EngPhase& P, int pp, int r0, float s0, float s1, float eb0, float eb1, float er0, float er1) {
    if (P.epi == EPI_SWIGLU16) {
        const float gg = bfround(s0), uu = bfround(s1);
        eng_st_u16(P.C + pp, f2bf(bfround(silu_f(gg)) * uu));
    } else {
        float o[2] = {s0, s1};
        const float eb[2] = {eb0, eb1}, er[2] = {er0, er1};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (P.bias) o[t] += eb[t];
            if (P.epi == EPI_GELU_TANH) o[t] = gelu_tanh_f(bfround(o[t]));
            else if (P.epi == EPI_SILU) o[t] = silu_f(bfround(o[t]));
            if (P.R) o[t] = bfround(o[t]) + er[t];
        }
        eng_st_u32(P.C + r0, pack2bf(o[0], o[1]));                  // rows r0, r0 + 1: one aligned dword
    }
}

// the 16 FMAs of one 16-byte chunk pair: gemv_fma's order (low halves into a.0, high halves into a.1, element by element)
__device__ __forceinline__ void eng_fma8(const u32x4_t wa, const u32x4_t wb, const u32x4_t xv, float& a00, float& a01, float& a10, float& a11) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float xl = lo2f(xv[e]), xh = hi2f(xv[e]);
        a00 = fmaf(lo2f(wa[e]), xl, a00);
        a01 = fmaf(hi2f(wa[e]), xh, a01);
        a10 = fmaf(lo2f(wb[e]), xl, a10);
        a11 = fmaf(hi2f(wb[e]), xh, a11);
    }
}
// A unit of NG whole groups: the slot's 2 NG KB go to registers in one burst and the slot is handed back BEFORE the arithmetic -- a ring slot is
// then busy for the DMA flight plus one LDS read burst instead of the flight plus the whole lane-FMA time (first version: slots cycled in 4.3 us, the
// loader sat blocked on "no free slot" 40 % of the gate/up phase: profiles/r05_decode_engine.log).  Same accumulation order as the generic loop.
template <int NG>
__device__ __forceinline__ void eng_unit_fma(const unsigned char* sl, int rowimg, const bf16_t* xs_lane, unsigned* freed, unsigned seq, int lane,
                                             float& a00, float& a01, float& a10, float& a11) {
    u32x4_t wa[NG], wb[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        wa[g] = *(const u32x4_t*)(sl + g * 1024);
        wb[g] = *(const u32x4_t*)(sl + rowimg + g * 1024);
    }
    if (lane == 0) __hip_atomic_store(freed, seq, __ATOMIC_RELEASE, ENG_WG);       // (the release waits for the reads above)
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const u32x4_t xv = *(const u32x4_t*)(xs_lane + g * 512);
        eng_fma8(wa[g], wb[g], xv, a00, a01, a10, a11);
    }
}
// wave_sum (common.h) with the four inner butterfly steps on the DPP crossbar instead of ds_bpermute: same pairing, same order (32, 16, 8, 4, 2, 1),
// so the same bits; xor 8 = row_ror:8, xor 4 = row_half_mirror then quad_perm [3,2,1,0], xor 2 / xor 1 = quad_perm [2,3,0,1] / [1,0,3,2]
template <int CTRL>
__device__ __forceinline__ float eng_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float eng_wave_sum(float v) {
    v += __shfl_xor(v, 32, 64);
    v += __shfl_xor(v, 16, 64);
    v += eng_dpp<0x128>(v);
    v += eng_dpp<0x1B>(eng_dpp<0x141>(v));
    v += eng_dpp<0x4E>(v);
    v += eng_dpp<0xB1>(v);
    return v;
}

__device__ __forceinline__ void eng_consumer(const EngParams& p, unsigned char* smem, EngSync* sy, int cw, int nc, int lane) {
    const EngKnobs k = eng_knobs(p);
    const int nph = eng_uni(p.nph);
    float eps = p.eps;
    asm volatile("" : "+s"(eps));
    const bf16_t* xs = (const bf16_t*)smem;
    const unsigned char* ring = smem + k.xs_bytes;
    int q = cw;
    int slot = cw % k.nslot;
    const int slot_step = nc % k.nslot;
    for (int ph = 0; ph < nph; ++ph) {
        EngPhase P = p.ph[ph];
        eng_pin(P);
        if (!eng_stage(P, k, eps, ph, smem, sy, cw, nc, lane)) return;
        const int ubeg = eng_uni(sy->ubeg[ph]), uend = eng_uni(sy->ubeg[ph + 1]);
        const int pair0 = eng_uni(sy->pair0[ph]);
        const int nch = P.K >> 3;
        const bool swiglu = P.epi == EPI_SWIGLU16;
        const bool rb_lds = eng_uni(sy->npair[ph]) <= ENG_RB_PAIRS;
        const bool tr = k.tr;
        unsigned long long t_wait = 0ull;
        for (; q < uend; q += nc) {
            const int lu = q - ubeg;
            const int pp = pair0 + (P.kind == 2 ? (lu >> 2) : lu);
            int r0, r1, g_lo, g_hi;
            eng_unit_rows(P, pp, r0, r1);
            eng_unit_groups(P, lu, g_lo, g_hi);
            // epilogue operands first: their latency hides under the wait for the slot
            // (raw dwords, converted in the epilogue: a use here would make hipcc wait for them here)
            unsigned bias_raw = 0u, r_raw = 0u;
            if (lane == 0 && P.kind != 2 && !swiglu && !rb_lds) {    // (shares too large for the LDS table: loaded here; split-K: by whoever completes the pair)
                if (P.bias) bias_raw = *(const unsigned*)(P.bias + r0);          // r0 is even: rows r0, r0 + 1 in one aligned dword
                if (P.R) r_raw = eng_ld_u32(P.R + r0);
            }
            const unsigned long long t_w0 = tr ? eng_now() : 0ull;
            if (!eng_wait_lds(&sy->full[slot], (unsigned)(q + 1), sy, k.status, 0x60u + (unsigned)ph, lane)) return;
            if (tr) t_wait += eng_now() - t_w0;
            const unsigned char* sl = ring + (long)slot * k.slot_bytes + lane * 16;
            const int ng = g_hi - g_lo;
            const int rowimg = ng * 1024;
            const int ch_hi = (g_hi * 64 < nch) ? g_hi * 64 : nch;
            float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;
            const bool whole = g_hi * 64 <= nch;                       // no ragged group in this unit
            if (k.abl & 1) { if (lane == 0) __hip_atomic_store(&sy->freed[slot], (unsigned)(q + 1), __ATOMIC_RELEASE, ENG_WG); }
            else if (whole && ng == 7) eng_unit_fma<7>(sl, rowimg, xs + ((long)g_lo * 64 + lane) * 8, &sy->freed[slot], (unsigned)(q + 1), lane, a00, a01, a10, a11);
            else if (whole && ng == 10) eng_unit_fma<10>(sl, rowimg, xs + ((long)g_lo * 64 + lane) * 8, &sy->freed[slot], (unsigned)(q + 1), lane, a00, a01, a10, a11);
            else {
                // any other geometry: one group ahead in registers (the LDS latency of group g + 1 hides under the FMAs of group g), slot handed back last
                u32x4_t wa, wb, xv;
                {
                    const int ch = g_lo * 64 + lane;
                    const bool ok = ch < ch_hi;
                    wa = *(const u32x4_t*)(sl);
                    wb = *(const u32x4_t*)(sl + rowimg);
                    xv = *(const u32x4_t*)(xs + (long)(ok ? ch : 0) * 8);
                    if (!ok) xv = u32x4_t{0u, 0u, 0u, 0u};
                }
                for (int g = g_lo; g < g_hi; ++g) {
                    u32x4_t na = wa, nb = wb, nx = xv;
                    if (g + 1 < g_hi) {
                        const int ch = (g + 1) * 64 + lane;
                        const bool ok = ch < ch_hi;
                        na = *(const u32x4_t*)(sl + (g + 1 - g_lo) * 1024);
                        nb = *(const u32x4_t*)(sl + rowimg + (g + 1 - g_lo) * 1024);
                        nx = *(const u32x4_t*)(xs + (long)(ok ? ch : 0) * 8);
                        if (!ok) nx = u32x4_t{0u, 0u, 0u, 0u};
                    }
                    eng_fma8(wa, wb, xv, a00, a01, a10, a11);
                    wa = na; wb = nb; xv = nx;
                }
                if (lane == 0) __hip_atomic_store(&sy->freed[slot], (unsigned)(q + 1), __ATOMIC_RELEASE, ENG_WG);
            }
            slot += slot_step;
            if (slot >= k.nslot) slot -= k.nslot;
            float s0 = eng_wave_sum(a00 + a01);
            float s1 = eng_wave_sum(a10 + a11);
            if (lane == 0) {
                if (P.kind == 2) {
                    const int lp = lu >> 2, j = lu & 3;
                    sy->part[lp][j][0] = s0;
                    sy->part[lp][j][1] = s1;
                    const unsigned old = __hip_atomic_fetch_add(&sy->pair_cnt[lp], 1u, __ATOMIC_ACQ_REL, ENG_WG);
                    if ((old & 3u) == 3u) {                            // the four K quarters have met: fixed summation order
                        s0 = (sy->part[lp][0][0] + sy->part[lp][1][0]) + (sy->part[lp][2][0] + sy->part[lp][3][0]);
                        s1 = (sy->part[lp][0][1] + sy->part[lp][1][1]) + (sy->part[lp][2][1] + sy->part[lp][3][1]);
                        if (rb_lds) { bias_raw = sy->rb_bias[pp - pair0]; r_raw = sy->rb_res[pp - pair0]; }
                        else {
                            if (P.bias) bias_raw = *(const unsigned*)(P.bias + r0);
                            if (P.R) r_raw = eng_ld_u32(P.R + r0);
                        }
                        eng_epilogue(P, pp, r0, s0, s1, lo2f(bias_raw), hi2f(bias_raw), lo2f(r_raw), hi2f(r_raw));
                    }
                } else {
                    if (rb_lds && !swiglu) { bias_raw = sy->rb_bias[pp - pair0]; r_raw = sy->rb_res[pp - pair0]; }
                    eng_epilogue(P, pp, r0, s0, s1, lo2f(bias_raw), hi2f(bias_raw), lo2f(r_raw), hi2f(r_raw));
                }
            }
        }
        if (tr && lane == 0 && cw < 3) { sy->tr[ph][ENG_T_UDONE + cw] = eng_now(); sy->tr[ph][ENG_T_UWAIT + cw] = t_wait; }
        // publish: every consumer drains its own stores, the last one to arrive sets the workgroup's flag of this phase
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            const unsigned old = __hip_atomic_fetch_add(&sy->done[ph], 1u, __ATOMIC_ACQ_REL, ENG_WG);
            if (old == (unsigned)(nc - 1)) {
                if (ph + 1 < nph) eng_st_u32(k.flags + (long)ph * gridDim.x + blockIdx.x, 1u);
                if (tr) sy->tr[ph][ENG_T_FLAG] = eng_now();
            }
        }
    }
    if (k.tr) {                                    // copy out what the consumers own (slots 4..15); consumer 0 waits for the others' last phase
        if (cw == 0) {
            eng_wait_lds(&sy->done[nph - 1], (unsigned)nc, sy, k.status, 0x70u, lane);
            if (lane < 10 * ENG_MAXPH) {
                const int ph = lane / 10, k = 4 + lane % 10;
                p.trace[((long)blockIdx.x * ENG_MAXPH + ph) * 16 + k] = sy->tr[ph][k];
            }
        }
    }
}

__global__ __launch_bounds__(512) void decode_engine_kernel(const EngParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char eng_smem[];
    EngSync* sy = (EngSync*)(eng_smem + p.xs_bytes + (long)p.nslot * p.slot_bytes);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = eng_uni(tid >> 6);
    const int nwaves = blockDim.x >> 6;
    for (int i = tid; i < (int)(sizeof(EngSync) / 4); i += blockDim.x) ((unsigned*)sy)[i] = 0u;
    __syncthreads();
    if (tid == 0) eng_share_init(p, sy);
    __syncthreads();
    const int nl = p.nloaders, nc = nwaves - nl;                 // waves 0 .. nc-1 consume, the last nl waves load
    if (wave >= nc) {
        if (p.nt) eng_loader<true>(p, eng_smem, sy, lane, wave - nc, nl);
        else eng_loader<false>(p, eng_smem, sy, lane, wave - nc, nl);
    }
    else eng_consumer(p, eng_smem, sy, wave, nc, lane);
}

// ---- host -------------------------------------------------------------------------------------------------------------------------
static int eng_env(const char* name, int dflt) {
    const char* e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}

extern "C" int bagel_decode_engine_workgroups(void) {
    // per device (a process may drive several, and a CU-masked device reports its own count); a racing first call computes the same value twice
    static int n_of[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess)
        return bagel_set_error(BAGEL_ERR_LAUNCH, "decode_engine: cannot query the device");
    const bool cached = dev >= 0 && dev < 64;
    if (cached && n_of[dev] > 0) return n_of[dev];
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
        return bagel_set_error(BAGEL_ERR_LAUNCH, "decode_engine: cannot query the device");
    if (cached) n_of[dev] = n;
    return n;
}

extern "C" int bagel_decode_engine_sync_bytes(int32_t n_phases) {
    const int n_wg = bagel_decode_engine_workgroups();
    if (n_wg <= 0) return n_wg;
    if (n_phases <= 0) return 0;
    return (n_phases * n_wg * 4 + 255) / 256 * 256;
}

static int eng_launch(const void* const* ptrs, const int64_t* dims, int32_t n_phases, float eps, void* sync_ws, void* status, void* trace,
                      hipStream_t stream);
extern "C" int bagel_decode_engine_bf16(const void* const* ptrs, const int64_t* dims, int32_t n_phases, float eps, void* sync_ws,
                                        void* status, hipStream_t stream) {
    return eng_launch(ptrs, dims, n_phases, eps, sync_ws, status, nullptr, stream);
}
extern "C" int bagel_decode_engine_traced_bf16(const void* const* ptrs, const int64_t* dims, int32_t n_phases, float eps, void* sync_ws,
                                               void* status, void* trace, hipStream_t stream) {
    BAGEL_REQUIRE(trace, "decode_engine_traced: null trace buffer");
    return eng_launch(ptrs, dims, n_phases, eps, sync_ws, status, trace, stream);
}
static int eng_launch(const void* const* ptrs, const int64_t* dims, int32_t n_phases, float eps, void* sync_ws, void* status, void* trace,
                      hipStream_t stream) {
    BAGEL_REQUIRE(ptrs && dims && sync_ws && status, "decode_engine: null pointer");
    BAGEL_REQUIRE(n_phases >= 1 && n_phases <= ENG_MAXPH, "decode_engine: 1..%d phases (got %d)", ENG_MAXPH, n_phases);
    const int n_wg = bagel_decode_engine_workgroups();
    if (n_wg <= 0) return n_wg;
    EngParams p;
    p.nph = n_phases;
    p.eps = eps;
    p.flags = (unsigned*)sync_ws;
    p.status = (unsigned*)status;
    p.trace = (unsigned long long*)trace;
    int max_k = 0, slot = 0;
    for (int i = 0; i < n_phases; ++i) {
        EngPhase& P = p.ph[i];
        const void* const* q = ptrs + 6 * i;
        const int64_t* d = dims + 4 * i;
        P.A = (const bf16_t*)q[0]; P.W = (const bf16_t*)q[1]; P.bias = (const bf16_t*)q[2]; P.norm_w = (const bf16_t*)q[3];
        P.R = (const bf16_t*)q[4]; P.C = (bf16_t*)q[5];
        P.N = (int)d[0]; P.K = (int)d[1]; P.ldw = d[2]; P.epi = (int)d[3];
        BAGEL_REQUIRE(P.A && P.W && P.C, "decode_engine: phase %d: null A / W / C", i);
        BAGEL_REQUIRE(P.K > 0 && (P.K % 8) == 0 && (P.ldw % 8) == 0, "decode_engine: phase %d: K / ldw must be multiples of 8", i);
        BAGEL_REQUIRE(P.N > 0 && (P.N % 2) == 0, "decode_engine: phase %d: N = %d must be even", i, P.N);
        BAGEL_REQUIRE(P.epi >= 0 && P.epi <= 3, "decode_engine: phase %d: unknown epilogue %d", i, P.epi);
        BAGEL_REQUIRE(P.epi != EPI_SWIGLU16 || ((P.N % 32) == 0 && !P.bias && !P.R), "decode_engine: phase %d: swiglu needs N %% 32 == 0, no bias / residual", i);
        BAGEL_REQUIRE((((uintptr_t)P.A | (uintptr_t)P.W | (uintptr_t)P.norm_w) & 15) == 0, "decode_engine: phase %d: A / W / norm_w must be 16-byte aligned", i);
        BAGEL_REQUIRE((((uintptr_t)P.C | (uintptr_t)P.R) & 3) == 0, "decode_engine: phase %d: C / R must be 4-byte aligned", i);
        BAGEL_REQUIRE(!P.norm_w || (P.K >> 3) <= 512, "decode_engine: phase %d: the fused RMSNorm serves K <= 4096", i);
        BAGEL_REQUIRE(i == 0 || P.A == (const bf16_t*)p.ph[i - 1].C, "decode_engine: phase %d must read the previous phase's output", i);
        BAGEL_REQUIRE(i == 0 || P.K == (p.ph[i - 1].epi == EPI_SWIGLU16 ? p.ph[i - 1].N / 2 : p.ph[i - 1].N),
                      "decode_engine: phase %d: K = %d does not match the previous phase's output length", i, P.K);
        P.in_launch = i > 0;
        P.ngr = (P.K / 8 + 63) / 64;
        // the split-K form of bagel_gemv_bf16 (launch_gemv_any): long un-normalised rows, few of them
        P.kind = (!P.norm_w && P.K >= 8192 && P.N / 2 <= 8192) ? 2 : 0;
        P.gq = (P.ngr + 3) / 4;
        const int unit_groups = P.kind == 2 ? P.gq : P.ngr;
        BAGEL_REQUIRE(2 * unit_groups <= 60, "decode_engine: phase %d: K = %d gives %d KB units (at most 60 LDS-DMA instructions in flight)", i, P.K, 2 * unit_groups);
        if (P.kind == 2) {
            const long max_pairs = ((long)P.N / 2 + n_wg - 1) / n_wg;
            BAGEL_REQUIRE(max_pairs <= ENG_MAX_DPAIRS, "decode_engine: phase %d: %ld split-K pairs per workgroup (at most %d)", i, max_pairs, ENG_MAX_DPAIRS);
        }
        if (2 * unit_groups * 1024 > slot) slot = 2 * unit_groups * 1024;
        if (P.K > max_k) max_k = P.K;
    }
    p.xs_bytes = (max_k * 2 + 1023) / 1024 * 1024;
    p.slot_bytes = slot;
    int nslot = (ENG_LDS_MAX - p.xs_bytes - ENG_SYNC_BYTES) / slot;
    const int want = eng_env("BAGEL_ENGINE_SLOTS", ENG_MAX_SLOTS);
    if (nslot > want) nslot = want;
    if (nslot > ENG_MAX_SLOTS) nslot = ENG_MAX_SLOTS;
    int waves = eng_env("BAGEL_ENGINE_WAVES", 8);
    if (waves < 2) waves = 2;
    if (waves > 8) waves = 8;
    int nl = eng_env("BAGEL_ENGINE_LOADERS", 2);
    if (nl < 1) nl = 1;
    if (nl > 2) nl = 2;
    if (nl >= waves) nl = waves - 1;
    nslot -= nslot % nl;                                       // every loader owns the slots of its units
    BAGEL_REQUIRE(nslot >= 2 * nl && nslot >= 3, "decode_engine: the LDS ring holds %d slots of %d bytes beside a %d-byte activation vector", nslot, slot, p.xs_bytes);
    p.nslot = nslot;
    p.nloaders = nl;
    // units one loader keeps unpublished: fewer than its own slots, and `depth` of the largest unit must stay countable (<= 63 DMA instructions)
    p.depth = eng_env("BAGEL_ENGINE_DEPTH", 2);
    if (p.depth < 1) p.depth = 1;
    if (p.depth > nslot / nl - 1) p.depth = nslot / nl - 1;
    while (p.depth > 1 && p.depth * (slot / 1024) > 63) --p.depth;
    p.nt = eng_env("BAGEL_ENGINE_NT", 1);
    p.abl = eng_env("BAGEL_ENGINE_ABL", 0);
    const int smem = p.xs_bytes + nslot * slot + ENG_SYNC_BYTES;
    if (smem > 48 * 1024)
        if (int rc = bagel_enable_lds((const void*)decode_engine_kernel, ENG_LDS_MAX, "decode_engine_kernel")) return rc;
    hipLaunchKernelGGL(decode_engine_kernel, dim3(n_wg), dim3(64 * waves), smem, stream, p);
    return bagel_check_launch("decode_engine_kernel");
}
