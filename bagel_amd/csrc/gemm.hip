// bf16 MFMA GEMM  C[M,N] = A[M,K] * W[N,K]^T  (+bias, activation, SwiGLU pairing, residual) for gfx950.
//
// Replaces the reference's cuBLAS F.linear call sites on the hot path (qwen2_navit.py:515-517,529-536,
// 591-594; modeling_qwen2.py:200-201; bagel.py:803,832,978; siglip_navit.py:216-218,243,256-258).
//
// MoT routing (qwen2_navit.py:526-548, 784-787, 812-820) is done here instead of by gather/scatter
// kernels: one launch runs up to two row GROUPS, each with its own weight matrix and an optional
// row-index list (rows of A/C/R that belong to the group: `packed_text_indexes` -> und expert,
// `packed_vae_token_indexes` -> gen expert).  The glds loader takes per-lane global addresses, so the
// gather costs nothing extra.
//
// Structure: BMxBNx64 tile, (WM x WN) waves of 64 lanes, mfma_f32_16x16x32_bf16 with SWAPPED operands
// (D = Wfrag * Afrag -> each lane owns 4 consecutive n of one output row => 8-byte stores, in-register
// bias / SwiGLU pairing).  A/W tiles go HBM -> LDS with global_load_lds_dwordx4 (16 B/lane, LDS image
// lane-linear, XOR swizzle applied on the SOURCE chunk so ds_read_b128 fragment reads are conflict-free),
// double-buffered, counted vmcnt so the next tile's DMA stays in flight across the barrier.
// XCD-aware tile order: blocks that land on the same XCD (blockIdx % 8) walk a 4-M-tile-wide band so the
// A/W panels they share stay in that XCD's 4 MiB L2.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

#define EPI_NONE 0
#define EPI_GELU_TANH 1
#define EPI_SILU 2
#define EPI_SWIGLU16 3

struct GemmGroup {
    const bf16_t* W;
    const bf16_t* bias;
    const int* a_rows;   // physical A row of logical row i (nullptr: identity)
    const int* c_rows;   // physical C / residual row of logical row i (nullptr: identity)
    int M;
    int tile0;         // first M-tile index of this group
};

struct GemmParams {
    const bf16_t* A;
    const bf16_t* R;
    bf16_t* C;
    long lda, ldw, ldr, ldc;
    int N, K;
    int tiles_m, tiles_n;
    int gm;         // band height in M tiles of the XCD-local tile walk (ping-pong kernel)
    int epi;
    int ngroups;
    GemmGroup g[2];
    // FP8 operands (gemm_pq_kernel<MODE, true>): A / W are e4m3 bytes, sa[physical A row] and sw[n] the fp32 de-quantisation scales
    const float* sa;
    const float* sw;
    // QOUT (gemm_pq_kernel<0, true, false, true>): the SwiGLU result leaves the kernel as e4m3 bytes quantised with a scale the CALLER supplies per physical C row
    // (delayed scaling: it comes from the previous denoise step's row maxima), and the row maxima of THIS step's values are collected for the next one
    unsigned char* Cq;       // [physical row][ldcq bytes]
    long ldcq;
    const float* cs;         // scale in use, per physical C row (the consumer GEMM's `sa`)
    unsigned* cmax;          // per physical C row: max |value| seen (fp32 bit pattern, atomicMax; non-negative floats order like unsigned integers)
    // split launches of the persistent kernel (launch_gemm_pq): this launch covers the tile indices [w_begin, w_begin + w_count) of the
    // XCD-aware walk; ksplit > 1 cuts every one of them into ksplit work items along K that leave fp32 partial tiles in `part`
    // (gemm_pq_kernel<4>) for gemm_splitk_reduce_kernel
    int w_begin, w_count, ksplit;
    float* part;
};

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    // 16 bytes per lane, LDS destination = wave-uniform base + lane*16.
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void gemm_tn_kernel(const GemmParams p) {
    constexpr int NW = WM * WN;
    constexpr int MB = BM / WM / 16;   // 16-row fragments per wave along M
    constexpr int NB = BN / WN / 16;   // 16-col fragments per wave along N
    constexpr int A_BYTES = BM * 128;
    constexpr int B_BYTES = BN * 128;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int LA = BM / 8 / NW;    // glds instructions per wave for the A tile (8 rows each)
    constexpr int LB = BN / 8 / NW;
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile/wave mismatch");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // ---- XCD-aware tile mapping (bijective for any grid size) ----
    const int nblk = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    constexpr int GM = 4;   // band height in M tiles
    const int band = bid / (GM * p.tiles_n);
    const int band_rows = min(GM, p.tiles_m - band * GM);
    const int inb = bid - band * GM * p.tiles_n;
    const int tm = band * GM + inb % band_rows;
    const int tn = inb / band_rows;

    const int gi = (p.ngroups > 1 && tm >= p.g[1].tile0) ? 1 : 0;
    const bf16_t* __restrict__ Wg = p.g[gi].W;
    const bf16_t* __restrict__ biasg = p.g[gi].bias;
    const int* __restrict__ a_rows = p.g[gi].a_rows;
    const int* __restrict__ c_rows = p.g[gi].c_rows;
    const int Mg = p.g[gi].M;
    const int m0 = (tm - p.g[gi].tile0) * BM;
    const int n0 = tn * BN;

    // ---- per-lane global source pointers for the LDS-DMA loads ----
    // instruction j covers tile rows [8j, 8j+8): lane -> row 8j + lane/8, LDS chunk lane%8,
    // global chunk = (lane%8) ^ ((row>>1)&7)   (source-side swizzle; the LDS image stays linear)
    const char* pa[LA];
    const char* pb[LB];
    int ka[LA], kb[LB];   // k offset (elements) of this lane's chunk inside a k-tile
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int j = wave + i * NW;
        const int row = j * 8 + (lane >> 3);
        const int gch = (lane & 7) ^ ((row >> 1) & 7);
        ka[i] = gch * 8;
        int m = m0 + row;
        m = m < Mg ? m : Mg - 1;
        const long prow = a_rows ? (long)a_rows[m] : (long)m;
        pa[i] = (const char*)(p.A + prow * p.lda) + gch * 16;
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const int j = wave + i * NW;
        const int row = j * 8 + (lane >> 3);
        const int gch = (lane & 7) ^ ((row >> 1) & 7);
        kb[i] = gch * 8;
        int n = n0 + row;
        n = n < p.N ? n : p.N - 1;
        pb[i] = (const char*)(Wg + (long)n * p.ldw) + gch * 16;
    }

    // ---- per-lane LDS fragment read offsets ----
    // fragment row r = base16 + (lane&15), k-chunk (lane>>4) + 4*kh, physical chunk = chunk ^ ((r>>1)&7)
    const int fr = lane & 15;
    const int sw = fr >> 1;
    const int ch0 = (lane >> 4) ^ sw;          // kh = 0
    const int ch1 = ((lane >> 4) + 4) ^ sw;    // kh = 1
    const int a_off = (wm * (BM / WM) + fr) * 128;
    const int b_off = A_BYTES + (wn * (BN / WN) + fr) * 128;

    f32x4_t acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + 63) >> 6;
    const bool ktail = (p.K & 63) != 0;
    const char* zsrc = (const char*)bagel_zero16;

    auto issue = [&](int stage, int kt) {
        char* sb = smem + stage * STAGE;
        const long koff = (long)kt * 128;
        if (ktail && kt == nk - 1) {   // last, partial k-tile: chunks at k >= K read zeros
            const int k0 = kt * 64;
#pragma unroll
            for (int i = 0; i < LA; ++i) glds16(k0 + ka[i] < p.K ? pa[i] + koff : zsrc, sb + (wave + i * NW) * 1024);
#pragma unroll
            for (int i = 0; i < LB; ++i) glds16(k0 + kb[i] < p.K ? pb[i] + koff : zsrc, sb + A_BYTES + (wave + i * NW) * 1024);
        } else {
#pragma unroll
            for (int i = 0; i < LA; ++i) glds16(pa[i] + koff, sb + (wave + i * NW) * 1024);
#pragma unroll
            for (int i = 0; i < LB; ++i) glds16(pb[i] + koff, sb + A_BYTES + (wave + i * NW) * 1024);
        }
    };

    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int st = kt & 1;
        if (kt + 1 < nk) {
            issue(st ^ 1, kt + 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LA + LB) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_barrier" ::: "memory");

        const char* sb = smem + st * STAGE;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const int ch = kh ? ch1 : ch0;
            bf16x8_t af[MB], bfr[NB];
#pragma unroll
            for (int i = 0; i < MB; ++i) af[i] = *(const bf16x8_t*)(sb + a_off + i * 2048 + ch * 16);
#pragma unroll
            for (int j = 0; j < NB; ++j) bfr[j] = *(const bf16x8_t*)(sb + b_off + j * 2048 + ch * 16);
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
        // every ds_read of this stage has returned before any wave may overwrite it
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    // ---- epilogue: lane owns C[row = m][n .. n+3], n = frag base + (lane>>4)*4 ----
    const int nsub = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int m = m0 + wm * (BM / WM) + i * 16 + fr;
        if (m >= Mg) continue;
        const long prow = c_rows ? (long)c_rows[m] : (long)m;
        if (p.epi == EPI_SWIGLU16) {
#pragma unroll
            for (int j = 0; j < NB; j += 2) {
                const int n = n0 + wn * (BN / WN) + j * 16 + nsub;   // gate columns; up = +16
                if (n >= p.N) continue;
                const int oc = ((n0 + wn * (BN / WN) + j * 16) >> 1) + nsub;
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = bfround(acc[i][j][e]);
                    const float u = bfround(acc[i][j + 1][e]);
                    o[e] = bfround(silu_f(g)) * u;
                }
                u32x2_t v = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
                *(u32x2_t*)(p.C + prow * p.ldc + oc) = v;
            }
        } else {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int n = n0 + wn * (BN / WN) + j * 16 + nsub;
                if (n >= p.N) continue;
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = acc[i][j][e];
                if (biasg) {
                    const u32x2_t bv = *(const u32x2_t*)(biasg + n);
                    o[0] += lo2f(bv[0]); o[1] += hi2f(bv[0]); o[2] += lo2f(bv[1]); o[3] += hi2f(bv[1]);
                }
                if (p.epi == EPI_GELU_TANH) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = gelu_tanh_f(bfround(o[e]));
                } else if (p.epi == EPI_SILU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = silu_f(bfround(o[e]));
                }
                if (p.R) {
                    const u32x2_t rv = *(const u32x2_t*)(p.R + prow * p.ldr + n);
                    o[0] = bfround(o[0]) + lo2f(rv[0]); o[1] = bfround(o[1]) + hi2f(rv[0]);
                    o[2] = bfround(o[2]) + lo2f(rv[1]); o[3] = bfround(o[3]) + hi2f(rv[1]);
                }
                u32x2_t v = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
                *(u32x2_t*)(p.C + prow * p.ldc + n) = v;
            }
        }
    }
}


// =========================================================================================================
// "Ping-pong" 256x256x64 kernel (variant 3) for the large-M projections of the denoise forward.
//
// 8 waves = two GROUPS of four (group = wave/4 owns output rows [128*group, +128), wave%4 owns 64 columns).  The
// hardware places wave w and wave w+4 of a workgroup on the same SIMD, so the two groups are run ONE SLOT OUT OF
// PHASE: while a group-0 wave issues its 16 MFMAs of a quadrant (M slot) the group-1 wave on the same SIMD reads its
// next fragments from LDS and issues DMA (L slot), and vice versa; every slot ends in one s_barrier.  The matrix pipe
// of each SIMD therefore always has one wave in an M slot (cdna_hip_programming.md T3/T4/T5, built here on a
// two-group stagger instead of the 8-phase template).
//
// A k-tile (64 deep) is four 16 KB LDS PIECES -- A0/A1 = the two 64-row halves of both groups' A rows, B0/B1 = the
// two 32-column halves of all four waves' W rows -- consumed in the quadrant order (A0,B0) (A0,B1) (A1,B1) (A1,B0);
// the B0 fragments stay in registers for the 4th quadrant.  Two stages (128 KB LDS).  In the L slot of quadrant p of
// tile t every wave issues its 2 DMA instructions of piece p of tile t+1, so every piece has 6-8 slots (~2000 cycles)
// of flight; completion is enforced with a counted `s_waitcnt vmcnt(4)` (never 0 in steady state) in front of the
// barrier that precedes the first read of that piece.  DMA goes through inline asm (see attention.hip: hipcc would
// otherwise drain it in front of every ds_read).
// =========================================================================================================
__device__ __forceinline__ void glds16_asm(const void* gsrc, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst_uniform)
                 : "memory");
}

#define PP_BARRIER()                         \
    do {                                     \
        __builtin_amdgcn_sched_barrier(0);   \
        __builtin_amdgcn_s_barrier();        \
        __builtin_amdgcn_sched_barrier(0);   \
    } while (0)
#define PP_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define PP_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

template <int ABL>   // ABL != 0: timing-only ablations (wrong results): 1 = no DMA after the prologue, 2 = also no vmcnt waits
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(const GemmParams p) {
    constexpr int BM = 256, BN = 256;
    constexpr int PIECE = 128 * 128;            // 128 rows x 64 bf16
    constexpr int STAGE = 4 * PIECE;            // [A0 | B0 | B1 | A1]
    constexpr int OFF_A0 = 0, OFF_B0 = PIECE, OFF_B1 = 2 * PIECE, OFF_A1 = 3 * PIECE;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wj = wave & 3;
    const bool g1 = grp != 0;

    // ---- XCD-aware tile mapping (same as gemm_tn_kernel) ----
    const int nblk = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int GM = p.gm;   // band height in M tiles (4 by default; BAGEL_GEMM_GM is a tuning knob)
    const int band = bid / (GM * p.tiles_n);
    const int band_rows = min(GM, p.tiles_m - band * GM);
    const int inb = bid - band * GM * p.tiles_n;
    const int tm = band * GM + inb % band_rows;
    const int tn = inb / band_rows;

    const int gi = (p.ngroups > 1 && tm >= p.g[1].tile0) ? 1 : 0;
    const bf16_t* __restrict__ Wg = p.g[gi].W;
    const bf16_t* __restrict__ biasg = p.g[gi].bias;
    const int* __restrict__ a_rows = p.g[gi].a_rows;
    const int* __restrict__ c_rows = p.g[gi].c_rows;
    const int Mg = p.g[gi].M;
    const int m0 = (tm - p.g[gi].tile0) * BM;
    const int n0 = tn * BN;

    // ---- DMA sources: piece-local row lr = 8*j + lane/8 (j = wave + 8*i), LDS chunk lane%8, global chunk swizzled ----
    const char* src[4][2];   // [piece][i]
    unsigned dst[4][2];      // LDS byte offset inside a stage (wave-uniform)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int j = wave + 8 * i;
        const int lr = j * 8 + (lane >> 3);
        const int gch = (lane & 7) ^ ((lr >> 1) & 7);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            // A piece `half`: tile row = (lr>>6)*128 + half*64 + (lr&63)
            int m = m0 + (lr >> 6) * 128 + half * 64 + (lr & 63);
            m = m < Mg ? m : Mg - 1;
            const long prow = a_rows ? (long)a_rows[m] : (long)m;
            src[half ? 3 : 0][i] = (const char*)(p.A + prow * p.lda) + gch * 16;
            // B piece `half`: tile col = (lr>>5)*64 + half*32 + (lr&31)
            int n = n0 + (lr >> 5) * 64 + half * 32 + (lr & 31);
            n = n < p.N ? n : p.N - 1;
            src[half ? 2 : 1][i] = (const char*)(Wg + (long)n * p.ldw) + gch * 16;
        }
        dst[0][i] = OFF_A0 + j * 1024; dst[1][i] = OFF_B0 + j * 1024; dst[2][i] = OFF_B1 + j * 1024; dst[3][i] = OFF_A1 + j * 1024;
    }
    // retire hipcc's own loads (a_rows gathers) before any hand-counted DMA is in flight
#pragma unroll
    for (int pc = 0; pc < 4; ++pc)
#pragma unroll
        for (int i = 0; i < 2; ++i) asm volatile("" ::"v"(src[pc][i]));

    const unsigned smem_base = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)smem);

    // ---- fragment read offsets (mfma_f32_16x16x32_bf16: lane -> row lane&15, 8 k-elements at chunk lane/16 + 4*kh) ----
    // (the 32x32x16 shape was measured 15-20 % slower in this structure: profiles/r01_gemm_experiments.md)
    const int fr = lane & 15;
    const int sw = fr >> 1;
    const int ch0 = ((lane >> 4) ^ sw) << 4;          // kh = 0
    const int ch1 = (((lane >> 4) + 4) ^ sw) << 4;    // kh = 1
    const int a_row = (grp * 64 + fr) * 128;          // + i*2048 (16 rows)
    const int b_row = (wj * 32 + fr) * 128;           // + jn*2048

    f32x4_t acc[2][4][2][2];   // [ma][i][nb][jn]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][i][b][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int nk = p.K >> 6;

    bool dma_on = true;
    auto issue = [&](int piece, unsigned stage_base, long koff) {
        if (ABL != 0 && !dma_on) return;
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16_asm(src[piece][i] + koff, stage_base + dst[piece][i]);
    };
    bf16x8_t af[4][2], b0f[2][2], b1f[2][2];
    auto read_a = [&](const char* sb, int off) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i][0] = *(const bf16x8_t*)(sb + off + a_row + i * 2048 + ch0);
            af[i][1] = *(const bf16x8_t*)(sb + off + a_row + i * 2048 + ch1);
        }
    };
    auto read_b = [&](bf16x8_t (&bf)[2][2], const char* sb, int off) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bf[j][0] = *(const bf16x8_t*)(sb + off + b_row + j * 2048 + ch0);
            bf[j][1] = *(const bf16x8_t*)(sb + off + b_row + j * 2048 + ch1);
        }
    };
    auto mma = [&](int ma, bf16x8_t (&bf0)[2][2], bf16x8_t (&bf1)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[ma][i][0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf0[j][kh], af[i][kh], acc[ma][i][0][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[ma][i][1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf1[j][kh], af[i][kh], acc[ma][i][1][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- schedule ---------------------------------------------------------------------------------------------------
    // Two phases per k-tile, each an L slot (LDS reads + DMA issue) followed by an M slot of 32 MFMAs (~530 cycles):
    //     L(t,0): reads A0,B0,B1(t)  issues A1(t+1)            M(t,0): A0 x (B0,B1)
    //     L(t,1): reads A1(t)        issues A0,B0,B1(t+2)      M(t,1): A1 x (B0,B1)
    // (16-MFMA slots -- four phases per tile -- measured ~1.40 PFLOP/s in the steady state; halving the number of
    // barriers per tile amortises the ~120-cycle slot turn-around.)  The two LDS stages act as a prefetch ring: a
    // piece's region is refilled one slot after BOTH groups have read it, so every piece has ~6 slots (~3.5k cycles)
    // of flight.  Counted waits (vmcnt counts this wave's DMA instructions, 2 per piece, oldest first), placed in front
    // of the barrier that precedes group 0's L slot (end of the M slot for group 0, end of the L slot for group 1):
    //     before L(t,0): A0,B0,B1(t) landed; newer in flight: A1(t), A0,B0,B1(t+1)      -> vmcnt(8)
    //     before L(t,1): A1(t) landed;       newer in flight: A0,B0,B1(t+1), A1(t+1)    -> vmcnt(8)
    issue(0, smem_base, 0); issue(1, smem_base, 0); issue(2, smem_base, 0); issue(3, smem_base, 0);
    if (nk > 1) {
        issue(0, smem_base + STAGE, 128); issue(1, smem_base + STAGE, 128); issue(2, smem_base + STAGE, 128);
        PP_VMCNT(8);
    } else {
        PP_VMCNT(2);
    }
    PP_BARRIER();
    if (g1) PP_BARRIER();   // group 1 runs one slot behind
    if (ABL != 0) { dma_on = false; PP_VMCNT(0); }

    for (int t = 0; t < nk; ++t) {
        const bool more1 = t + 1 < nk, more2 = t + 2 < nk;
        const char* sb = smem + (t & 1) * STAGE;
        const unsigned s_same = smem_base + (t & 1) * STAGE;          // tile t+2 reuses tile t's stage
        const unsigned s_other = smem_base + ((t + 1) & 1) * STAGE;
        // ---------------- phase 0: A0 x (B0, B1) ----------------
        if (more1) issue(3, s_other, (long)(t + 1) * 128);            // A1(t+1): its region was last read in L(t-1,1)
        read_a(sb, OFF_A0);
        read_b(b0f, sb, OFF_B0);
        read_b(b1f, sb, OFF_B1);
        PP_LGKM0();
        if (g1) { if (more1) PP_VMCNT(8); else PP_VMCNT(0); }         // A1(t) landed
        PP_BARRIER();
        mma(0, b0f, b1f);
        if (!g1) { if (more1) PP_VMCNT(8); else PP_VMCNT(0); }
        PP_BARRIER();
        // ---------------- phase 1: A1 x (B0, B1) ----------------
        if (more2) { issue(0, s_same, (long)(t + 2) * 128); issue(1, s_same, (long)(t + 2) * 128); issue(2, s_same, (long)(t + 2) * 128); }
        read_a(sb, OFF_A1);
        PP_LGKM0();
        if (g1 && more1) { if (more2) PP_VMCNT(8); else PP_VMCNT(2); }   // A0,B0,B1(t+1) landed
        PP_BARRIER();
        mma(1, b0f, b1f);
        if (!g1 && more1) { if (more2) PP_VMCNT(8); else PP_VMCNT(2); }
        PP_BARRIER();
    }
    if (!g1) PP_BARRIER();

    // ---- epilogue -------------------------------------------------------------------------------------------------
    // 1) in registers (fragment layout: lane owns 4 consecutive columns of one row): bias, activation, SwiGLU pairing,
    //    rounding to bf16 -- exactly the reference's cast points;
    // 2) the bf16 tile goes through LDS (the k-loop stages are dead now) so that
    // 3) every wave stores whole rows: 16 bytes per lane, 256/512 contiguous bytes per row, residual rows read the same
    //    way -- 16 full-line store instructions per wave instead of 64 scattered 8-byte ones.
    // LDS image: row-major [256][OWB] bf16, 8-byte chunk c8 of row r stored at chunk c8 ^ (r & 15) (conflict-free
    // ds_write_b64 for the 16 rows of a fragment, conflict-free ds_read_b128 on the way out).
    const bool swiglu = p.epi == EPI_SWIGLU16;
    const int OWB = swiglu ? 128 : 256;            // output columns of the block tile
    const int RS = OWB * 2;                        // row stride in bytes
    PP_BARRIER();                                  // every wave is past its last fragment read
    const int nsub = (lane >> 4) * 4;
#pragma unroll
    for (int ma = 0; ma < 2; ++ma)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = grp * 128 + ma * 64 + i * 16 + fr;       // row inside the block tile
            char* rowp = smem + r * RS;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                if (swiglu) {
                    const int c = wj * 32 + nb * 16 + nsub;         // output column inside the block tile
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float g = bfround(acc[ma][i][nb][0][e]);
                        const float u = bfround(acc[ma][i][nb][1][e]);
                        o[e] = bfround(silu_f(g)) * u;
                    }
                    u32x2_t v = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
                    *(u32x2_t*)(rowp + (((c >> 2) ^ (r & 15)) << 3)) = v;
                } else {
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn) {
                        const int c = wj * 64 + nb * 32 + jn * 16 + nsub;
                        const int n = n0 + c;
                        float o[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = acc[ma][i][nb][jn][e];
                        if (biasg && n < p.N) {
                            const u32x2_t bv = *(const u32x2_t*)(biasg + n);
                            o[0] += lo2f(bv[0]); o[1] += hi2f(bv[0]); o[2] += lo2f(bv[1]); o[3] += hi2f(bv[1]);
                        }
                        if (p.epi == EPI_GELU_TANH) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = gelu_tanh_f(bfround(o[e]));
                        } else if (p.epi == EPI_SILU) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = silu_f(bfround(o[e]));
                        }
                        u32x2_t v = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
                        *(u32x2_t*)(rowp + (((c >> 2) ^ (r & 15)) << 3)) = v;
                    }
                }
            }
        }
    PP_LGKM0();
    PP_BARRIER();
    {
        const int lpr = RS >> 4;                    // lanes per row (16-byte units): 32 or 16
        const int rpi = 64 / lpr;                   // rows per wave-instruction: 2 or 4
        const int q = lane % lpr;                   // 16-byte unit inside the row
        const int nrow_it = 256 / (8 * rpi);        // iterations per wave: 16 or 8
        const int ocol0 = swiglu ? (n0 >> 1) : n0;  // first output column of the block tile
        const int ncols = swiglu ? (p.N >> 1) : p.N;
        for (int it = 0; it < nrow_it; ++it) {
            const int r = (it * 8 + wave) * rpi + lane / lpr;
            const int m = m0 + r;
            const int oc = ocol0 + q * 8;
            if (m >= Mg || oc >= ncols) continue;
            const int x = r & 15;
            const int P = (q & ~7) | ((q & 7) ^ (x >> 1));
            u32x4_t v = *(const u32x4_t*)(smem + r * RS + (P << 4));
            if (x & 1) { v = (u32x4_t){v[2], v[3], v[0], v[1]}; }
            const long prow = c_rows ? (long)c_rows[m] : (long)m;
            if (p.R) {
                const u32x4_t rv = *(const u32x4_t*)(p.R + prow * p.ldr + oc);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = pack2bf(lo2f(v[e]) + lo2f(rv[e]), hi2f(v[e]) + hi2f(rv[e]));
            }
            *(u32x4_t*)(p.C + prow * p.ldc + oc) = v;
        }
    }
}

template <int ABL>
static int launch_gemm_pp(const GemmParams& p0, hipStream_t stream) {
    GemmParams p = p0;
    int t = 0;
    for (int g = 0; g < p.ngroups; ++g) {
        p.g[g].tile0 = t;
        t += ceil_div(p.g[g].M, 256);
    }
    p.tiles_m = t;
    p.tiles_n = ceil_div(p.N, 256);
    if (t == 0) return BAGEL_OK;
    static int gm = 0;
    if (gm == 0) {
        const char* e = getenv("BAGEL_GEMM_GM");
        gm = (e && atoi(e) > 0) ? atoi(e) : 4;
    }
    p.gm = gm;
    constexpr int smem = 2 * 4 * 128 * 128;
    if (int rc = bagel_enable_lds((const void*)gemm_pp_kernel<ABL>, smem, "gemm_pp_kernel")) return rc;
    hipLaunchKernelGGL(gemm_pp_kernel<ABL>, dim3(p.tiles_m * p.tiles_n), dim3(512), smem, stream, p);
    return bagel_check_launch("gemm_pp_kernel");
}

// =========================================================================================================
// Persistent form of the ping-pong kernel (variant 4).
//
// One workgroup per CU (the LDS allows no second one) walks the tile list with stride gridDim.x, so the per-tile costs
// that a fresh workgroup pays in the open -- dispatch, the HBM latency of the first k-tile, the store tail -- run
// underneath neighbouring work instead:
//   * when the k-loop of tile X ends, the DMA of k-tile 0 of tile X+1 is issued into LDS stage 0 BEFORE the epilogue of
//     tile X starts; the epilogue stages the bf16 tile through stage 1 only (two 128-row passes for a plain epilogue,
//     one pass for SwiGLU whose tile is half as wide), then k-tile 1 goes into stage 1 and the k-loop restarts with
//     exactly the in-flight DMA sequence its counted waits expect;
//   * hipcc does not see the asm DMA, so ANY wait it inserts for a load of its own drains the DMA pipe as well.  The
//     MoT row lists (A gather rows, C scatter rows) therefore never pass through a hipcc-counted load: the 256 + 256
//     indices of tile X+2 are fetched at the end of tile X by a 4-byte LDS-DMA (one instruction per wave) into a
//     three-deep table ring behind the two k-tile stages and read back with ds_read.  What is left for hipcc are the
//     residual rows (issued before k-tile 0, so waiting for them costs their own latency only) and the bias;
//   * the epilogue's stores are never waited for individually: the restart wait `vmcnt(6)` (only the six newest DMA
//     instructions may be in flight) covers k-tile 0 and every store, whatever order loads and stores retire in.
// Tile order: work item w keeps its XCD (gridDim.x is a multiple of 8), same band walk as variant 3.
// =========================================================================================================
// SGPR-base form of the LDS-DMA: address = wave-uniform 64-bit base + the lane's unsigned 32-bit byte offset (gemm_pq_kernel<.., SADDR>).
#ifndef BAGEL_PQ_M0MODE
#define BAGEL_PQ_M0MODE 0              /* experiment: how the SGPR-base DMA handles M0 (see issue() in gemm_pq_kernel) */
#endif
#ifndef BAGEL_PQ_ABL
#define BAGEL_PQ_ABL 0                 /* timing-only ablations of the SwiGLU epilogue (tools/ab_build.sh): 1 no exp/rcp, 2 no global stores */
#endif
__device__ __forceinline__ const char* pq_uniform(const char* ptr) {          // makes wave-uniformity provable to hipcc (an "s" operand)
    const unsigned long v = (unsigned long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(((unsigned long)hi << 32) | lo);
}
__device__ __forceinline__ void glds16_saddr(unsigned voff, const void* sbase_uniform, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase_uniform), "s"(lds_dst_uniform)
                 : "memory");
}
__device__ __forceinline__ void glds4_asm(const void* gsrc, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst_uniform)
                 : "memory");
}

// MODE fixes the epilogue at compile time (0 SwiGLU pairing, 1 residual add, 2 bias, 3 plain, 4 fp32 partials of a K part, 5 bias + residual, 6 bias + GELU-tanh:
// the SigLIP out / fc2 and fc1 projections, siglip_navit.py:216-258) so that no load of hipcc's sits
// behind a run-time branch: after such a join its wait counting turns conservative and drains the DMA pipe.
// FP8 = true: the operands are OCP e4m3 bytes with per-row fp32 scales (A: per activation row, W: per output column).  The LDS
// image, the DMA and the whole pipeline are byte-for-byte those of the bf16 kernel -- a 128-byte LDS row is 128 fp8 values instead of
// 64 bf16 (the host passes K, lda, ldw in 2-byte units) -- and one v_mfma_scale_f32_16x16x128_f8f6f4 (block scales fixed at 2^0)
// replaces the two 16x16x32 bf16 MFMAs of a fragment pair at the same pipe time: twice the arithmetic per k-tile, per LDS byte and
// per DMA byte.  The epilogue multiplies every accumulator by sa[row] * sw[col] before the bf16 roundings of the bf16 kernel.
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
__device__ __forceinline__ i32x8_t cat_frag(const bf16x8_t& lo, const bf16x8_t& hi) {
    const i32x4_t a = __builtin_bit_cast(i32x4_t, lo), b = __builtin_bit_cast(i32x4_t, hi);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// max over the 16 lanes of a DPP row (all 16 end up with it): quads, half-row mirror, row mirror
__device__ __forceinline__ float pq_rowmax16(float x) {
    x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, true)));     // quad_perm [1, 0, 3, 2]
    x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, true)));     // quad_perm [2, 3, 0, 1]
    x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xf, 0xf, true)));    // row_half_mirror
    x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x140, 0xf, 0xf, true)));    // row_mirror
    return x;
}

template <int MODE, bool FP8 = false, bool SADDR = false, bool QOUT = false>
__global__ __launch_bounds__(512, 2) void gemm_pq_kernel(const GemmParams p) {
    static_assert(!QOUT || (MODE == 0 && FP8), "the fp8 output exists for the SwiGLU epilogue of the fp8 kernel");
    constexpr bool SWIGLU = MODE == 0, HAS_R = MODE == 1 || MODE == 5, HAS_BIAS = MODE == 2 || MODE == 5 || MODE == 6, PARTIAL = MODE == 4, GELU = MODE == 6;
    constexpr int BM = 256, BN = 256;
    constexpr int PIECE = 128 * 128;
    constexpr int STAGE = 4 * PIECE;
    constexpr int OFF_A0 = 0, OFF_B0 = PIECE, OFF_B1 = 2 * PIECE, OFF_A1 = 3 * PIECE;
    constexpr int TBL = 2 * STAGE;                 // row tables: [3 buffers][A rows | C rows][256] int
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wj = wave & 3;
    const bool g1 = grp != 0;
    const int nblk = p.tiles_m * p.tiles_n;
    const int nk_all = p.K >> 6;
    const int nitems = p.w_count * p.ksplit;       // work items of this launch: item j = K part (j % ksplit) of tile w_begin + j / ksplit
    const int stride = gridDim.x;
    const unsigned smem_base = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)smem);
    // k-tile range [kt0, kt0 + nkt) of work item j
    auto item_k = [&](int j, int& kt0, int& nkt) {
        if (!PARTIAL) { kt0 = 0; nkt = nk_all; return; }
        const int part = j % p.ksplit;
        kt0 = (int)((long)nk_all * part / p.ksplit);
        nkt = (int)((long)nk_all * (part + 1) / p.ksplit) - kt0;
    };

    auto tile_of = [&](int j, int& tm, int& tn) {
        const int w = p.w_begin + (PARTIAL ? j / p.ksplit : j);
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = w & 7, loc = w >> 3;
        const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        const int GM = p.gm;
        const int band = bid / (GM * p.tiles_n);
        const int band_rows = min(GM, p.tiles_m - band * GM);
        const int inb = bid - band * GM * p.tiles_n;
        tm = band * GM + inb % band_rows;
        tn = inb / band_rows;
    };
    auto group_of = [&](int tm) { return (p.ngroups > 1 && tm >= p.g[1].tile0) ? 1 : 0; };
    // Row tables of tile row tm into ring slot `buf`: waves 0-3 fetch the 256 physical A rows, waves 4-7 the 256 physical
    // C/R rows (logical rows past the group's end are clamped; their results are never stored).
    auto fetch_tables = [&](int tm, int buf) {
        const int gi = group_of(tm);
        const int Mg = p.g[gi].M;
        const int m0 = (tm - p.g[gi].tile0) * BM;
        const int* __restrict__ lst = g1 ? p.g[gi].c_rows : p.g[gi].a_rows;
        int m = m0 + wj * 64 + lane;
        m = m < Mg ? m : Mg - 1;
        const unsigned off = TBL + buf * 2048 + grp * 1024 + wj * 256;
        if (lst) glds4_asm(lst + m, smem_base + off);
        else *(int*)(smem + off + lane * 4) = m;
    };
    // SADDR (variant 5): every DMA address = wave-uniform operand base (SGPR pair, advanced along K on the scalar unit) + the lane's
    // unsigned 32-bit byte offset -- ONE address VGPR per instruction instead of a pair.  The kernel is LDS-DMA-ISSUE bound (per wave and
    // k-tile 8 DMA + 24 ds_read instructions have to fit under two 32-MFMA slots of the partner group): round 4 measured gate+up at
    // M = 32 768 6.83 -> 6.31 ms (1 304 -> 1 411 TFLOP/s) from this alone.  Contract: every row the launch touches lies within 4 GiB of its
    // operand's base pointer (the host checks what it can see; gathered rows are the caller's promise: bagel_gemm_bf16 variant 5).
    unsigned soff[4][2];           // SADDR: [piece][i] byte offset from the operand base (A: pieces 0 / 3, W: pieces 1 / 2)
    const char* baseA = nullptr;   // SADDR: wave-uniform bases of the current tile (operand base + the item's first k-tile)
    const char* baseW = nullptr;
    const char* src[4][2];         // !SADDR: [piece][i] per-lane pointers
    // DMA sources of tile (tm, tn): piece-local row lr = 8*j + lane/8 (j = wave + 8*i), LDS chunk lane%8, global chunk
    // swizzled; A rows come from the table in ring slot `buf`.  `ln` is an opaque copy of the lane id (keeps hipcc from
    // hoisting this arithmetic out of the tile loop and spilling it across the k-loop).
    auto make_src = [&](int tm, int tn, int buf, int ln, int kt0) {
        const bf16_t* __restrict__ Wg = p.g[group_of(tm)].W;
        const long kb = (long)kt0 * 128;           // byte offset of the item's first k-tile inside a row
        const int n0 = tn * BN;
        const int* atab = (const int*)(smem + TBL + buf * 2048);
        if constexpr (SADDR) {
            baseA = pq_uniform((const char*)p.A + kb);
            baseW = pq_uniform((const char*)Wg + kb);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int lr = (wave + 8 * i) * 8 + (ln >> 3);
            const int gch16 = ((ln & 7) ^ ((lr >> 1) & 7)) * 16;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int row = atab[(lr >> 6) * 128 + half * 64 + (lr & 63)];
                int n = n0 + (lr >> 5) * 64 + half * 32 + (lr & 31);
                n = n < p.N ? n : p.N - 1;
                if constexpr (SADDR) {
                    soff[half ? 3 : 0][i] = (unsigned)row * (unsigned)(p.lda * 2) + gch16;
                    soff[half ? 2 : 1][i] = (unsigned)n * (unsigned)(p.ldw * 2) + gch16;
                } else {
                    src[half ? 3 : 0][i] = (const char*)(p.A + (long)row * p.lda) + gch16 + kb;
                    src[half ? 2 : 1][i] = (const char*)(Wg + (long)n * p.ldw) + gch16 + kb;
                }
            }
        }
    };
    auto pin_src = [&]() {
#pragma unroll
        for (int pc = 0; pc < 4; ++pc)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if constexpr (SADDR) asm volatile("" ::"v"(soff[pc][i]));
                else asm volatile("" ::"v"(src[pc][i]));
            }
    };
    auto issue = [&](int piece, unsigned stage_base, long koff) {
        const unsigned d0 = (piece == 0 ? OFF_A0 : piece == 1 ? OFF_B0 : piece == 2 ? OFF_B1 : OFF_A1) + wave * 1024;
        if constexpr (SADDR) {
            const char* sb = ((piece == 0 || piece == 3) ? baseA : baseW) + koff;
#if BAGEL_PQ_M0MODE == 0
            glds16_saddr(soff[piece][0], sb, stage_base + d0);
            glds16_saddr(soff[piece][1], sb, stage_base + d0 + 8192);
#elif BAGEL_PQ_M0MODE == 1          /* the two DMA instructions of a piece share one save / restore of M0 */
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(soff[piece][0]), "v"(soff[piece][1]), "s"(sb), "s"(stage_base + d0) : "memory", "scc");
#else                               /* M0 is not preserved at all (hipcc uses it nowhere in this kernel: checked on the ISA) */
            asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2\n\ts_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, %2"
                         :: "v"(soff[piece][0]), "v"(soff[piece][1]), "s"(sb), "s"(stage_base + d0) : "memory", "scc");
#endif
        } else {
            glds16_asm(src[piece][0] + koff, stage_base + d0);
            glds16_asm(src[piece][1] + koff, stage_base + d0 + 8192);
        }
    };

    const int fr = lane & 15;
    const int sw = fr >> 1;
    // bf16: fragment kh = chunks (lane/16) + 4 kh (k = 32 kh + 8 (lane/16) ..);  fp8: ONE fragment = the 32 bytes k = 32 (lane/16) ..,
    // i.e. chunks 2 (lane/16) and 2 (lane/16) + 1, kept in the same two registers quads
    const int ch0 = ((FP8 ? 2 * (lane >> 4) : (lane >> 4)) ^ sw) << 4;
    const int ch1 = ((FP8 ? 2 * (lane >> 4) + 1 : (lane >> 4) + 4) ^ sw) << 4;
    const int a_row = (grp * 64 + fr) * 128;
    const int b_row = (wj * 32 + fr) * 128;

    f32x4_t acc[2][4][2][2];   // [ma][i][nb][jn]
    bf16x8_t af[4][2], b0f[2][2], b1f[2][2];
    auto read_a = [&](const char* sb, int off) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i][0] = *(const bf16x8_t*)(sb + off + a_row + i * 2048 + ch0);
            af[i][1] = *(const bf16x8_t*)(sb + off + a_row + i * 2048 + ch1);
        }
    };
    auto read_b = [&](bf16x8_t (&bf)[2][2], const char* sb, int off) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bf[j][0] = *(const bf16x8_t*)(sb + off + b_row + j * 2048 + ch0);
            bf[j][1] = *(const bf16x8_t*)(sb + off + b_row + j * 2048 + ch1);
        }
    };
    auto mma = [&](int ma, bf16x8_t (&bf0)[2][2], bf16x8_t (&bf1)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
        if constexpr (FP8) {
            // 16 MFMAs of K = 128 (32 pipe cycles each) = the pipe time of the 32 bf16 MFMAs below, twice their arithmetic
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[ma][i][0][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(cat_frag(bf0[j][0], bf0[j][1]), cat_frag(af[i][0], af[i][1]),
                                                                                        acc[ma][i][0][j], 0, 0, 0, 127, 0, 127);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[ma][i][1][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(cat_frag(bf1[j][0], bf1[j][1]), cat_frag(af[i][0], af[i][1]),
                                                                                        acc[ma][i][1][j], 0, 0, 0, 127, 0, 127);
        } else
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[ma][i][0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf0[j][kh], af[i][kh], acc[ma][i][0][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[ma][i][1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf1[j][kh], af[i][kh], acc[ma][i][1][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- first tile: row tables of tiles 0 and 1, then the standard two-k-tile prologue ----
    int w = blockIdx.x, tm, tn;
    tile_of(w, tm, tn);
    int kt0, nk;
    item_k(w, kt0, nk);
    int wn = w + stride, tmn = 0, tnn = 0;
    fetch_tables(tm, 0);
    if (wn < nitems) { tile_of(wn, tmn, tnn); fetch_tables(tmn, 1); }
    PP_VMCNT(0);
    PP_LGKM0();
    PP_BARRIER();
    make_src(tm, tn, 0, lane, kt0);
    pin_src();
    issue(0, smem_base, 0); issue(1, smem_base, 0); issue(2, smem_base, 0); issue(3, smem_base, 0);
    issue(0, smem_base + STAGE, 128); issue(1, smem_base + STAGE, 128); issue(2, smem_base + STAGE, 128);
    PP_VMCNT(8);
    PP_BARRIER();
    if (g1) PP_BARRIER();   // group 1 runs one slot behind

    char* const stg = smem + STAGE;                // the epilogue stages through LDS stage 1 only
    int slot = 0;                                  // ring slot of the current tile's tables (tile counter mod 3)
    for (;;) {
        const int gi = group_of(tm);
        const bf16_t* __restrict__ biasg = p.g[gi].bias;
        const int Mg = p.g[gi].M;
        const int m0 = (tm - p.g[gi].tile0) * BM;
        const int n0 = tn * BN;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[a][i][b][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

        // ---- k-loop: identical to variant 3 (see the schedule comment there) ----
        for (int t = 0; t < nk; ++t) {
            const bool more1 = t + 1 < nk, more2 = t + 2 < nk;
            const char* sb = smem + (t & 1) * STAGE;
            const unsigned s_same = smem_base + (t & 1) * STAGE;
            const unsigned s_other = smem_base + ((t + 1) & 1) * STAGE;
            if (more1) issue(3, s_other, (long)(t + 1) * 128);
            read_a(sb, OFF_A0);
            read_b(b0f, sb, OFF_B0);
            read_b(b1f, sb, OFF_B1);
            PP_LGKM0();
            if (g1) { if (more1) PP_VMCNT(8); else PP_VMCNT(0); }
            PP_BARRIER();
            mma(0, b0f, b1f);
            if (!g1) { if (more1) PP_VMCNT(8); else PP_VMCNT(0); }
            PP_BARRIER();
            if (more2) { issue(0, s_same, (long)(t + 2) * 128); issue(1, s_same, (long)(t + 2) * 128); issue(2, s_same, (long)(t + 2) * 128); }
            read_a(sb, OFF_A1);
            PP_LGKM0();
            if (g1 && more1) { if (more2) PP_VMCNT(8); else PP_VMCNT(2); }
            PP_BARRIER();
            mma(1, b0f, b1f);
            if (!g1 && more1) { if (more2) PP_VMCNT(8); else PP_VMCNT(2); }
            PP_BARRIER();
        }
        if (!g1) PP_BARRIER();
        PP_BARRIER();                              // every wave is past its last fragment read: both stages are free

        // ---- between two k-loops.  Opaque copies of the lane / wave ids keep hipcc from hoisting this block's address
        //      arithmetic out of the tile loop (it spilled ~200 VGPRs across the k-loop doing so) ----
        int elane = lane, ewave = wave;
        asm volatile("" : "+v"(elane));
        asm volatile("" : "+s"(ewave));
        const int efr = elane & 15, ensub = (elane >> 4) * 4, egrp = ewave >> 2, ewj = ewave & 3;
        const int* ctab = (const int*)(smem + TBL + slot * 2048 + 1024);
        const bool has_next = wn < nitems;
        const int slot1 = slot == 2 ? 0 : slot + 1, slot2 = slot1 == 2 ? 0 : slot1 + 1;
        const int w2 = wn + stride;
        int tm2 = 0, tn2 = 0;
        int kt0n = 0, nkn = nk_all;
        if (has_next) item_k(wn, kt0n, nkn);

        // next tile: tables of the tile after it, then k-tile 0 into stage 0
        auto start_next = [&]() {
            if (has_next) {
                if (w2 < nitems) { tile_of(w2, tm2, tn2); fetch_tables(tm2, slot2); }
                make_src(tmn, tnn, slot1, elane, kt0n);
                pin_src();
                issue(0, smem_base, 0); issue(1, smem_base, 0); issue(2, smem_base, 0); issue(3, smem_base, 0);
            }
        };

        // ---- epilogue of the current tile, staged through LDS stage 1 (64 KB); only the stores are predicated ----
        if constexpr (PARTIAL) {
            // one K part of a tile: the raw fp32 accumulators, fragment by fragment, 16 bytes per lane (whole 1 KB lines per wave);
            // gemm_splitk_reduce_kernel sums the parts in this layout and applies the epilogue
            start_next();
            f32x4_t* dst = (f32x4_t*)p.part + (long)w * (32 * 512) + (ewave * 64 + elane);
#pragma unroll
            for (int ma = 0; ma < 2; ++ma)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                        for (int jn = 0; jn < 2; ++jn) dst[(((ma * 4 + i) * 2 + nb) * 2 + jn) * 512] = acc[ma][i][nb][jn];
        } else if constexpr (SWIGLU) {
            // LDS image: [256 rows][128 bf16], 8-byte chunk c8 of row r stored at c8 ^ (r & 15)
            const int qs = elane & 15;
            const int oc = (n0 >> 1) + qs * 8;
            const bool col_ok = oc < (p.N >> 1);
            int crow[8];                           // physical C rows of this lane's 8 store rows
            const int rb = ewave * 4 + (elane >> 4);          // store row of iteration `it` = it*32 + rb, rb < 32
#pragma unroll
            for (int it = 0; it < 8; ++it) crow[it] = ctab[it * 32 + rb];
            float sa_r[2][4];
            f32x4_t sw_c[2][2];
            if constexpr (FP8) {                   // de-quantisation scales: hipcc's own loads, retired before the next tile's DMA
                const int* atab = (const int*)(smem + TBL + slot * 2048);
#pragma unroll
                for (int ma = 0; ma < 2; ++ma)
#pragma unroll
                    for (int i = 0; i < 4; ++i) sa_r[ma][i] = p.sa[atab[egrp * 128 + ma * 64 + i * 16 + efr]];
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn) {
                        const int n = n0 + ewj * 64 + nb * 32 + jn * 16 + ensub;
                        sw_c[nb][jn] = *(const f32x4_t*)(p.sw + min(n, p.N - 4));
                    }
#pragma unroll
                for (int ma = 0; ma < 2; ++ma)
#pragma unroll
                    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(sa_r[ma][i]));
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn) asm volatile("" : "+v"(sw_c[nb][jn]));
            }
            float cinv[8];                         // QOUT: 1 / (the scale in use) of this lane's 8 store rows -- loaded with the other epilogue operands
            if constexpr (QOUT) {
#pragma unroll
                for (int it = 0; it < 8; ++it) cinv[it] = p.cs[crow[it]];
#pragma unroll
                for (int it = 0; it < 8; ++it) { asm volatile("" : "+v"(cinv[it])); cinv[it] = 1.0f / cinv[it]; }
            }
            start_next();
#pragma unroll
            for (int ma = 0; ma < 2; ++ma)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = egrp * 128 + ma * 64 + i * 16 + efr;
                    char* rowp = stg + r * 256;
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        const int c = ewj * 32 + nb * 16 + ensub;
                        float o[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float g = bfround(FP8 ? acc[ma][i][nb][0][e] * (sa_r[ma][i] * sw_c[nb][0][e]) : acc[ma][i][nb][0][e]);
                            const float u = bfround(FP8 ? acc[ma][i][nb][1][e] * (sa_r[ma][i] * sw_c[nb][1][e]) : acc[ma][i][nb][1][e]);
#if BAGEL_PQ_ABL & 1                 /* timing-only ablation (wrong results): no exp / rcp */
                            o[e] = g * u;
#else
                            o[e] = bfround(silu_f(g)) * u;
#endif
                        }
                        u32x2_t v = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
                        *(u32x2_t*)(rowp + (((c >> 2) ^ efr) << 3)) = v;
                    }
                }
            PP_LGKM0();
            PP_BARRIER();
            {
                const int x = rb & 15;             // = row & 15 for every iteration
                const int P = (qs & ~7) | ((qs & 7) ^ (x >> 1));
                const char* rd = stg + rb * 256 + (P << 4);
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    u32x4_t v = *(const u32x4_t*)(rd + it * 8192);
                    if (x & 1) { v = (u32x4_t){v[2], v[3], v[0], v[1]}; }
#if BAGEL_PQ_ABL & 2                 /* timing-only ablation (wrong results): no global stores; the staged tile is kept live */
                    asm volatile("" ::"v"(v));
#else
                    if constexpr (QOUT) {
                        // 8 bf16 of one row -> 8 e4m3 bytes with the caller's scale (clamped to the format's range: a row that grew past the delayed scale's
                        // headroom saturates instead of turning into NaN codes), and this row's max |value| for the next step's scale
                        const bool ok = col_ok && m0 + it * 32 + rb < Mg;
                        float f[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) { f[2 * e] = __uint_as_float(v[e] << 16); f[2 * e + 1] = __uint_as_float(v[e] & 0xffff0000u); }
                        float am = fmaxf(fmaxf(fabsf(f[0]), fabsf(f[1])), fmaxf(fabsf(f[2]), fabsf(f[3])));
                        am = fmaxf(am, fmaxf(fmaxf(fabsf(f[4]), fabsf(f[5])), fmaxf(fabsf(f[6]), fabsf(f[7]))));
                        am = pq_rowmax16(ok ? am : 0.f);                       // the 16 lanes of a DPP row hold the 128 columns of one tile row
                        const float inv = cinv[it];
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = fminf(fmaxf(f[e] * inv, -448.f), 448.f);
                        int w0 = 0, w1 = 0;
                        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w0, false);
                        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w0, true);
                        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], w1, false);
                        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], w1, true);
                        if (ok) *(u32x2_t*)(p.Cq + (long)crow[it] * p.ldcq + oc) = (u32x2_t){(unsigned)w0, (unsigned)w1};
                        if (qs == 0 && m0 + it * 32 + rb < Mg) atomicMax(p.cmax + crow[it], __float_as_uint(am));
                    } else {
                        if (col_ok && m0 + it * 32 + rb < Mg) *(u32x4_t*)(p.C + (long)crow[it] * p.ldc + oc) = v;
                    }
#endif
                }
            }
            PP_LGKM0();
        } else {
            // hipcc's own loads go out FIRST: bias (retired here), residual rows of pass 0 (waited for in the store-out, by
            // which time k-tile 0 -- issued right behind them -- costs nothing extra)
            const int qs = elane & 31;
            const int oc = n0 + qs * 8;
            const bool col_ok = oc < p.N;
            const int occ = col_ok ? oc : 0;
            u32x2_t bv[2][2];
            if constexpr (HAS_BIAS) {
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn) {
                        const int n = n0 + ewj * 64 + nb * 32 + jn * 16 + ensub;
                        bv[nb][jn] = *(const u32x2_t*)(biasg + min(n, p.N - 4));
                    }
            }
            // staging row of iteration `it` = it*16 + rbase (rbase < 16) -> tile row (it>>2)*128 + ma*64 + (it&3)*16 + rbase
            const int rbase = ewave * 2 + (elane >> 5);
            int crow[8];                           // physical C rows of this lane's 8 store rows of the pass
#pragma unroll
            for (int it = 0; it < 8; ++it) crow[it] = ctab[rbase + (it >> 2) * 128 + (it & 3) * 16];
            if constexpr (HAS_BIAS) {              // retired here, before the next tile's DMA is in flight
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn) asm volatile("" : "+v"(bv[nb][jn]));
            }
            float sa_r[2][4];
            f32x4_t sw_c[2][2];
            if constexpr (FP8) {                   // de-quantisation scales (see the SwiGLU branch)
                const int* atab = (const int*)(smem + TBL + slot * 2048);
#pragma unroll
                for (int ma = 0; ma < 2; ++ma)
#pragma unroll
                    for (int i = 0; i < 4; ++i) sa_r[ma][i] = p.sa[atab[egrp * 128 + ma * 64 + i * 16 + efr]];
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn) {
                        const int n = n0 + ewj * 64 + nb * 32 + jn * 16 + ensub;
                        sw_c[nb][jn] = *(const f32x4_t*)(p.sw + min(n, p.N - 4));
                    }
#pragma unroll
                for (int ma = 0; ma < 2; ++ma)
#pragma unroll
                    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(sa_r[ma][i]));
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn) asm volatile("" : "+v"(sw_c[nb][jn]));
            }
            u32x4_t rv[8];
            if constexpr (HAS_R) {
#pragma unroll
                for (int it = 0; it < 8; ++it) rv[it] = *(const u32x4_t*)(p.R + (long)crow[it] * p.ldr + occ);
            }
            start_next();
            // two passes of 128 rows (pass ma = rows [64*ma, 64*ma+64) of both groups); LDS image [128][256 bf16]
#pragma unroll
            for (int ma = 0; ma < 2; ++ma) {
                if (ma == 1) {
                    PP_BARRIER();                  // pass 0's reads of the staging area have retired
#pragma unroll
                    for (int it = 0; it < 8; ++it) crow[it] = ctab[rbase + (it >> 2) * 128 + 64 + (it & 3) * 16];
                    if constexpr (HAS_R) {         // residual rows of pass 1: in flight underneath the fragment phase
#pragma unroll
                        for (int it = 0; it < 8; ++it) rv[it] = *(const u32x4_t*)(p.R + (long)crow[it] * p.ldr + occ);
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int rr = egrp * 64 + i * 16 + efr;
                    char* rowp = stg + rr * 512;
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                        for (int jn = 0; jn < 2; ++jn) {
                            const int c = ewj * 64 + nb * 32 + jn * 16 + ensub;
                            float o[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = FP8 ? acc[ma][i][nb][jn][e] * (sa_r[ma][i] * sw_c[nb][jn][e]) : acc[ma][i][nb][jn][e];
                            if constexpr (HAS_BIAS) {
                                o[0] += lo2f(bv[nb][jn][0]); o[1] += hi2f(bv[nb][jn][0]);
                                o[2] += lo2f(bv[nb][jn][1]); o[3] += hi2f(bv[nb][jn][1]);
                            }
                            if constexpr (GELU) {      // the rounding points of the one-tile kernels: act(bf16(acc + bias))
#pragma unroll
                                for (int e = 0; e < 4; ++e) o[e] = gelu_tanh_f(bfround(o[e]));
                            }
                            u32x2_t v = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
                            *(u32x2_t*)(rowp + (((c >> 2) ^ efr) << 3)) = v;
                        }
                }
                PP_LGKM0();
                PP_BARRIER();
                const int P = (qs & ~7) | ((qs & 7) ^ (rbase >> 1));   // rbase = staging row & 15 for every iteration
                const char* rd = stg + rbase * 512 + (P << 4);
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int r = (it >> 2) * 128 + ma * 64 + (it & 3) * 16 + rbase;
                    u32x4_t v = *(const u32x4_t*)(rd + it * 8192);
                    if (rbase & 1) { v = (u32x4_t){v[2], v[3], v[0], v[1]}; }
                    if constexpr (HAS_R) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = pack2bf(lo2f(v[e]) + lo2f(rv[it][e]), hi2f(v[e]) + hi2f(rv[it][e]));
                    }
                    if (col_ok && m0 + r < Mg) *(u32x4_t*)(p.C + (long)crow[it] * p.ldc + oc) = v;
                }
                PP_LGKM0();
            }
        }
        if (!has_next) break;
        // ---- restart: k-tile 1 of the next tile into stage 1 once the staging area is drained ----
        PP_BARRIER();
        make_src(tmn, tnn, slot1, elane, kt0n);
        pin_src();
        issue(0, smem_base + STAGE, 128); issue(1, smem_base + STAGE, 128); issue(2, smem_base + STAGE, 128);
        PP_VMCNT(6);                               // k-tile 0 (issued before the epilogue), the tables and every store have retired
        PP_BARRIER();
        if (g1) PP_BARRIER();
        w = wn; tm = tmn; tn = tnn; nk = nkn;
        wn = w2; tmn = tm2; tnn = tn2;
        slot = slot1;
    }
}

// Second half of a K-split: out tile = epilogue(sum over the ksplit parts of a tile), parts in gemm_pq_kernel<4>'s fragment layout
// (part j of the launch at ((j * 32 + frag) * 512 + thread) float4).  EIGHT workgroups per tile (one per (ma, i) fragment row group: a
// few dozen leftover tiles must still cover the chip -- with one workgroup per tile this pass was slower than what the split saved),
// thread t = the GEMM's thread t, so the (row, column) of every accumulator is the kernel's own:  row = (wave/4)*128 + ma*64 + i*16 +
// lane%16,  column = (wave%4)*64 + nb*32 + jn*16 + (lane/16)*4 + e.  Rounding points as in the one-pass epilogue: bf16(acc + bias), then
// bf16(that + residual).
template <int MODE>
__global__ __launch_bounds__(512) void gemm_splitk_reduce_kernel(const GemmParams p) {
    constexpr bool HAS_R = MODE == 1 || MODE == 5, HAS_BIAS = MODE == 2 || MODE == 5 || MODE == 6, GELU = MODE == 6;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblk = p.tiles_m * p.tiles_n;
    const int t = blockIdx.x >> 3, mi = blockIdx.x & 7;           // leftover tile, fragment row group ma * 4 + i
    const int w = p.w_begin + t;
    int tm, tn;
    {
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = w & 7, loc = w >> 3;
        const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        const int GM = p.gm;
        const int band = bid / (GM * p.tiles_n);
        const int band_rows = min(GM, p.tiles_m - band * GM);
        const int inb = bid - band * GM * p.tiles_n;
        tm = band * GM + inb % band_rows;
        tn = inb / band_rows;
    }
    const int gi = (p.ngroups > 1 && tm >= p.g[1].tile0) ? 1 : 0;
    const int Mg = p.g[gi].M;
    const int m0 = (tm - p.g[gi].tile0) * 256, n0 = tn * 256;
    const int* __restrict__ crows = p.g[gi].c_rows;
    const bf16_t* __restrict__ bias = p.g[gi].bias;
    const int fr = lane & 15, nsub = (lane >> 4) * 4, grp = wave >> 2, wj = wave & 3;
    const int m = m0 + grp * 128 + (mi >> 2) * 64 + (mi & 3) * 16 + fr;
    if (m >= Mg) return;
    const long row = crows ? crows[m] : m;
    const f32x4_t* src = (const f32x4_t*)p.part + (long)t * p.ksplit * (32 * 512) + tid;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
            const int n = n0 + wj * 64 + nb * 32 + jn * 16 + nsub;
            if (n >= p.N) continue;
            const int frag = (mi * 2 + nb) * 2 + jn;
            f32x4_t a = src[(long)frag * 512];
            for (int k = 1; k < p.ksplit; ++k) a = a + src[((long)k * 32 + frag) * 512];
            float o[4] = {a[0], a[1], a[2], a[3]};
            if constexpr (HAS_BIAS) {
                const u32x2_t bv = *(const u32x2_t*)(bias + n);
                o[0] += lo2f(bv[0]); o[1] += hi2f(bv[0]); o[2] += lo2f(bv[1]); o[3] += hi2f(bv[1]);
            }
            if constexpr (GELU) {
                for (int e = 0; e < 4; ++e) o[e] = gelu_tanh_f(bfround(o[e]));
            }
            u32x2_t v = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
            if constexpr (HAS_R) {
                const u32x2_t rv = *(const u32x2_t*)(p.R + row * p.ldr + n);
                v[0] = pack2bf(lo2f(v[0]) + lo2f(rv[0]), hi2f(v[0]) + hi2f(rv[0]));
                v[1] = pack2bf(lo2f(v[1]) + lo2f(rv[1]), hi2f(v[1]) + hi2f(rv[1]));
            }
            *(u32x2_t*)(p.C + row * p.ldc + n) = v;
        }
}

// How a tile count that leaves the last round of the persistent kernel nearly empty is scheduled (LLM prefill of the understanding
// request: M = 4936 -> 20 row tiles; o / down have 280 tiles = 1.09 rounds of 256 CUs, qkv 360 = 1.41): the FULL rounds run as they
// are, the leftover tiles are cut along K into `ksplit` parts each so that they fill the chip once more with work items 1 / ksplit as
// long, and a small pass adds the parts up and applies the epilogue.  Returns the split (1 = do not split).
static int pq_leftover_split(int nblk, int wgs, int nk, size_t ws_bytes, int* n_full_out) {
    const int n_full = nblk / wgs * wgs, rest = nblk - n_full;
    *n_full_out = n_full;
    if (rest == 0) return 1;
    // A/B knobs, READ ONCE PER PROCESS (C++11 thread-safe statics: host threads launching GEMMs concurrently do not race on them; a change of the
    // environment after the first GEMM has no effect):
    static const int policy = [] { const char* e = getenv("BAGEL_GEMM_SPLIT_POLICY"); return e ? atoi(e) : 0; }();   // 0 = the round-3 rule (fill the chip ONCE with the leftovers), 1 = the round-5 cost model
    static const int force = [] { const char* f = getenv("BAGEL_GEMM_SPLIT_FORCE"); return f ? atoi(f) : 0; }();     // experiment: this many parts for every launch with leftovers
    if (force > 1) {
        int sf = force;
        if (sf > nk / 4) sf = nk / 4;
        while (sf > 1 && (size_t)rest * sf * (32 * 512 * 16) > ws_bytes) --sf;
        return sf < 2 ? 1 : sf;
    }
    if (policy == 0) {
        if (2 * rest > wgs + wgs / 4) return 1;                                // the last round is > 5/8 full anyway
        if (n_full == 0 && (2 * rest > wgs || nk < 16)) return 1;              // a single round: split only when it fills less than half the chip
        int s = wgs / rest;                                                    // parts per leftover tile: fill the chip once
        if (s > nk / 4) s = nk / 4;                                            // a part keeps >= 4 k-tiles (prologue + steady state)
        while (s > 1 && (size_t)rest * s * (32 * 512 * 16) > ws_bytes) --s;     // 256 KB of fp32 per part
        return s < 2 ? 1 : s;
    }
    // Round 5: the parts of the leftover tiles may take SEVERAL passes over the chip (rest * s > wgs), priced with a small model calibrated on the bench's own
    // launches: a 64-deep k-tile step of a 256 x 256 tile takes 1.52 us (gate+up at M = 32 768: 8.9 TFLOP in 6.32 ms = 4 144 steps), an item pays ~6 us of
    // pipeline fill + epilogue, a part costs 512 KB of fp32 partial traffic (written here, read by the reduce pass: ~0.13 us when the chip shares it) and the
    // reduce pass one more launch.  Un-split, the leftovers cost one whole tile time with (wgs - rest) CUs idle.  Example: the 3-stream edit forward has
    // 672 o / down tiles = 2.63 rounds; o (K = 3 584) is left alone (a 3-way split costs 137 us against 90), down (K = 18 944) is cut three ways: two passes of
    // third-length items = 380 us against 455.
    const double t_step = 1.52, t_item = 6.0, t_part = 0.13, t_reduce = 6.0;
    const double whole = nk * t_step + t_item;
    if (n_full == 0 && nk < 16) return 1;
    int best = 1;
    double best_cost = whole * 0.85;                                           // a split has to win 15 % of the leftover round to be worth a second launch
    for (int sp = 2; sp <= nk / 4 && sp <= 16; ++sp) {
        if ((size_t)rest * sp * (32 * 512 * 16) > ws_bytes) break;
        const int passes = (rest * sp + wgs - 1) / wgs;
        const double cost = passes * ((double)nk / sp * t_step + t_item) + rest * sp * t_part + t_reduce;
        // several passes (rest * sp > wgs) have to win a quarter of the round, not 15 %: measured on the edit request, the modelled 16 % of the three-way split of
        // its `down` leftovers (160 tiles) came out as +1 % of the request (profiles/r05_gemm_split_policy.log)
        if (cost < (passes > 1 ? whole * 0.75 : best_cost)) { best_cost = cost; best = sp; }
    }
    return best;
}

template <int MODE, bool FP8 = false, bool SADDR = false, bool QOUT = false>
static int launch_gemm_pq(const GemmParams& p0, hipStream_t stream, void* ws = nullptr, size_t ws_bytes = 0) {
    GemmParams p = p0;
    int t = 0;
    for (int g = 0; g < p.ngroups; ++g) {
        p.g[g].tile0 = t;
        t += ceil_div(p.g[g].M, 256);
    }
    p.tiles_m = t;
    p.tiles_n = ceil_div(p.N, 256);
    if (t == 0) return BAGEL_OK;
    static int gm = 0, wgs_of_dev[16] = {0};   // read-once tuning knobs; the CU count is cached per device
    if (gm == 0) {
        const char* e = getenv("BAGEL_GEMM_GM");
        gm = (e && atoi(e) > 0) ? atoi(e) : 4;
    }
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 15;
    if (wgs_of_dev[dev] == 0) {
        int cus = 0;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const char* w = getenv("BAGEL_GEMM_PERSIST_WGS");
        int v = (w && atoi(w) > 0) ? atoi(w) : cus;
        wgs_of_dev[dev] = v < 8 ? 8 : (v & ~7);    // a multiple of 8 keeps every work item of a workgroup on its XCD
    }
    const int wgs = wgs_of_dev[dev];
    p.gm = gm;
    constexpr int smem = 2 * 4 * 128 * 128 + 3 * 2048;   // two k-tile stages + the three-deep row-table ring
    if (int rc = bagel_enable_lds((const void*)gemm_pq_kernel<MODE, FP8, SADDR, QOUT>, smem, "gemm_pq_kernel")) return rc;
    const int nblk = p.tiles_m * p.tiles_n;
    p.w_begin = 0; p.w_count = nblk; p.ksplit = 1; p.part = nullptr;
    if constexpr (!FP8 && MODE != 0) {
        int n_full = 0;
        const int s = ws ? pq_leftover_split(nblk, wgs, p.K >> 6, ws_bytes, &n_full) : 1;
        if (s > 1) {
            // (1) the full rounds, one pass; (2) the leftover tiles as K parts -> fp32 partials; (3) sum + epilogue
            if (n_full > 0) {
                p.w_count = n_full;
                hipLaunchKernelGGL((gemm_pq_kernel<MODE, FP8, SADDR>), dim3(wgs), dim3(512), smem, stream, p);
                if (int rc = bagel_check_launch("gemm_pq_kernel")) return rc;
            }
            if (int rc = bagel_enable_lds((const void*)gemm_pq_kernel<4, false, SADDR>, smem, "gemm_pq_kernel<4>")) return rc;
            p.w_begin = n_full; p.w_count = nblk - n_full; p.ksplit = s; p.part = (float*)ws;
            const int items = p.w_count * s;
            hipLaunchKernelGGL((gemm_pq_kernel<4, false, SADDR>), dim3(items < wgs ? items : wgs), dim3(512), smem, stream, p);
            if (int rc = bagel_check_launch("gemm_pq_kernel<4>")) return rc;
            hipLaunchKernelGGL((gemm_splitk_reduce_kernel<MODE>), dim3(8 * p.w_count), dim3(512), 0, stream, p);
            return bagel_check_launch("gemm_splitk_reduce_kernel");
        }
    }
    hipLaunchKernelGGL((gemm_pq_kernel<MODE, FP8, SADDR, QOUT>), dim3(nblk < wgs ? nblk : wgs), dim3(512), smem, stream, p);
    return bagel_check_launch("gemm_pq_kernel");
}

// Round-2 experiments that did NOT beat the persistent ping-pong kernel and were removed (tools/gemm_v5_check.py history):
//   variant 5: one wave per SIMD (4 waves, 128 x 128 per wave, 256 accumulators in AGPRs through inline-asm MFMAs), 4-stage ring of
//              32-deep k-stages, one barrier per stage: 1 248 TFLOP/s at M 32768 / K 18944 (variant 4: 1 355-1 389), 1 698 with the
//              in-loop LDS-DMA switched off -- a single wave pays the 60-180 cycle issue cost of every global_load_lds itself;
//   variant 6: the same ring with two free-running waves per SIMD (8 waves, 64 x 128 each): 1 232; 1 350 with the DMA re-reading
//              L2-hot data, 1 616 without DMA.  So ~16 % of this structure goes to DMA issue / LDS write traffic and ~9 % to L2 misses;
//              the ping-pong kernels hide the DMA issue in the partner group's MFMA slot, which is why they stay.
template <int BM, int BN, int WM, int WN>
static int launch_gemm(const GemmParams& p0, hipStream_t stream) {
    GemmParams p = p0;
    int t = 0;
    for (int g = 0; g < p.ngroups; ++g) {
        p.g[g].tile0 = t;
        t += ceil_div(p.g[g].M, BM);
    }
    p.tiles_m = t;
    p.tiles_n = ceil_div(p.N, BN);
    if (t == 0) return BAGEL_OK;
    constexpr int smem = 2 * (BM + BN) * 128;
    if (int rc = bagel_enable_lds((const void*)gemm_tn_kernel<BM, BN, WM, WN>, smem, "gemm_tn_kernel")) return rc;
    hipLaunchKernelGGL((gemm_tn_kernel<BM, BN, WM, WN>), dim3(p.tiles_m * p.tiles_n), dim3(WM * WN * 64), smem, stream, p);
    return bagel_check_launch("gemm_tn_kernel");
}

static int gemm_bf16_impl(const void* A, int64_t lda,
                          const void* W0, const void* bias0, const int32_t* a_rows0, const int32_t* c_rows0, int32_t M0,
                          const void* W1, const void* bias1, const int32_t* a_rows1, const int32_t* c_rows1, int32_t M1,
                          int64_t ldw, const void* R, int64_t ldr, void* C, int64_t ldc,
                          int32_t N, int32_t K, int32_t epilogue, int32_t variant, void* ws, size_t ws_bytes, hipStream_t stream);

extern "C" int bagel_gemm_bf16(const void* A, int64_t lda,
                               const void* W0, const void* bias0, const int32_t* a_rows0, const int32_t* c_rows0, int32_t M0,
                               const void* W1, const void* bias1, const int32_t* a_rows1, const int32_t* c_rows1, int32_t M1,
                               int64_t ldw, const void* R, int64_t ldr, void* C, int64_t ldc,
                               int32_t N, int32_t K, int32_t epilogue, int32_t variant, hipStream_t stream) {
    return gemm_bf16_impl(A, lda, W0, bias0, a_rows0, c_rows0, M0, W1, bias1, a_rows1, c_rows1, M1, ldw, R, ldr, C, ldc, N, K, epilogue, variant,
                          nullptr, 0, stream);
}

// bagel_gemm_bf16 with a caller-owned fp32 workspace (>= 16-byte aligned): variant 4 may then cut the tiles of a nearly empty last round
// along K (see pq_leftover_split) -- same result up to the fp32 summation order of those tiles.  The workspace is only touched by this call.
extern "C" int bagel_gemm_bf16_ws(const void* A, int64_t lda,
                                  const void* W0, const void* bias0, const int32_t* a_rows0, const int32_t* c_rows0, int32_t M0,
                                  const void* W1, const void* bias1, const int32_t* a_rows1, const int32_t* c_rows1, int32_t M1,
                                  int64_t ldw, const void* R, int64_t ldr, void* C, int64_t ldc,
                                  int32_t N, int32_t K, int32_t epilogue, int32_t variant, void* workspace, int64_t workspace_bytes,
                                  hipStream_t stream) {
    BAGEL_REQUIRE(!workspace || ((((uintptr_t)workspace) & 15) == 0 && workspace_bytes >= 0), "gemm_ws: the workspace must be 16-byte aligned");
    return gemm_bf16_impl(A, lda, W0, bias0, a_rows0, c_rows0, M0, W1, bias1, a_rows1, c_rows1, M1, ldw, R, ldr, C, ldc, N, K, epilogue, variant,
                          workspace, (size_t)workspace_bytes, stream);
}

static int gemm_bf16_impl(const void* A, int64_t lda,
                          const void* W0, const void* bias0, const int32_t* a_rows0, const int32_t* c_rows0, int32_t M0,
                          const void* W1, const void* bias1, const int32_t* a_rows1, const int32_t* c_rows1, int32_t M1,
                          int64_t ldw, const void* R, int64_t ldr, void* C, int64_t ldc,
                          int32_t N, int32_t K, int32_t epilogue, int32_t variant, void* ws, size_t ws_bytes, hipStream_t stream) {
    BAGEL_REQUIRE(A && C && W0, "gemm: null pointer");
    BAGEL_REQUIRE(K > 0 && (K % 8) == 0, "gemm: K=%d must be a positive multiple of 8 (pad the operand)", K);
    BAGEL_REQUIRE(N > 0 && (N % 8) == 0, "gemm: N=%d must be a multiple of 8", N);
    BAGEL_REQUIRE((lda % 8) == 0 && (ldw % 8) == 0 && (ldc % 4) == 0 && (ldr % 4) == 0, "gemm: leading dims must keep rows 16-byte aligned");
    if ((variant == 3 || variant == 4) && ((ldc % 8) != 0 || (ldr % 8) != 0 || (((uintptr_t)C | (uintptr_t)R) & 15) != 0 || (epilogue == EPI_SWIGLU16 && (N % 16) != 0)))
        variant = 1;   // the ping-pong kernel stores 16-byte row chunks
    BAGEL_REQUIRE(epilogue >= 0 && epilogue <= 3, "gemm: unknown epilogue %d", epilogue);
    BAGEL_REQUIRE(epilogue != EPI_SWIGLU16 || ((N % 32) == 0 && !bias0 && !R), "gemm: swiglu needs N%%32==0, no bias/residual");
    BAGEL_REQUIRE(M0 >= 0 && M1 >= 0 && (M1 == 0 || W1), "gemm: bad group sizes");
    GemmParams p;
    p.A = (const bf16_t*)A; p.R = (const bf16_t*)R; p.C = (bf16_t*)C;
    p.lda = lda; p.ldw = ldw; p.ldr = ldr; p.ldc = ldc;
    p.N = N; p.K = K; p.epi = epilogue;
    p.ngroups = 0;
    if (M0 > 0) { p.g[p.ngroups] = GemmGroup{(const bf16_t*)W0, (const bf16_t*)bias0, a_rows0, c_rows0, M0, 0}; ++p.ngroups; }
    if (M1 > 0) { p.g[p.ngroups] = GemmGroup{(const bf16_t*)W1, (const bf16_t*)bias1, a_rows1, c_rows1, M1, 0}; ++p.ngroups; }
    if (p.ngroups == 0) return BAGEL_OK;
    if (p.ngroups == 1) p.g[1] = p.g[0];
    switch (variant) {
        case 0: return launch_gemm<128, 128, 2, 2>(p, stream);
        case 1: return launch_gemm<256, 256, 2, 4>(p, stream);
        case 2: return launch_gemm<256, 128, 2, 2>(p, stream);
        case 3:
            if ((K & 63) != 0) return launch_gemm<256, 256, 2, 4>(p, stream);   // the ping-pong kernel has no K tail
            return launch_gemm_pp<0>(p, stream);
        case 4:     // persistent ping-pong; epilogue combinations it does not instantiate go to variant 3
        case 5: {   // the same with SGPR-base DMA addresses: the CALLER promises that every A / W row of the launch lies within 4 GiB of the
                    // operand's base pointer (dense rows are checked here; gathered rows cannot be seen from the host side of the ABI)
            if ((K & 63) != 0) return launch_gemm<256, 256, 2, 4>(p, stream);
            const bool has_bias = p.g[0].bias != nullptr || p.g[1].bias != nullptr;
            const bool all_bias = p.g[0].bias != nullptr && p.g[1].bias != nullptr;
            bool saddr = variant == 5 && (uint64_t)N * (uint64_t)ldw * 2 < (1ull << 32);
            for (int g = 0; g < p.ngroups && saddr; ++g)
                if (!p.g[g].a_rows && (uint64_t)p.g[g].M * (uint64_t)lda * 2 >= (1ull << 32)) saddr = false;
            // round 6: the ViT's epilogues (bias + residual: out / fc2; bias + GELU-tanh: fc1) on the persistent kernel, SGPR-base form only
            const bool vit_epi = saddr && all_bias && K >= 128 && ((epilogue == EPI_NONE && R) || (epilogue == EPI_GELU_TANH && !R));
            if (!vit_epi && (K < 128 || epilogue == EPI_GELU_TANH || epilogue == EPI_SILU || (has_bias && (R || !all_bias)))) return launch_gemm_pp<0>(p, stream);
            if (vit_epi) {
                if (epilogue == EPI_GELU_TANH) return launch_gemm_pq<6, false, true>(p, stream, ws, ws_bytes);
                return launch_gemm_pq<5, false, true>(p, stream, ws, ws_bytes);
            }
            if (saddr) {
                if (epilogue == EPI_SWIGLU16) return launch_gemm_pq<0, false, true>(p, stream);
                if (R) return launch_gemm_pq<1, false, true>(p, stream, ws, ws_bytes);
                if (has_bias) return launch_gemm_pq<2, false, true>(p, stream, ws, ws_bytes);
                return launch_gemm_pq<3, false, true>(p, stream, ws, ws_bytes);
            }
            if (epilogue == EPI_SWIGLU16) return launch_gemm_pq<0>(p, stream);
            if (R) return launch_gemm_pq<1>(p, stream, ws, ws_bytes);
            if (has_bias) return launch_gemm_pq<2>(p, stream, ws, ws_bytes);
            return launch_gemm_pq<3>(p, stream, ws, ws_bytes);
        }
#ifdef BAGEL_ENABLE_ABLATIONS
        case 13: return launch_gemm_pp<1>(p, stream);   // timing-only ablation (results are garbage): no DMA in the k-loop
#endif
        default: return bagel_set_error(BAGEL_ERR_ARG, "gemm: unknown variant %d", variant);
    }
}

// FP8 (OCP e4m3) operands with per-row fp32 scales: C[c_rows[i], :] = epilogue((sa[a_rows[i]] * sw[n]) * sum_k Aq[a_rows[i], k] Wq[n, k]).
// One row group (the MoT gen expert: the und marker rows take the bf16 side path).  The same persistent kernel, DMA and LDS image as
// the bf16 path; K % 128 == 0 (one fp8 k-tile = 128 bytes), leading dimensions in BYTES and even.
extern "C" int bagel_gemm_fp8_bf16(const void* Aq, int64_t lda_bytes, const float* sa, const void* Wq, int64_t ldw_bytes, const float* sw,
                                   const void* bias, const int32_t* a_rows, const int32_t* c_rows, int32_t M, const void* R,
                                   int64_t ldr, void* C, int64_t ldc, int32_t N, int32_t K, int32_t epilogue, hipStream_t stream) {
    BAGEL_REQUIRE(Aq && Wq && sa && sw && C, "gemm_fp8: null pointer");
    BAGEL_REQUIRE(K >= 256 && (K % 128) == 0, "gemm_fp8: K=%d must be a multiple of 128 (>= 256)", K);
    BAGEL_REQUIRE(N > 0 && (N % 8) == 0, "gemm_fp8: N=%d must be a multiple of 8", N);
    BAGEL_REQUIRE((lda_bytes % 16) == 0 && (ldw_bytes % 16) == 0 && (ldc % 8) == 0 && (ldr % 8) == 0 && (((uintptr_t)C | (uintptr_t)R) & 15) == 0,
                  "gemm_fp8: leading dims must keep rows 16-byte aligned");
    BAGEL_REQUIRE(epilogue == EPI_NONE || epilogue == EPI_SWIGLU16, "gemm_fp8: epilogue %d not built", epilogue);
    BAGEL_REQUIRE(epilogue != EPI_SWIGLU16 || ((N % 32) == 0 && !bias && !R), "gemm_fp8: swiglu needs N%%32==0, no bias/residual");
    BAGEL_REQUIRE(!(bias && R), "gemm_fp8: bias + residual not built");
    if (M <= 0) return BAGEL_OK;
    GemmParams p;
    p.A = (const bf16_t*)Aq; p.R = (const bf16_t*)R; p.C = (bf16_t*)C;
    p.lda = lda_bytes / 2; p.ldw = ldw_bytes / 2; p.ldr = ldr; p.ldc = ldc;      // the kernel addresses operands in 2-byte units
    p.N = N; p.K = K / 2; p.epi = epilogue;
    p.sa = sa; p.sw = sw;
    p.ngroups = 1;
    p.g[0] = GemmGroup{(const bf16_t*)Wq, (const bf16_t*)bias, a_rows, c_rows, M, 0};
    p.g[1] = p.g[0];
    // (the fp8 path keeps per-lane 64-bit DMA addresses: its gathered A rows carry no extent promise in this entry point)
    if (epilogue == EPI_SWIGLU16) return launch_gemm_pq<0, true>(p, stream);
    if (R) return launch_gemm_pq<1, true>(p, stream);
    if (bias) return launch_gemm_pq<2, true>(p, stream);
    return launch_gemm_pq<3, true>(p, stream);
}

// The gate/up projection of the FP8 gen expert with the SwiGLU result written as e4m3 bytes (no bf16 round trip, no stand-alone quantiser pass in front of the
// down projection): Cq[c_rows[i], :N/2] = e4m3(clamp(swiglu(...) / cs[c_rows[i]], +-448)), cmax[c_rows[i]] = max(cmax[..], max_n |swiglu(...)|) (fp32 bits,
// atomicMax).  DELAYED scaling: `cs` is chosen by the caller before the values exist (bagel_fp8_delayed_scales: from the previous denoise step's row maxima,
// with headroom), `cmax` collects this step's maxima for the next one.  oracle/fp8.py restates the scheme.
extern "C" int bagel_gemm_fp8_swiglu_q8(const void* Aq, int64_t lda_bytes, const float* sa, const void* Wq, int64_t ldw_bytes, const float* sw,
                                        const int32_t* a_rows, const int32_t* c_rows, int32_t M, void* Cq, int64_t ldcq_bytes, const float* cs,
                                        void* cmax, int32_t N, int32_t K, hipStream_t stream) {
    BAGEL_REQUIRE(Aq && Wq && sa && sw && Cq && cs && cmax, "gemm_fp8_swiglu_q8: null pointer");
    BAGEL_REQUIRE(K >= 256 && (K % 128) == 0, "gemm_fp8_swiglu_q8: K=%d must be a multiple of 128 (>= 256)", K);
    BAGEL_REQUIRE(N > 0 && (N % 32) == 0, "gemm_fp8_swiglu_q8: N=%d must be a multiple of 32", N);
    BAGEL_REQUIRE((lda_bytes % 16) == 0 && (ldw_bytes % 16) == 0 && (ldcq_bytes % 8) == 0 && (((uintptr_t)Cq) & 7) == 0,
                  "gemm_fp8_swiglu_q8: leading dims must keep rows aligned (operands 16 bytes, the fp8 output 8)");
    if (M <= 0) return BAGEL_OK;
    GemmParams p;
    p.A = (const bf16_t*)Aq; p.R = nullptr; p.C = nullptr;
    p.lda = lda_bytes / 2; p.ldw = ldw_bytes / 2; p.ldr = 0; p.ldc = 0;
    p.N = N; p.K = K / 2; p.epi = EPI_SWIGLU16;
    p.sa = sa; p.sw = sw;
    p.Cq = (unsigned char*)Cq; p.ldcq = ldcq_bytes; p.cs = cs; p.cmax = (unsigned*)cmax;
    p.ngroups = 1;
    p.g[0] = GemmGroup{(const bf16_t*)Wq, nullptr, a_rows, c_rows, M, 0};
    p.g[1] = p.g[0];
    return launch_gemm_pq<0, true, false, true>(p, stream);
}
