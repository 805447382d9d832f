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
    int epi;
    int ngroups;
    GemmGroup g[2];
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
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<BM, BN, WM, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_tn_kernel<BM, BN, WM, WN>), dim3(p.tiles_m * p.tiles_n), dim3(WM * WN * 64), smem, stream, p);
    return bagel_check_launch("gemm_tn_kernel");
}

extern "C" int bagel_gemm_bf16(const void* A, int64_t lda,
                               const void* W0, const void* bias0, const int32_t* a_rows0, const int32_t* c_rows0, int32_t M0,
                               const void* W1, const void* bias1, const int32_t* a_rows1, const int32_t* c_rows1, int32_t M1,
                               int64_t ldw, const void* R, int64_t ldr, void* C, int64_t ldc,
                               int32_t N, int32_t K, int32_t epilogue, int32_t variant, hipStream_t stream) {
    BAGEL_REQUIRE(A && C && W0, "gemm: null pointer");
    BAGEL_REQUIRE(K > 0 && (K % 8) == 0, "gemm: K=%d must be a positive multiple of 8 (pad the operand)", K);
    BAGEL_REQUIRE(N > 0 && (N % 8) == 0, "gemm: N=%d must be a multiple of 8", N);
    BAGEL_REQUIRE((lda % 8) == 0 && (ldw % 8) == 0 && (ldc % 4) == 0 && (ldr % 4) == 0, "gemm: leading dims must keep rows 16-byte aligned");
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
        default: return bagel_set_error(BAGEL_ERR_ARG, "gemm: unknown variant %d", variant);
    }
}
