// Skinny MFMA GEMM: C[M <= 64, N] = A[M,K] W[N,K]^T (+bias)(act)(SwiGLU16)(+R) for 2..64 rows -- batched decode steps and
// short-prompt prefill (qwen2_navit.py:515-517,591; modeling_qwen2.py:200-201; bagel.py:978 at a few rows).
//
// At these sizes the product is still a weight stream (HBM-bound: W is read once, A lives in L2), but M rows of VALU
// dot products per weight byte no longer fit under the HBM time (the lane-FMA kernel of decode.hip is VALU-bound from
// M = 2), and a 128x128 MFMA tile leaves 7/8 of the CUs idle (28 workgroups for N = 3584).  So: no LDS tile at all.
// A wave owns ONE 16-column block of the output (16 weight rows; SwiGLU16: the 16 gate rows and their 16 up rows) and a
// K range; per 32-deep step a lane loads 16 bytes of its weight row straight from HBM (the A operand of
// mfma_f32_16x16x32_bf16, fragment = row lane%16, k chunk lane/16 -- exactly the row-major layout, no staging needed)
// and 16 bytes of activation row lane%16 from L2 (B operand); 8 steps = 8 KB of weights per wave are in flight at a
// time.  Swapped operands (D = Wfrag . Afrag) as in gemm.hip: a lane ends up with 4 consecutive output columns of one
// row, so the epilogue (same rounding points as gemm.hip) stores 8 bytes.  Few column blocks (N = 3584): the four waves
// of a workgroup split K and meet in LDS.
#include "common.h"
#include <stdlib.h>

#define EPI_NONE 0
#define EPI_GELU_TANH 1
#define EPI_SILU 2
#define EPI_SWIGLU16 3

struct SkinnyParams {
    const bf16_t* A; long lda;
    const bf16_t* W; long ldw;
    const bf16_t* bias;
    const bf16_t* R; long ldr;
    bf16_t* C; long ldc;
    int M, N, K, epi;
    int split;      // K splits per column block (1, 2, 4 or 8 waves of the workgroup share one block)
};

template <int MT, bool SWIGLU>
__global__ __launch_bounds__(512) void gemm_skinny_kernel(SkinnyParams p) {
    constexpr int KU = 8;                       // 32-deep steps per batch (8 x 16 B per lane and operand in flight)
    constexpr int NACC = SWIGLU ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char skinny_smem[];
    f32x4_t (*part)[NACC * MT][64] = (f32x4_t (*)[NACC * MT][64])skinny_smem;     // [wave][acc][lane], only when split > 1
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int ncb = SWIGLU ? p.N / 32 : p.N / 16;
    const int nk = p.K >> 5;                    // 32-deep steps
    // the workgroup's waves = (column blocks per workgroup) x (K splits): wave -> (block, split)
    const int S = p.split;
    const int nw = blockDim.x >> 6;
    const int cb = blockIdx.x * (nw / S) + wave / S;
    const int ks = wave % S;
    const bool live = cb < ncb;
    int k_lo = 0, k_hi = 0;
    if (live) {
        const int per = (nk + S - 1) / S;
        k_lo = ks * per;
        k_hi = (k_lo + per < nk) ? k_lo + per : nk;
        if (k_lo > k_hi) k_lo = k_hi;
    }
    const int cbc = live ? cb : ncb - 1;        // idle waves keep valid addresses (they run zero steps)
    const int wrow = SWIGLU ? cbc * 32 : cbc * 16;
    const bf16_t* wg = p.W + (long)(wrow + r) * p.ldw + q * 8;           // gate (or the only) weight row of this lane
    const bf16_t* wu = wg + (long)16 * p.ldw;                           // SwiGLU: the matching up row
    const bf16_t* xa[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int m = mt * 16 + r;
        m = m < p.M ? m : p.M - 1;              // rows >= M compute a duplicate that is never stored
        xa[mt] = p.A + (long)m * p.lda + q * 8;
    }
    f32x4_t acc[NACC][MT];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[a][mt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    for (int k0 = k_lo; k0 < k_hi; k0 += KU) {
        bf16x8_t wf[NACC][KU], xf[MT][KU];
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int k = (k0 + u < k_hi) ? k0 + u : k_hi - 1;          // tail: re-read the last step, masked below
#if BAGEL_NT_SKINNY
            wf[0][u] = __builtin_nontemporal_load((const bf16x8_t*)(wg + (long)k * 32));          // weights: read once, non-temporal
            if (SWIGLU) wf[NACC - 1][u] = __builtin_nontemporal_load((const bf16x8_t*)(wu + (long)k * 32));
#else
            wf[0][u] = *(const bf16x8_t*)(wg + (long)k * 32);
            if (SWIGLU) wf[NACC - 1][u] = *(const bf16x8_t*)(wu + (long)k * 32);
#endif
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xf[mt][u] = *(const bf16x8_t*)(xa[mt] + (long)k * 32);
        }
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            if (k0 + u < k_hi) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    acc[0][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][u], xf[mt][u], acc[0][mt], 0, 0, 0);
                    if (SWIGLU) acc[NACC - 1][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[NACC - 1][u], xf[mt][u], acc[NACC - 1][mt], 0, 0, 0);
                }
            }
        }
    }

    if (S > 1) {                                // the K splits of a column block meet in LDS; split 0 finishes
        if (ks > 0) {
#pragma unroll
            for (int a = 0; a < NACC; ++a)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) part[wave][a * MT + mt][lane] = acc[a][mt];
        }
        __syncthreads();
        if (ks > 0 || !live) return;
        for (int o = 1; o < S; ++o)
#pragma unroll
            for (int a = 0; a < NACC; ++a)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[a][mt] = acc[a][mt] + part[wave + o][a * MT + mt][lane];
    } else if (!live) {
        return;
    }

    // ---- epilogue: lane owns C[m = mt*16 + r][n .. n+3], n = cb*16 + 4*q (same rounding points as gemm.hip) ----
    const int n = cb * 16 + q * 4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 16 + r;
        if (m >= p.M) continue;
        float o[4];
        if (SWIGLU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float g = bfround(acc[0][mt][e]);
                const float u = bfround(acc[NACC - 1][mt][e]);
                o[e] = bfround(silu_f(g)) * u;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = acc[0][mt][e];
            if (p.bias) {
                const u32x2_t bv = *(const u32x2_t*)(p.bias + n);
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
                const u32x2_t rv = *(const u32x2_t*)(p.R + (long)m * p.ldr + n);
                o[0] = bfround(o[0]) + lo2f(rv[0]); o[1] = bfround(o[1]) + hi2f(rv[0]);
                o[2] = bfround(o[2]) + lo2f(rv[1]); o[3] = bfround(o[3]) + hi2f(rv[1]);
            }
        }
        u32x2_t v = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
        *(u32x2_t*)(p.C + (long)m * p.ldc + n) = v;
    }
}

template <int MT>
static int launch_skinny(const SkinnyParams& p, hipStream_t stream) {
    const bool sw = p.epi == EPI_SWIGLU16;
    const int ncb = sw ? p.N / 32 : p.N / 16;
    const int nk = p.K / 32;
    SkinnyParams q = p;
    // enough waves to keep ~8 KB x 3000 of weights in flight: split K over 2/4/8 waves when there are few column blocks
    // (N = 3584 -> 224 blocks x 8; gate+up -> 1184 x 4; lm_head -> 9504 x 1), never below 4 steps per split
    static int target = 0;
    if (target == 0) {
        const char* e = getenv("BAGEL_SKINNY_WAVES");
        target = (e && atoi(e) > 0) ? atoi(e) : 3000;
    }
    int S = 1;
    while (S < 8 && ncb * S < target && nk / (2 * S) >= 4) S *= 2;
    q.split = S;
    const int nw = S > 4 ? S : 4;                       // waves per workgroup
    const int grid = ceil_div(ncb, nw / S);
    const size_t smem = S > 1 ? (size_t)nw * (sw ? 2 : 1) * MT * 64 * sizeof(f32x4_t) : 0;
    if (sw) hipLaunchKernelGGL((gemm_skinny_kernel<MT, true>), dim3(grid), dim3(64 * nw), smem, stream, q);
    else hipLaunchKernelGGL((gemm_skinny_kernel<MT, false>), dim3(grid), dim3(64 * nw), smem, stream, q);
    return bagel_check_launch("gemm_skinny_kernel");
}

extern "C" int bagel_gemm_skinny_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias, const void* R,
                                      int64_t ldr, void* C, int64_t ldc, int32_t M, int32_t N, int32_t K, int32_t epilogue,
                                      hipStream_t stream) {
    BAGEL_REQUIRE(A && W && C, "gemm_skinny: null pointer");
    BAGEL_REQUIRE(M >= 1 && M <= 64, "gemm_skinny: M=%d not in [1,64]", M);
    BAGEL_REQUIRE(K > 0 && (K % 32) == 0 && (lda % 8) == 0 && (ldw % 8) == 0, "gemm_skinny: K %% 32 == 0 and 16-byte rows required");
    BAGEL_REQUIRE(epilogue >= 0 && epilogue <= 3, "gemm_skinny: unknown epilogue %d", epilogue);
    BAGEL_REQUIRE(epilogue == EPI_SWIGLU16 ? ((N % 32) == 0 && !bias && !R) : (N % 16) == 0, "gemm_skinny: N %% 16 (SwiGLU: N %% 32, no bias/residual)");
    BAGEL_REQUIRE((ldc % 4) == 0 && (ldr % 4) == 0, "gemm_skinny: ldc/ldr must be multiples of 4");
    BAGEL_REQUIRE((((uintptr_t)A | (uintptr_t)W) & 15) == 0 && (((uintptr_t)C | (uintptr_t)R | (uintptr_t)bias) & 7) == 0, "gemm_skinny: alignment");
    SkinnyParams p;
    p.A = (const bf16_t*)A; p.lda = lda; p.W = (const bf16_t*)W; p.ldw = ldw; p.bias = (const bf16_t*)bias;
    p.R = (const bf16_t*)R; p.ldr = ldr; p.C = (bf16_t*)C; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.epi = epilogue; p.split = 1;
    const int mt = (M + 15) / 16;
    switch (mt) {
        case 1: return launch_skinny<1>(p, stream);
        case 2: return launch_skinny<2>(p, stream);
        case 3: return launch_skinny<3>(p, stream);
        default: return launch_skinny<4>(p, stream);
    }
}
