// Autoregressive text decode (Bagel.generate_text, bagel.py:930-1000; Qwen2 forward at Lq = 1): HBM-bound kernels.
//
// One decoded token streams every und-expert weight once (14.14 GB at 7B) plus the KV context; nothing here is
// GEMM-shaped, so nothing goes to the MFMA.  The design rules are the HBM ones: 16-byte lanes, whole 1 KB wave
// requests on contiguous weight rows, many loads in flight per lane, no re-reads, and all per-step state (token,
// position, KV length, step counter) in device memory so a hipGraph of one step can be replayed without the host.
//
//   bagel_gemv_bf16              C[M<=4.., N] = A W^T (+bias)(act)(SwiGLU)(+R) with an optional fused Qwen2RMSNorm of
//                                the A rows (modeling_qwen2.py:54-59, 200-201; qwen2_navit.py:515-517,591; bagel.py:978)
//   bagel_kv_append_paged_bf16   this step's K/V rows -> page slot kv_len[b] (qwen2_navit.py:563-575 without the
//                                whole-cache re-scatter)
//   bagel_attn_decode_paged_bf16 Lq = 1 attention over a paged KV cache, split over the keys (flash-decoding), GQA
//                                group shares every K/V byte (qwen2_navit.py:579-588)
//   bagel_decode_advance         token bookkeeping of bagel.py:984-994 on the device
#include "common.h"
#include <stdlib.h>

#ifndef BAGEL_GEMV_U4
#define BAGEL_GEMV_U4 10
#endif
#define EPI_NONE 0
#define EPI_GELU_TANH 1
#define EPI_SILU 2
#define EPI_SWIGLU16 3

// =====================================================================================================================
// Skinny GEMM.  A wave owns a PAIR of weight rows at a time (for SwiGLU16 the gate row and its up row, 16 apart, so
// the product needs no second pass); every lane streams 16-byte chunks of both rows, U chunk-groups (= U KB per row)
// in flight, and multiplies them with the activation rows staged once per workgroup in LDS (bf16, conflict-free
// ds_read_b128).  fp32 accumulate, wave reduction, epilogue by lane 0 with the roundings of gemm.hip.
// =====================================================================================================================
struct GemvParams {
    const bf16_t* A; long lda;
    const bf16_t* W; long ldw;
    const bf16_t* bias;
    const bf16_t* R; long ldr;
    bf16_t* C; long ldc;
    const bf16_t* norm_w; float eps;
    int M, N, K, epi;
    int ppw;   // weight-row pairs per wave
};

template <int MR>
__device__ __forceinline__ void gemv_fma(float (&a0)[MR][2], float (&a1)[MR][2], const u32x4_t wa, const u32x4_t wb,
                                         const bf16_t* xs, int K, int chunk, bool ok) {
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        u32x4_t xv = *(const u32x4_t*)(xs + (long)m * K + (long)chunk * 8);
        if (!ok) xv = u32x4_t{0u, 0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xl = lo2f(xv[e]), xh = hi2f(xv[e]);
            a0[m][0] = fmaf(lo2f(wa[e]), xl, a0[m][0]);
            a0[m][1] = fmaf(hi2f(wa[e]), xh, a0[m][1]);
            a1[m][0] = fmaf(lo2f(wb[e]), xl, a1[m][0]);
            a1[m][1] = fmaf(hi2f(wb[e]), xh, a1[m][1]);
        }
    }
}

// same, activation chunk already in a register (split-K path: the lane that owns a weight chunk loads its own x chunk)
template <int MR>
__device__ __forceinline__ void gemv_fma_reg(float (&a0)[MR][2], float (&a1)[MR][2], const u32x4_t wa, const u32x4_t wb,
                                             const u32x4_t (&xv)[MR]) {
#pragma unroll
    for (int m = 0; m < MR; ++m) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xl = lo2f(xv[m][e]), xh = hi2f(xv[m][e]);
            a0[m][0] = fmaf(lo2f(wa[e]), xl, a0[m][0]);
            a0[m][1] = fmaf(hi2f(wa[e]), xh, a0[m][1]);
            a1[m][0] = fmaf(lo2f(wb[e]), xl, a1[m][0]);
            a1[m][1] = fmaf(hi2f(wb[e]), xh, a1[m][1]);
        }
    }
}

__device__ __forceinline__ u32x4_t ldw_nt(const bf16_t* ptr) { return ld_stream<u32x4_t>(ptr); }

template <int U>
__device__ __forceinline__ void gemv_load_batch(u32x4_t (&wa)[U], u32x4_t (&wb)[U], const bf16_t* w0, const bf16_t* w1, int g,
                                                int lane, int nch) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int ch = (g + u) * 64 + lane;
        const long off = (long)(ch < nch ? ch : 0) * 8;     // out-of-range chunks re-read chunk 0 and are multiplied by 0
        wa[u] = ldw_nt(w0 + off);
        wb[u] = ldw_nt(w1 + off);
    }
}

// whole batch in range: one base address per row, the U groups at constant strides (1 KB apart)
template <int U>
__device__ __forceinline__ void gemv_load_full(u32x4_t (&wa)[U], u32x4_t (&wb)[U], const bf16_t* w0, const bf16_t* w1, int g, int lane) {
    const bf16_t* a = w0 + ((long)g * 64 + lane) * 8;
    const bf16_t* b = w1 + ((long)g * 64 + lane) * 8;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        wa[u] = ldw_nt(a + u * 512);
        wb[u] = ldw_nt(b + u * 512);
    }
}

// One workgroup = 4 waves.  WPP = 1: every wave owns `ppw` consecutive row pairs and the whole K range.  WPP = 4 (long
// rows, few of them: the down projection): the four waves split the K range of ONE pair and their partial sums meet in
// LDS, so N/2 workgroups keep 4x more rows in flight than N/8 would.
//
// Issue order at the top is what the latency chain needs: (1) the activation chunks, the norm weights and the first
// pair's bias/residual, (2) the wave's first weight batch, (3) the reduction / scaling / LDS staging, which waits only
// for (1) (loads return in order: had the weights gone first, the staging would have waited for HBM), (4) the FMAs,
// by which time the weights have landed.
template <int MR, int WPP>
__device__ __forceinline__ void gemv_body(const GemvParams& p) {
    // chunk groups in flight per row: K = 3584 -> exactly one batch of 7; split-K (K = 18944 over 4 waves = 10, 10, 10, 7 groups): the
    // whole quarter row in ONE batch (with 7 + 3 every wave streamed in two bursts with its FMAs in between)
    constexpr int U = (WPP == 4) ? BAGEL_GEMV_U4 : 7;
    extern __shared__ __attribute__((aligned(16))) unsigned char gemv_smem[];
    bf16_t* xs = (bf16_t*)gemv_smem;   // [MR][K]
    __shared__ float red[MR][4];
    __shared__ float part[4][2][MR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = p.K, nch = K >> 3;
    const int NP = p.N >> 1;
    const bool swiglu = p.epi == EPI_SWIGLU16;
    const int ngr = (nch + 63) >> 6, nfull = nch >> 6;
    // this wave's pairs and K range (in 64-chunk groups)
    int pbase, pend, g_lo, g_hi;
    if (WPP == 1) {
        pbase = (blockIdx.x * 4 + wave) * p.ppw;
        pend = (pbase + p.ppw < NP) ? pbase + p.ppw : NP;
        g_lo = 0;
        g_hi = ngr;
    } else {
        pbase = blockIdx.x;
        pend = pbase + 1;                       // grid == NP
        const int gq = (ngr + 3) >> 2;
        g_lo = wave * gq;
        g_hi = (g_lo + gq < ngr) ? g_lo + gq : ngr;
        if (g_lo > g_hi) g_lo = g_hi;
    }
    const int full_hi = g_hi < nfull ? g_hi : nfull;     // groups below this bound are complete for every lane
    const int ch_hi = (g_hi * 64 < nch) ? g_hi * 64 : nch;
    const bool fin = (WPP == 1) ? (lane == 0) : (tid == 0);   // the lane that runs the epilogue

    // ---- (1) activation chunks (K <= 4096: kept in registers across the two norm passes), norm weights, epilogue operands
    const bool small = nch <= 512;
    u32x4_t xr[MR][2], gw[2];
    if (small) {
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const bf16_t* ar = p.A + (long)(m < p.M ? m : p.M - 1) * p.lda;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = tid + 256 * i;
                xr[m][i] = c < nch ? *(const u32x4_t*)(ar + (long)c * 8) : u32x4_t{0u, 0u, 0u, 0u};
            }
        }
        if (p.norm_w) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = tid + 256 * i;
                gw[i] = c < nch ? *(const u32x4_t*)(p.norm_w + (long)c * 8) : u32x4_t{0u, 0u, 0u, 0u};
            }
        }
    }
    float eb[2] = {0.f, 0.f}, er[MR][2];
#pragma unroll
    for (int m = 0; m < MR; ++m) er[m][0] = er[m][1] = 0.f;
    if (fin && pbase < pend && !swiglu) {
        const int r0 = 2 * pbase;
        if (p.bias) { eb[0] = bf2f(p.bias[r0]); eb[1] = bf2f(p.bias[r0 + 1]); }
        if (p.R) {
#pragma unroll
            for (int m = 0; m < MR; ++m)
                if (m < p.M) {
                    const unsigned rv = *(const unsigned*)(p.R + (long)m * p.ldr + r0);     // r0 is even: one aligned dword
                    er[m][0] = lo2f(rv);
                    er[m][1] = hi2f(rv);
                }
        }
    }

    // ---- (2) first weight batch of this wave
    u32x4_t wa[U], wb[U];
    const bool pre_full = g_lo + U <= full_hi;
    if (pbase < pend && g_lo < g_hi) {
        const int r0 = swiglu ? ((pbase >> 4) << 5) + (pbase & 15) : 2 * pbase;
        const int r1 = swiglu ? r0 + 16 : r0 + 1;
        if (pre_full) gemv_load_full<U>(wa, wb, p.W + (long)r0 * p.ldw, p.W + (long)r1 * p.ldw, g_lo, lane);
        else gemv_load_batch<U>(wa, wb, p.W + (long)r0 * p.ldw, p.W + (long)r1 * p.ldw, g_lo, lane, ch_hi);
    }

    // ---- (3) stage the activation rows in LDS (optionally RMS-normalised).  The split-K path has no staging at all:
    //      a lane needs exactly the x chunks at its own weight-chunk positions and reads them from L2 next to the weights
    //      (no norm there: the launcher only picks WPP = 4 for un-normalised inputs)
    if (WPP == 1) {
    if (p.norm_w) {
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            float ss = 0.f;
            if (small) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a = lo2f(xr[m][i][e]), b = hi2f(xr[m][i][e]);
                        ss += a * a + b * b;
                    }
            } else {
                const bf16_t* ar = p.A + (long)(m < p.M ? m : p.M - 1) * p.lda;
                for (int c = tid; c < nch; c += 256) {
                    const u32x4_t v = *(const u32x4_t*)(ar + (long)c * 8);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a = lo2f(v[e]), b = hi2f(v[e]);
                        ss += a * a + b * b;
                    }
                }
            }
            ss = wave_sum(ss);
            if (lane == 0) red[m][wave] = ss;
        }
        __syncthreads();
    }
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        const bf16_t* ar = p.A + (long)(m < p.M ? m : p.M - 1) * p.lda;
        float inv = 1.f;
        if (p.norm_w) inv = rsqrtf((red[m][0] + red[m][1] + red[m][2] + red[m][3]) / (float)K + p.eps);
        if (small) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = tid + 256 * i;
                if (c < nch) {
                    u32x4_t v = xr[m][i];
                    if (p.norm_w) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            v[e] = pack2bf(bfround(lo2f(v[e]) * inv) * lo2f(gw[i][e]), bfround(hi2f(v[e]) * inv) * hi2f(gw[i][e]));
                    }
                    *(u32x4_t*)(xs + (long)m * K + (long)c * 8) = v;
                }
            }
        } else {
            for (int c = tid; c < nch; c += 256) {
                u32x4_t v = *(const u32x4_t*)(ar + (long)c * 8);
                if (p.norm_w) {
                    const u32x4_t g = *(const u32x4_t*)(p.norm_w + (long)c * 8);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        v[e] = pack2bf(bfround(lo2f(v[e]) * inv) * lo2f(g[e]), bfround(hi2f(v[e]) * inv) * hi2f(g[e]));
                }
                *(u32x4_t*)(xs + (long)m * K + (long)c * 8) = v;
            }
        }
    }
    __syncthreads();
    }

    // ---- (4) weight-row pairs ----------------------------------------------------------------------------------------
    for (int pp = pbase; pp < pend; ++pp) {
        const int r0 = swiglu ? ((pp >> 4) << 5) + (pp & 15) : 2 * pp;
        const int r1 = swiglu ? r0 + 16 : r0 + 1;
        const bf16_t* w0 = p.W + (long)r0 * p.ldw;
        const bf16_t* w1 = p.W + (long)r1 * p.ldw;
        float a0[MR][2], a1[MR][2];
#pragma unroll
        for (int m = 0; m < MR; ++m) a0[m][0] = a0[m][1] = a1[m][0] = a1[m][1] = 0.f;
        int g = g_lo;
        for (; g + U <= full_hi; g += U) {                     // whole batches: no predicates
            if (pp != pbase || g != g_lo) gemv_load_full<U>(wa, wb, w0, w1, g, lane);
            if (WPP == 1) {
#pragma unroll
                for (int u = 0; u < U; ++u) gemv_fma<MR>(a0, a1, wa[u], wb[u], xs, K, (g + u) * 64 + lane, true);
            } else {
                u32x4_t xv[U][MR];
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int m = 0; m < MR; ++m)
                        xv[u][m] = *(const u32x4_t*)(p.A + (long)(m < p.M ? m : p.M - 1) * p.lda + ((long)(g + u) * 64 + lane) * 8);
#pragma unroll
                for (int u = 0; u < U; ++u) gemv_fma_reg<MR>(a0, a1, wa[u], wb[u], xv[u]);
            }
        }
        if (g < g_hi) {                                         // ragged tail batch (clamped loads, zeroed activations)
            if (pp != pbase || g != g_lo) gemv_load_batch<U>(wa, wb, w0, w1, g, lane, ch_hi);
            if (WPP == 1) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int ch = (g + u) * 64 + lane;
                    const bool ok = ch < ch_hi;
                    gemv_fma<MR>(a0, a1, wa[u], wb[u], xs, K, ok ? ch : 0, ok);
                }
            } else {
                u32x4_t xv[U][MR];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int ch = (g + u) * 64 + lane;
                    const bool ok = ch < ch_hi;
#pragma unroll
                    for (int m = 0; m < MR; ++m) {
                        xv[u][m] = *(const u32x4_t*)(p.A + (long)(m < p.M ? m : p.M - 1) * p.lda + (long)(ok ? ch : 0) * 8);
                        if (!ok) xv[u][m] = u32x4_t{0u, 0u, 0u, 0u};
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) gemv_fma_reg<MR>(a0, a1, wa[u], wb[u], xv[u]);
            }
        }
        float s0[MR], s1[MR];
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            s0[m] = wave_sum(a0[m][0] + a0[m][1]);
            s1[m] = wave_sum(a1[m][0] + a1[m][1]);
        }
        if (WPP > 1) {                                          // the four K quarters meet in LDS
            if (lane == 0) {
#pragma unroll
                for (int m = 0; m < MR; ++m) { part[wave][0][m] = s0[m]; part[wave][1][m] = s1[m]; }
            }
            __syncthreads();
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                s0[m] = (part[0][0][m] + part[1][0][m]) + (part[2][0][m] + part[3][0][m]);
                s1[m] = (part[0][1][m] + part[1][1][m]) + (part[2][1][m] + part[3][1][m]);
            }
        }
        if (fin) {
            if (pp != pbase && !swiglu) {                       // later pairs of this wave: operands were not prefetched
                if (p.bias) { eb[0] = bf2f(p.bias[r0]); eb[1] = bf2f(p.bias[r1]); }
                if (p.R) {
#pragma unroll
                    for (int m = 0; m < MR; ++m)
                        if (m < p.M) {
                            const unsigned rv = *(const unsigned*)(p.R + (long)m * p.ldr + r0);
                            er[m][0] = lo2f(rv);
                            er[m][1] = hi2f(rv);
                        }
                }
            }
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                if (m >= p.M) continue;
                if (swiglu) {
                    const float gg = bfround(s0[m]), uu = bfround(s1[m]);
                    p.C[(long)m * p.ldc + pp] = f2bf(bfround(silu_f(gg)) * uu);
                } else {
                    float o[2] = {s0[m], s1[m]};
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        if (p.bias) o[t] += eb[t];
                        if (p.epi == EPI_GELU_TANH) o[t] = gelu_tanh_f(bfround(o[t]));
                        else if (p.epi == EPI_SILU) o[t] = silu_f(bfround(o[t]));
                        if (p.R) o[t] = bfround(o[t]) + er[m][t];
                    }
                    *(unsigned*)(p.C + (long)m * p.ldc + r0) = pack2bf(o[0], o[1]);      // rows r0, r0+1: one aligned dword
                }
            }
        }
    }
}

template <int MR, int WPP>
__global__ __launch_bounds__(256) void gemv_kernel(GemvParams p) { gemv_body<MR, WPP>(p); }

#define GEMV_MAX_LDS (144 * 1024)

template <int MR, int WPP>
static int launch_gemv(const GemvParams& p, hipStream_t stream) {
    const size_t smem = WPP == 1 ? (size_t)MR * p.K * sizeof(bf16_t) : 0;     // the split-K path stages nothing
    if (smem > 48 * 1024)
        if (int rc = bagel_enable_lds((const void*)gemv_kernel<MR, WPP>, (int)GEMV_MAX_LDS, "gemv_kernel")) return rc;
    const int NP = p.N / 2;
    GemvParams q = p;
    int grid;
    if (WPP == 1) {
        // one row pair per wave while that still gives <= ~4096 workgroups (the hardware dispatcher balances the tail),
        // more pairs per wave beyond that so the per-workgroup activation staging stays a small share of the L2 traffic
        static int wg_target = 0;
        if (wg_target == 0) {
            const char* e = getenv("BAGEL_GEMV_WGS");     // tuning knob; the default is what bench.py measures
            wg_target = (e && atoi(e) > 0) ? atoi(e) : 4096;
        }
        int ppw = NP / (4 * wg_target);
        if (ppw < 1) ppw = 1;
        q.ppw = ppw;
        grid = ceil_div(NP, 4 * ppw);
    } else {
        q.ppw = 1;
        grid = NP;
    }
    hipLaunchKernelGGL((gemv_kernel<MR, WPP>), dim3(grid), dim3(256), smem, stream, q);
    return bagel_check_launch("gemv_kernel");
}

template <int MR>
static int launch_gemv_any(const GemvParams& p, hipStream_t stream) {
    // long rows and few of them (down projection: K = 18944, N = 3584): split K over the 4 waves of a workgroup
    static int split_k = -1;
    if (split_k < 0) {
        const char* e = getenv("BAGEL_GEMV_SPLITK");
        split_k = e ? atoi(e) : 1;
    }
    if (split_k && !p.norm_w && p.K >= 8192 && p.N / 2 <= 8192 && (p.N % 2) == 0) return launch_gemv<MR, 4>(p, stream);
    return launch_gemv<MR, 1>(p, stream);
}

extern "C" int bagel_gemv_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias, const void* R,
                               int64_t ldr, void* C, int64_t ldc, const void* norm_w, float eps, int32_t M, int32_t N,
                               int32_t K, int32_t epilogue, hipStream_t stream) {
    BAGEL_REQUIRE(A && W && C, "gemv: null pointer");
    BAGEL_REQUIRE(K > 0 && (K % 8) == 0 && (lda % 8) == 0 && (ldw % 8) == 0, "gemv: K/lda/ldw must be multiples of 8 (16-byte rows)");
    BAGEL_REQUIRE(N > 0 && (N % 2) == 0, "gemv: N=%d must be even", N);
    BAGEL_REQUIRE(epilogue >= 0 && epilogue <= 3, "gemv: unknown epilogue %d", epilogue);
    BAGEL_REQUIRE(epilogue != EPI_SWIGLU16 || ((N % 32) == 0 && !bias && !R), "gemv: swiglu needs N%%32==0, no bias/residual");
    BAGEL_REQUIRE((((uintptr_t)A | (uintptr_t)W | (uintptr_t)norm_w) & 15) == 0, "gemv: A/W/norm_w must be 16-byte aligned");
    BAGEL_REQUIRE((size_t)K * sizeof(bf16_t) <= GEMV_MAX_LDS, "gemv: K=%d does not fit the LDS staging buffer", K);
    BAGEL_REQUIRE((ldc % 2) == 0 && (ldr % 2) == 0 && (((uintptr_t)C | (uintptr_t)R) & 3) == 0, "gemv: C/R rows must be 4-byte aligned");
    if (M <= 0) return BAGEL_OK;
    int m0 = 0;
    while (m0 < M) {
        int mr = (M - m0 >= 4) ? 4 : (M - m0 >= 2 ? 2 : 1);
        while (mr > 1 && (size_t)mr * K * sizeof(bf16_t) > GEMV_MAX_LDS) mr >>= 1;
        GemvParams p;
        p.A = (const bf16_t*)A + (long)m0 * lda; p.lda = lda;
        p.W = (const bf16_t*)W; p.ldw = ldw;
        p.bias = (const bf16_t*)bias;
        p.R = R ? (const bf16_t*)R + (long)m0 * ldr : nullptr; p.ldr = ldr;
        p.C = (bf16_t*)C + (long)m0 * ldc; p.ldc = ldc;
        p.norm_w = (const bf16_t*)norm_w; p.eps = eps;
        p.M = mr; p.N = N; p.K = K; p.epi = epilogue; p.ppw = 1;
        int rc;
        if (mr == 4) rc = launch_gemv_any<4>(p, stream);
        else if (mr == 2) rc = launch_gemv_any<2>(p, stream);
        else rc = launch_gemv_any<1>(p, stream);
        if (rc != BAGEL_OK) return rc;
        m0 += mr;
    }
    return BAGEL_OK;
}

// =====================================================================================================================
// Paged KV cache.  Token j of sample b lives in row  block_table[b*bt_stride + j/PAGE] * PAGE + j%PAGE  of the layer's
// K pool and V pool ([pages*PAGE, nkv*DP] bf16).  kv_len[b] is device memory: one captured step serves every length.
// =====================================================================================================================
#define BAGEL_KV_PAGE 64

__global__ __launch_bounds__(256) void kv_append_paged_kernel(const bf16_t* __restrict__ k_new, const bf16_t* __restrict__ v_new,
                                                              long ld_new, bf16_t* __restrict__ kpool, bf16_t* __restrict__ vpool,
                                                              long ldp, const int* __restrict__ block_table, int bt_stride,
                                                              const int* __restrict__ kv_len, int width) {
    const int b = blockIdx.x, which = blockIdx.y;
    const int j = kv_len[b];
    const long row = (long)block_table[(long)b * bt_stride + j / BAGEL_KV_PAGE] * BAGEL_KV_PAGE + (j % BAGEL_KV_PAGE);
    const bf16_t* src = (which ? v_new : k_new) + (long)b * ld_new;
    bf16_t* dst = (which ? vpool : kpool) + row * ldp;
    for (int c = threadIdx.x; c < (width >> 3); c += 256) *(u32x4_t*)(dst + c * 8) = *(const u32x4_t*)(src + c * 8);
}

extern "C" int bagel_kv_append_paged_bf16(const void* k_new, const void* v_new, int64_t ld_new, void* kpool, void* vpool,
                                          int64_t ld_pool, const int32_t* block_table, int32_t bt_stride,
                                          const int32_t* kv_len, int32_t batch, int32_t width, hipStream_t stream) {
    BAGEL_REQUIRE(k_new && v_new && kpool && vpool && block_table && kv_len, "kv_append_paged: null pointer");
    BAGEL_REQUIRE(width > 0 && (width % 8) == 0 && (ld_new % 8) == 0 && (ld_pool % 8) == 0, "kv_append_paged: width/ld must be multiples of 8");
    BAGEL_REQUIRE((((uintptr_t)k_new | (uintptr_t)v_new | (uintptr_t)kpool | (uintptr_t)vpool) & 15) == 0, "kv_append_paged: 16-byte alignment");
    if (batch <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(kv_append_paged_kernel, dim3(batch, 2), dim3(256), 0, stream, (const bf16_t*)k_new, (const bf16_t*)v_new,
                       (long)ld_new, (bf16_t*)kpool, (bf16_t*)vpool, (long)ld_pool, block_table, bt_stride, kv_len, width);
    return bagel_check_launch("kv_append_paged_kernel");
}

// Decode-step epilogue of the fused QKV projection: q_norm/k_norm + RoPE (und cast points, identical arithmetic to
// qknorm_rope_kernel<EPL> in norm.hip: bf16(w * bf16(x * rsqrt)), bf16 products and bf16 sum) with q rewritten in place
// and the finished K row and the V row written straight into their page slot kv_len[b] -- one launch instead of
// qknorm_rope + kv_append, and every head has its own 16 lanes so nothing is serialised (the packed-prefill kernel
// walks the heads of a row in a loop, which is 12 us of pure latency at one row).
//   qkv row: [nq*DP | nkv*DP | nkv*DP]; HD = 16*EPL true head dim, pad lanes [HD, DP) stay zero in the pages.
template <int EPL>
__global__ __launch_bounds__(256) void decode_qkv_post_kernel(bf16_t* __restrict__ qkv, long ld, const bf16_t* __restrict__ cosb,
                                                              const bf16_t* __restrict__ sinb, const bf16_t* __restrict__ qw,
                                                              const bf16_t* __restrict__ kw, bf16_t* __restrict__ kpool,
                                                              bf16_t* __restrict__ vpool, long ldp,
                                                              const int* __restrict__ block_table, int bt_stride,
                                                              const int* __restrict__ kv_len, int nq, int nkv, int dp, float eps,
                                                              int use_norm) {
    constexpr int HD = 16 * EPL;
    const int b = blockIdx.y;
    const int sub = threadIdx.x & 15;
    const int h = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int nheads = nq + 2 * nkv;
    const bool act = h < nheads;
    const int hh = act ? h : nheads - 1;
    const bool is_q = hh < nq, is_v = hh >= nq + nkv;
    const int e0 = sub * EPL;
    const bool upper = sub >= 8;
    const int j = kv_len[b];
    const long prow = (long)block_table[(long)b * bt_stride + j / BAGEL_KV_PAGE] * BAGEL_KV_PAGE + (j % BAGEL_KV_PAGE);
    bf16_t* src = qkv + (long)b * ld + (long)hh * dp + e0;
    float cs[EPL], sn[EPL];
    {
        const int c0 = e0 & (HD / 2 - 1);
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            cs[e] = bf2f(cosb[(long)b * (HD / 2) + c0 + e]);
            sn[e] = bf2f(sinb[(long)b * (HD / 2) + c0 + e]);
        }
    }
    float x[EPL], wf[EPL];
    unsigned xr[EPL / 2];
#pragma unroll
    for (int e = 0; e < EPL / 2; ++e) xr[e] = *(const unsigned*)(src + 2 * e);
    {
        const bf16_t* wv = (is_q ? qw : kw) + e0;
#pragma unroll
        for (int e = 0; e < EPL / 2; ++e) {
            const unsigned wr = use_norm ? *(const unsigned*)(wv + 2 * e) : 0u;
            x[2 * e] = lo2f(xr[e]);
            x[2 * e + 1] = hi2f(xr[e]);
            wf[2 * e] = use_norm ? lo2f(wr) : 1.0f;
            wf[2 * e + 1] = use_norm ? hi2f(wr) : 1.0f;
        }
    }
    float nrm[EPL];
    if (use_norm) {
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) ss += x[e] * x[e];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        const float inv = rsqrtf(ss / (float)HD + eps);
#pragma unroll
        for (int e = 0; e < EPL; ++e) nrm[e] = bfround(wf[e] * bfround(x[e] * inv));
    } else {
#pragma unroll
        for (int e = 0; e < EPL; ++e) nrm[e] = x[e];
    }
    unsigned orr[EPL / 2];
    {
        float out[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const float partner = __shfl_xor(nrm[e], 8, 64);
            const float rot = upper ? partner : -partner;
            out[e] = bfround(nrm[e] * cs[e]) + bfround(rot * sn[e]);
        }
#pragma unroll
        for (int e = 0; e < EPL / 2; ++e) orr[e] = is_v ? xr[e] : pack2bf(out[2 * e], out[2 * e + 1]);
    }
    if (!act) return;
    bf16_t* dst;
    if (is_q) dst = src;
    else if (!is_v) dst = kpool + prow * ldp + (long)(hh - nq) * dp + e0;
    else dst = vpool + prow * ldp + (long)(hh - nq - nkv) * dp + e0;
#pragma unroll
    for (int e = 0; e < EPL / 2; ++e) *(unsigned*)(dst + 2 * e) = orr[e];
    if (!is_q)
        for (int off = HD; off < dp; off += HD)     // zero the pad lanes of the page row (the pools are uninitialised memory)
#pragma unroll
            for (int e = 0; e < EPL / 2; ++e) *(unsigned*)(dst + off + 2 * e) = 0u;
}

extern "C" int bagel_decode_qkv_post_bf16(void* qkv, int64_t ld, const void* cos_tab, const void* sin_tab, const void* q_w,
                                          const void* k_w, void* kpool, void* vpool, int64_t ld_pool, const int32_t* block_table,
                                          int32_t bt_stride, const int32_t* kv_len, int32_t batch, int32_t nq, int32_t nkv,
                                          int32_t head_dim, int32_t head_dim_padded, float eps, int32_t use_norm,
                                          hipStream_t stream) {
    BAGEL_REQUIRE(qkv && cos_tab && sin_tab && kpool && vpool && block_table && kv_len, "decode_qkv_post: null pointer");
    BAGEL_REQUIRE(!use_norm || (q_w && k_w), "decode_qkv_post: norm weights missing");
    BAGEL_REQUIRE(head_dim_padded >= head_dim && head_dim_padded % head_dim == 0 && ld % 2 == 0 && ld_pool % 2 == 0,
                  "decode_qkv_post: bad head_dim_padded/ld");
    if (batch <= 0) return BAGEL_OK;
    const dim3 grid(ceil_div(nq + 2 * nkv, 16), batch), block(256);
#define QKP_LAUNCH(EPL)                                                                                                          \
    hipLaunchKernelGGL(decode_qkv_post_kernel<EPL>, grid, block, 0, stream, (bf16_t*)qkv, (long)ld, (const bf16_t*)cos_tab,      \
                       (const bf16_t*)sin_tab, (const bf16_t*)q_w, (const bf16_t*)k_w, (bf16_t*)kpool, (bf16_t*)vpool,           \
                       (long)ld_pool, block_table, bt_stride, kv_len, nq, nkv, head_dim_padded, eps, use_norm)
    switch (head_dim) {
        case 128: QKP_LAUNCH(8); break;
        case 64: QKP_LAUNCH(4); break;
        case 32: QKP_LAUNCH(2); break;
        default: return bagel_set_error(BAGEL_ERR_UNSUPPORTED, "decode_qkv_post: head_dim %d not in {32,64,128}", head_dim);
    }
#undef QKP_LAUNCH
    return bagel_check_launch("decode_qkv_post_kernel");
}

// Lq = 1 attention, split over the keys.  Workgroup (split, kv head, sample) covers keys [split*CH, +CH) for the G query
// heads of one KV head, so each K/V byte is read once per GQA group.  DP/8 lanes own one key row (16 bytes each); the
// 256/(DP/8) lane groups walk the chunk with a running (max, sum, acc) per head in base 2; groups are merged by wave
// shuffles, waves through LDS.  Output: unnormalised fp32 partials + (max, sum) per (sample, head, split).
#define DEC_CH 128

// Optional fusion of the QKV epilogue (decode_qkv_post_kernel) into the attention kernel: `q` then points at the RAW fused
// projection row [nq*DP | nkv*DP | nkv*DP]; every lane group normalises/rotates the G query heads it needs itself (its DP/8
// lanes hold exactly one head slice each, so the per-head reduction and the rotate-half partner are group-local), and the
// lane group that owns key position kv_len[b] builds the new K row from the projection, uses it and stores K and V into the
// page slot.  One launch fewer per layer; arithmetic identical to decode_qkv_post_kernel (bit-exact, tested).
struct DecFuse {
    const bf16_t* cosb; const bf16_t* sinb;     // [B, HD/2]
    const bf16_t* qw; const bf16_t* kw;         // norm weights (HD) or nullptr
    bf16_t* kpool; bf16_t* vpool;               // writable views of the pools
    int nkv, hd, use_norm; float eps;
};

// one head slice (8 elements at lane position `sub` of a DP-padded head) through q_norm/k_norm + RoPE, und cast points
template <int LPK>
__device__ __forceinline__ void dec_post_head(const u32x4_t raw, const bf16_t* __restrict__ w, const float (&cs)[8], const float (&sn)[8],
                                              int sub, int hd, float eps, int use_norm, float (&out)[8]) {
    const bool real = sub * 8 < hd;
    const int hl = hd >> 4;                      // lanes per rotate half (hd/2 elements / 8)
    float x[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { x[2 * e] = lo2f(raw[e]); x[2 * e + 1] = hi2f(raw[e]); }
    float nrm[8];
    if (use_norm) {
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) ss += x[e] * x[e];
#pragma unroll
        for (int o = LPK / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        const float inv = rsqrtf(ss / (float)hd + eps);
        u32x4_t wr = {0u, 0u, 0u, 0u};
        if (real) wr = *(const u32x4_t*)(w + sub * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            nrm[2 * e] = bfround(lo2f(wr[e]) * bfround(x[2 * e] * inv));
            nrm[2 * e + 1] = bfround(hi2f(wr[e]) * bfround(x[2 * e + 1] * inv));
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) nrm[e] = x[e];
    }
    const bool upper = sub >= hl;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float partner = __shfl_xor(nrm[e], hl, 64);
        const float rot = upper ? partner : -partner;
        const float v = bfround(nrm[e] * cs[e]) + bfround(rot * sn[e]);
        out[e] = real ? bfround(v) : 0.f;
    }
}

// The kernel.  4 waves x 32 keys.  Both products run on the matrix pipe (mfma_f32_16x16x32_bf16), the GQA group padded to 16 columns:
//   S^T[key, head] = K[16 keys, DP] Q^T[DP, 16 heads]   K fragments straight from the page rows (lane = key l%16, 16 bytes at
//                                                        chunk 4kk + l/16), Q fragments from the projection row (or from LDS, FUSED);
//                                                        a lane ends up with the scores of ONE head (l%16) for keys 4g..4g+3 of both
//                                                        16-key blocks -- exactly the 8 keys an MFMA B operand of PV wants per lane;
//   O^T[d, head]   = V^T[16 d, 32 keys] P^T[32 keys, 16 heads]   P from the score registers (bf16, like flash-attn), V^T fragments by
//                                                        2-byte LDS reads of the wave's row-major V tile (row stride DP + 8: the four
//                                                        lane groups hit banks 16 apart).
// The VALU form this replaces spent 7.5 of its 12.3 us on 7 heads x 8 keys of FMA chains and ~360 ds_bpermute per lane
// (profiles/r02_decode_attn_ablation.log); here the softmax is 8 values per lane and two cross-group exchanges.
// FUSED: the q/k norm + RoPE of decode_qkv_post_kernel runs in the prologue -- 16-lane groups own one head each (G query heads + the
// new key: 8 items = two waves at 7B), same arithmetic (dec_post_head), results handed to all waves through LDS; the workgroup whose
// chunk holds position kv_len[b] uses the new K/V row from LDS / the projection and stores both into the page.
// cpw > 1 (batched decode: many requests x many chunks): a workgroup walks `cpw` consecutive 128-key chunks with a running (max, sum, O)
// per wave and leaves ONE partial -- the single-shot form sends B x nkv x 39 workgroups through the chip in rounds whose load and
// compute phases line up (49.5 us per layer at 16 requests = 3.3 TB/s); with a loop the resident workgroups drift apart and the
// page loads of one overlap the MFMAs of another, and the combine pass reads cpw times fewer partials.  The block-table entries of
// the NEXT chunk are fetched under the current one (the table -> page -> row chain is two dependent latencies otherwise).
template <int DP, bool FUSED>
__global__ __launch_bounds__(256) void attn_decode_kernel(const bf16_t* __restrict__ q, long ldq, const bf16_t* __restrict__ kpool,
                                                          const bf16_t* __restrict__ vpool, long ldp,
                                                          const int* __restrict__ block_table, int bt_stride,
                                                          const int* __restrict__ kv_len, int len_add, float* __restrict__ part_o,
                                                          float* __restrict__ part_ml, int nq, int G, int nsplit, float scale_log2e,
                                                          DecFuse fu, int cpw) {
    constexpr int CH = DEC_CH;           // keys per chunk
    constexpr int KS = DP / 32;          // k-steps of S^T
    constexpr int NDB = DP / 16;         // 16-wide d blocks of O^T
    constexpr int VST = DP + 8;          // LDS row stride of the V tile, elements
    constexpr int CPR = DP / 8;          // 16-byte chunks per K/V row
    constexpr int RPI = 64 / CPR;        // rows one wave-wide load covers
    constexpr int NVL = 32 / RPI;        // V loads per lane
    const int split = blockIdx.x, kvh = blockIdx.y, b = blockIdx.z;
    const int L = kv_len[b] + len_add;
    const int j0 = split * CH * cpw;                                 // this workgroup's key range [j0, j1)
    const int j1 = (L < j0 + CH * cpw) ? L : j0 + CH * cpw;
    if (j0 >= j1) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c16 = lane & 15;
    __shared__ __attribute__((aligned(16))) bf16_t sm_v[4][32 * VST];      // V tiles; afterwards the waves' O partials [16][DP] fp32
    __shared__ __attribute__((aligned(16))) bf16_t sm_q[17][DP];           // FUSED: finished q heads [0, G) and the new key [G]
    __shared__ float sm_m[4][16], sm_l[4][16];
    static_assert(16 * DP * 4 <= 32 * VST * 2, "O partial must fit the V tile");

    const int* bt = block_table + (long)b * bt_stride;
    const int jn = L - 1;                // FUSED: the position this step appends
    const bf16_t* qrow = q + (long)b * ldq;
    // page rows of the K fragments / V rows of the chunk that starts at key c0 (clamped past the workgroup's range)
    int krow[2], vrow[NVL];              // page ROW indices (pool rows) of the chunk whose loads go out next
    auto rows_of = [&](int c0) {
        const int jw = c0 + 32 * wave;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const int j = jw + 16 * blk + c16;
            const int jc = j < j1 ? j : j1 - 1;
            krow[blk] = bt[jc / BAGEL_KV_PAGE] * BAGEL_KV_PAGE + (jc % BAGEL_KV_PAGE);
        }
#pragma unroll
        for (int i = 0; i < NVL; ++i) {
            const int j = jw + lane / CPR + RPI * i;
            const int jc = j < j1 ? j : j1 - 1;
            vrow[i] = bt[jc / BAGEL_KV_PAGE] * BAGEL_KV_PAGE + (jc % BAGEL_KV_PAGE);
        }
    };
    u32x4_t kf[2][KS], vr[NVL];
    auto load_kv = [&](int c0) {
        const int jw = c0 + 32 * wave;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
                kf[blk][kk] = ld_stream_kv<u32x4_t>(kpool + (long)krow[blk] * ldp + (long)kvh * DP + 32 * kk + 8 * g);
#pragma unroll
        for (int i = 0; i < NVL; ++i) {
            const bf16_t* src = vpool + (long)vrow[i] * ldp + (long)kvh * DP + (lane % CPR) * 8;
            if (FUSED) {         // the new V row is not in the page yet: take it from the projection
                const int j = jw + lane / CPR + RPI * i;
                if (j == jn || (j >= j1 && j1 - 1 == jn)) src = qrow + (long)(nq + fu.nkv + kvh) * DP + (lane % CPR) * 8;
            }
            vr[i] = ld_stream_kv<u32x4_t>(src);
        }
    };
    // ---- one latency round: block-table entries -> K fragments and V rows of the first chunk, the query heads beside them
    rows_of(j0);
    load_kv(j0);
    if (j0 + CH < j1) rows_of(j0 + CH);          // the table entries of chunk 2 travel under chunk 1's page loads
    u32x4_t qf[KS];
    if (!FUSED) {
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            qf[kk] = u32x4_t{0u, 0u, 0u, 0u};
            if (c16 < G) qf[kk] = *(const u32x4_t*)(qrow + (long)(kvh * G + c16) * DP + 32 * kk + 8 * g);
        }
    } else {
        constexpr int LPK = DP / 8, IPW = 64 / LPK;
        const int sub = lane % LPK, item = wave * IPW + lane / LPK;      // items [0, G): query heads; G: the new key
        const int hd = fu.hd, half = hd >> 1;
        if (item <= G) {     // whole 16-lane groups take this branch together: the shuffles of dec_post_head stay inside a group
            const bool real = sub * 8 < hd;
            float cs[8], sn[8];
            {
                const int c0 = (sub * 8) % half;     // element e of the head uses table column e mod HD/2
                u32x4_t cv = {0u, 0u, 0u, 0u}, sv = {0u, 0u, 0u, 0u};
                if (real) {
                    cv = *(const u32x4_t*)(fu.cosb + (long)b * half + c0);
                    sv = *(const u32x4_t*)(fu.sinb + (long)b * half + c0);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) { cs[2 * e] = lo2f(cv[e]); cs[2 * e + 1] = hi2f(cv[e]); sn[2 * e] = lo2f(sv[e]); sn[2 * e + 1] = hi2f(sv[e]); }
            }
            const bool is_k = item == G;
            const u32x4_t raw = *(const u32x4_t*)(qrow + (long)(is_k ? nq + kvh : kvh * G + item) * DP + sub * 8);
            float o8[8];
            dec_post_head<LPK>(raw, is_k ? fu.kw : fu.qw, cs, sn, sub, hd, fu.eps, fu.use_norm, o8);
            const u32x4_t fin = {pack2bf(o8[0], o8[1]), pack2bf(o8[2], o8[3]), pack2bf(o8[4], o8[5]), pack2bf(o8[6], o8[7])};
            *(u32x4_t*)(&sm_q[item][sub * 8]) = fin;
            if (is_k && jn >= j0 && jn < j1) {       // this workgroup owns the new position: K and V into the page slot
                const long off = ((long)bt[jn / BAGEL_KV_PAGE] * BAGEL_KV_PAGE + (jn % BAGEL_KV_PAGE)) * ldp + (long)kvh * DP + sub * 8;
                *(u32x4_t*)(fu.kpool + off) = fin;
                *(u32x4_t*)(fu.vpool + off) = *(const u32x4_t*)(qrow + (long)(nq + fu.nkv + kvh) * DP + sub * 8);
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            qf[kk] = u32x4_t{0u, 0u, 0u, 0u};
            if (c16 < G) qf[kk] = *(const u32x4_t*)(&sm_q[c16][32 * kk + 8 * g]);
        }
    }

    // running state of this wave over its chunks: one (max, sum) per head (lane column c16), O^T[d, head] in oacc
    float m_run = -1e30f, l_run = 0.f;
    f32x4_t oacc[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db) oacc[db] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    bf16_t* svw = sm_v[wave];
    for (int c0 = j0; c0 < j1; c0 += CH) {
        const int jw = c0 + 32 * wave;
        const bool more = c0 + CH < j1;
        if (FUSED) {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const int j = jw + 16 * blk + c16;
                if (j == jn || (j >= j1 && j1 - 1 == jn)) {
#pragma unroll
                    for (int kk = 0; kk < KS; ++kk) kf[blk][kk] = *(const u32x4_t*)(&sm_q[G][32 * kk + 8 * g]);
                }
            }
        }
        // ---- S^T = K Q^T: lane (head c16) gets keys 4g + r of both blocks
        f32x4_t sacc[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            sacc[blk] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
                sacc[blk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf[blk][kk]), __builtin_bit_cast(bf16x8_t, qf[kk]),
                                                                   sacc[blk], 0, 0, 0);
        }
        // ---- V tile -> LDS (row major), while the scores settle; then the next chunk's page rows and loads go out
#pragma unroll
        for (int i = 0; i < NVL; ++i) *(u32x4_t*)(svw + (lane / CPR + RPI * i) * VST + (lane % CPR) * 8) = vr[i];
        if (more) {
            load_kv(c0 + CH);                             // its page rows were resolved one chunk ago
            if (c0 + 2 * CH < j1) rows_of(c0 + 2 * CH);
        }
        // ---- softmax over the wave's 32 keys, base 2, one max per head
        float sc[2][4];
        bool ok[2][4];
        float mx = -1e30f;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ok[blk][r] = (jw + 16 * blk + 4 * g + r) < j1;
                sc[blk][r] = ok[blk][r] ? sacc[blk][r] * scale_log2e : -1e30f;
                mx = fmaxf(mx, sc[blk][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));          // -1e30 when the wave holds no key of the chunk
        const float m_new = fmaxf(m_run, mx);
        const float alpha = exp2f(m_run - m_new);        // 1 on the first chunk's empty state (both -1e30), 0 when the first keys arrive
        float pw[2][4], ls = 0.f;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pw[blk][r] = ok[blk][r] ? exp2f(sc[blk][r] - m_new) : 0.f;
                ls += pw[blk][r];
            }
        ls += __shfl_xor(ls, 16, 64);
        ls += __shfl_xor(ls, 32, 64);
        l_run = l_run * alpha + ls;
        m_run = m_new;
        const u32x4_t pfrag = {pack2bf(pw[0][0], pw[0][1]), pack2bf(pw[0][2], pw[0][3]), pack2bf(pw[1][0], pw[1][1]), pack2bf(pw[1][2], pw[1][3])};
        // ---- O^T = O^T alpha + V^T P^T: the lane's 8 keys are rows 4g..4g+3 and 16+4g..16+4g+3 of the tile, column = d
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the wave's own ds_writes above (other lanes' rows) have landed
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            const bf16_t* col = svw + 16 * db + c16;
            unsigned short e[8];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                e[t] = col[(4 * g + t) * VST];
                e[4 + t] = col[(16 + 4 * g + t) * VST];
            }
            const u32x4_t vfrag = {(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16),
                                   (unsigned)e[4] | ((unsigned)e[5] << 16), (unsigned)e[6] | ((unsigned)e[7] << 16)};
            oacc[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vfrag), __builtin_bit_cast(bf16x8_t, pfrag),
                                                              oacc[db] * alpha, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();                  // every lane's reads of the tile are issued before the next chunk's rows overwrite it
    }
    // ---- the wave's (max, sum, O) -> LDS (O over its own V tile: every read of the tile is older than these writes), merge
    const float mx = m_run, ls = l_run;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float* so = (float*)svw;
    if (c16 < G) {
#pragma unroll
        for (int db = 0; db < NDB; ++db) *(f32x4_t*)(so + c16 * DP + 16 * db + 4 * g) = oacc[db];
        if (g == 0) { sm_m[wave][c16] = mx; sm_l[wave][c16] = ls; }
    }
    __syncthreads();
    for (int idx = tid; idx < G * DP; idx += 256) {
        const int h = idx / DP, d = idx - h * DP;
        const float m0 = sm_m[0][h], m1 = sm_m[1][h], m2 = sm_m[2][h], m3 = sm_m[3][h];
        const float mn = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
        const float c0 = exp2f(m0 - mn), c1 = exp2f(m1 - mn), c2 = exp2f(m2 - mn), c3 = exp2f(m3 - mn);
        const float acc = ((const float*)sm_v[0])[idx] * c0 + ((const float*)sm_v[1])[idx] * c1 + ((const float*)sm_v[2])[idx] * c2 +
                          ((const float*)sm_v[3])[idx] * c3;
        const long slot = ((long)b * nq + kvh * G + h) * nsplit + split;
        part_o[slot * DP + d] = acc;
        if (d == 0) {
            part_ml[slot * 2] = mn;
            part_ml[slot * 2 + 1] = sm_l[0][h] * c0 + sm_l[1][h] * c1 + sm_l[2][h] * c2 + sm_l[3][h] * c3;
        }
    }
}

// out[b, h, :] = sum_s part_o[s] 2^(m_s - M) / sum_s l_s 2^(m_s - M)  over the splits that hold keys.  One workgroup per
// (head, sample): DP/4 lanes cover the head dim with float4 loads, 256/(DP/4) lane groups stride over the splits with a
// running (max, sum, acc); the groups are merged through LDS.
template <int DP>
__global__ __launch_bounds__(256) void attn_decode_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                                                  const int* __restrict__ kv_len, int len_add, bf16_t* __restrict__ out,
                                                                  long ldo, int nq, int nsplit, int ch) {
    constexpr int LPD = DP / 4;
    constexpr int NSG = 256 / LPD;
    const int h = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    const int d4 = t % LPD, sg = t / LPD;
    const int L = kv_len[b] + len_add;
    int ns = (L + ch - 1) / ch;
    if (ns > nsplit) ns = nsplit;
    const long base = ((long)b * nq + h) * nsplit;
    __shared__ float sm_m[NSG], sm_l[NSG];
    __shared__ f32x4_t sm_o[NSG][LPD];
    float m = -1e30f, l = 0.f;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int s = sg; s < ns; s += NSG) {
        const float ms = part_ml[(base + s) * 2], lsv = part_ml[(base + s) * 2 + 1];
        const f32x4_t ov = *(const f32x4_t*)(part_o + (base + s) * DP + d4 * 4);
        const float mn = fmaxf(m, ms);
        const float c1 = exp2f(m - mn), c2 = exp2f(ms - mn);
        acc = acc * c1 + ov * c2;
        l = l * c1 + lsv * c2;
        m = mn;
    }
    sm_o[sg][d4] = acc;
    if (d4 == 0) { sm_m[sg] = m; sm_l[sg] = l; }
    __syncthreads();
    if (t < LPD) {
        float mn = -1e30f;
#pragma unroll
        for (int g = 0; g < NSG; ++g) mn = fmaxf(mn, sm_m[g]);
        f32x4_t tot = {0.f, 0.f, 0.f, 0.f};
        float den = 0.f;
#pragma unroll
        for (int g = 0; g < NSG; ++g) {
            const float c = exp2f(sm_m[g] - mn);
            tot = tot + sm_o[g][t] * c;
            den = fmaf(sm_l[g], c, den);
        }
        const float inv = den > 0.f ? 1.f / den : 0.f;
        u32x2_t v = {pack2bf(tot[0] * inv, tot[1] * inv), pack2bf(tot[2] * inv, tot[3] * inv)};
        *(u32x2_t*)(out + (long)b * ldo + (long)h * DP + t * 4) = v;
    }
}

template <int DP, bool FUSED>
static int launch_attn_decode(int G, dim3 grid, hipStream_t stream, const bf16_t* q, long ldq, const bf16_t* kpool, const bf16_t* vpool,
                              long ldp, const int* bt, int bt_stride, const int* kv_len, int len_add, float* po, float* pml, int nq,
                              int nsplit, float sl2e, DecFuse fu, int cpw) {
    hipLaunchKernelGGL((attn_decode_kernel<DP, FUSED>), grid, dim3(256), 0, stream, q, ldq, kpool, vpool, ldp, bt, bt_stride, kv_len,
                       len_add, po, pml, nq, G, nsplit, sl2e, fu, cpw);
    return bagel_check_launch("attn_decode_kernel");
}

static int attn_decode_common(const void* q, int64_t ldq, const void* kpool, const void* vpool, int64_t ld_pool,
                              const int32_t* block_table, int32_t bt_stride, const int32_t* kv_len, int32_t len_add, int32_t max_len,
                              float* part_o, float* part_ml, void* out, int64_t ldo, int32_t batch, int32_t nq, int32_t nkv,
                              int32_t head_dim, float softmax_scale, const DecFuse* fuse, hipStream_t stream) {
    BAGEL_REQUIRE(q && kpool && vpool && block_table && kv_len && part_o && part_ml && out, "attn_decode: null pointer");
    BAGEL_REQUIRE(head_dim == 64 || head_dim == 128, "attn_decode: head_dim %d not in {64,128} (pad the projection)", head_dim);
    BAGEL_REQUIRE(nkv > 0 && nq % nkv == 0, "attn_decode: nq must be a multiple of nkv");
    BAGEL_REQUIRE((ldq % 8) == 0 && (ld_pool % 8) == 0 && (ldo % 4) == 0 && (((uintptr_t)out) & 7) == 0, "attn_decode: leading dims / out alignment");
    BAGEL_REQUIRE((((uintptr_t)q | (uintptr_t)kpool | (uintptr_t)vpool) & 15) == 0, "attn_decode: 16-byte alignment");
    if (batch <= 0 || max_len <= 0) return BAGEL_OK;
    // chunks per workgroup: one while the launch is a single round of workgroups anyway (one request: 156 of them at 7B / 4 936 keys --
    // latency-bound, most parallel form), more once requests x KV heads x chunks would go through the chip in several rounds
    // (BAGEL_DEC_CPW overrides: tuning / tests)
    static int cpw_env = -1;
    if (cpw_env < 0) {
        const char* e = getenv("BAGEL_DEC_CPW");
        cpw_env = (e && atoi(e) > 0) ? atoi(e) : 0;
    }
    const int nchunks = ceil_div(max_len, DEC_CH);
    int cpw = cpw_env ? cpw_env : ceil_div((long)nchunks * nkv * batch, 512);     // two resident workgroups per CU, one round
    if (cpw > 5 && !cpw_env) cpw = 5;           // 32 requests x 4 936 keys: 4-5 chunks per workgroup (1 024-1 280 workgroups) 72.7 us per layer, 8 (the former cap) 79.2 -- profiles/r06_gemv_mb_32rows.log
    if (cpw > 16) cpw = 16;
    if (cpw > nchunks) cpw = nchunks;
    if (cpw < 1) cpw = 1;
    const int ch = DEC_CH * cpw;                   // keys per workgroup = per partial slot
    const int nsplit = ceil_div(max_len, ch);
    const float sl2e = softmax_scale * 1.4426950408889634f;
    const dim3 grid(nsplit, nkv, batch);
    const bf16_t *qq = (const bf16_t*)q, *kp = (const bf16_t*)kpool, *vp = (const bf16_t*)vpool;
    DecFuse fu = {};
    if (fuse) fu = *fuse;
    const int G = nq / nkv;
    BAGEL_REQUIRE(G <= 16, "attn_decode: GQA group %d > 16", G);
    // FUSED prologue: one 16-byte-per-lane group per item (G query heads + the new key), 4 waves x 64 / (DP / 8) groups, one round
    BAGEL_REQUIRE(!fuse || G + 1 <= 4 * (64 / (head_dim / 8)),
                  "attn_decode_fused: GQA group %d + the new key do not fit the %d prologue groups at padded head_dim %d (use decode_qkv_post + attn_decode_paged)",
                  G, 4 * (64 / (head_dim / 8)), head_dim);
    int rc;
#define DEC_GO(DPV)                                                                                                              \
    rc = fuse ? launch_attn_decode<DPV, true>(G, grid, stream, qq, (long)ldq, kp, vp, (long)ld_pool, block_table, bt_stride, kv_len, \
                                              len_add, part_o, part_ml, nq, nsplit, sl2e, fu, cpw)                               \
              : launch_attn_decode<DPV, false>(G, grid, stream, qq, (long)ldq, kp, vp, (long)ld_pool, block_table, bt_stride, kv_len, \
                                               len_add, part_o, part_ml, nq, nsplit, sl2e, fu, cpw)
    if (head_dim == 128) { DEC_GO(128); }
    else { DEC_GO(64); }
#undef DEC_GO
    if (rc != BAGEL_OK) return rc;
    if (head_dim == 128)
        hipLaunchKernelGGL((attn_decode_combine_kernel<128>), dim3(nq, batch), dim3(256), 0, stream, part_o, part_ml, kv_len, len_add,
                           (bf16_t*)out, (long)ldo, nq, nsplit, ch);
    else
        hipLaunchKernelGGL((attn_decode_combine_kernel<64>), dim3(nq, batch), dim3(256), 0, stream, part_o, part_ml, kv_len, len_add,
                           (bf16_t*)out, (long)ldo, nq, nsplit, ch);
    return bagel_check_launch("attn_decode_combine_kernel");
}

extern "C" int bagel_attn_decode_paged_bf16(const void* q, int64_t ldq, const void* kpool, const void* vpool, int64_t ld_pool,
                                            const int32_t* block_table, int32_t bt_stride, const int32_t* kv_len,
                                            int32_t len_add, int32_t max_len, float* part_o, float* part_ml, void* out,
                                            int64_t ldo, int32_t batch, int32_t nq, int32_t nkv, int32_t head_dim,
                                            float softmax_scale, hipStream_t stream) {
    return attn_decode_common(q, ldq, kpool, vpool, ld_pool, block_table, bt_stride, kv_len, len_add, max_len, part_o, part_ml, out, ldo,
                              batch, nq, nkv, head_dim, softmax_scale, nullptr, stream);
}

// decode_qkv_post + attn_decode_paged in one launch (+ the combine): qkv = RAW fused projection rows; the new K/V row of every
// sample goes to page slot kv_len[b] and takes part in the attention (keys [0, kv_len[b]]).
extern "C" int bagel_attn_decode_fused_bf16(const void* qkv, int64_t ld, const void* cos_tab, const void* sin_tab, const void* q_w,
                                            const void* k_w, void* kpool, void* vpool, int64_t ld_pool, const int32_t* block_table,
                                            int32_t bt_stride, const int32_t* kv_len, int32_t max_len, float* part_o, float* part_ml,
                                            void* out, int64_t ldo, int32_t batch, int32_t nq, int32_t nkv, int32_t head_dim,
                                            int32_t head_dim_padded, float eps, int32_t use_norm, float softmax_scale,
                                            hipStream_t stream) {
    BAGEL_REQUIRE(cos_tab && sin_tab, "attn_decode_fused: null rope tables");
    BAGEL_REQUIRE(!use_norm || (q_w && k_w), "attn_decode_fused: norm weights missing");
    BAGEL_REQUIRE((head_dim == 32 || head_dim == 64 || head_dim == 128) && head_dim <= head_dim_padded, "attn_decode_fused: head_dim %d / padded %d",
                  head_dim, head_dim_padded);
    BAGEL_REQUIRE((((uintptr_t)cos_tab | (uintptr_t)sin_tab | (uintptr_t)q_w | (uintptr_t)k_w) & 15) == 0 && (head_dim / 2) % 8 == 0,
                  "attn_decode_fused: rope tables / norm weights must be 16-byte aligned, head_dim/2 a multiple of 8");
    DecFuse fu;
    fu.cosb = (const bf16_t*)cos_tab; fu.sinb = (const bf16_t*)sin_tab; fu.qw = (const bf16_t*)q_w; fu.kw = (const bf16_t*)k_w;
    fu.kpool = (bf16_t*)kpool; fu.vpool = (bf16_t*)vpool; fu.nkv = nkv; fu.hd = head_dim; fu.use_norm = use_norm; fu.eps = eps;
    return attn_decode_common(qkv, ld, kpool, vpool, ld_pool, block_table, bt_stride, kv_len, 1, max_len, part_o, part_ml, out, ldo, batch,
                              nq, nkv, head_dim_padded, softmax_scale, &fu, stream);
}

// Token bookkeeping of one decode step on the device (bagel.py:984-994): the chosen token becomes the next input,
// positions and KV lengths advance, the token is logged at tokens_out[(step+1), b].
__global__ void decode_advance_kernel(const long* __restrict__ next_tok, int* __restrict__ cur_tok32, long* __restrict__ tokens_out,
                                      long* __restrict__ pos, int* __restrict__ kv_len, int* __restrict__ step, int batch,
                                      int max_steps) {
    const int b = threadIdx.x;
    const int s = *step;
    __syncthreads();
    if (b < batch) {
        const long t = next_tok[b];
        cur_tok32[b] = (int)t;
        if (s + 1 < max_steps) tokens_out[(long)(s + 1) * batch + b] = t;
        pos[b] += 1;
        kv_len[b] += 1;
    }
    if (b == 0) *step = s + 1;
}

extern "C" int bagel_decode_advance(const int64_t* next_tok, int32_t* cur_tok32, int64_t* tokens_out, int64_t* pos,
                                    int32_t* kv_len, int32_t* step, int32_t batch, int32_t max_steps, hipStream_t stream) {
    BAGEL_REQUIRE(next_tok && cur_tok32 && tokens_out && pos && kv_len && step, "decode_advance: null pointer");
    BAGEL_REQUIRE(batch > 0 && batch <= 1024, "decode_advance: batch %d not in [1,1024]", batch);
    hipLaunchKernelGGL(decode_advance_kernel, dim3(1), dim3(((batch + 63) / 64) * 64), 0, stream, (const long*)next_tok, cur_tok32,
                       (long*)tokens_out, (long*)pos, kv_len, step, batch, max_steps);
    return bagel_check_launch("decode_advance_kernel");
}
