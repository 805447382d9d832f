// MXFP4 weight-only projections for the text-decode path (option; the MI355X counterpart of the reference's bitsandbytes NF4 load mode,
// app.py:114-125): weights as OCP-MX FP4 (E2M1) with one E8M0 power-of-two scale per 32 elements along K, activations as FP8 (e4m3,
// one fp32 scale per row), the product on the matrix pipe by v_mfma_scale_f32_16x16x128_f8f6f4 -- the de-quantisation IS the MFMA
// (the INT8 option of quant.hip de-quantises on the VALU and is VALU-bound below ~2.4 ms/token).  oracle/mxfp4.py states the scheme.
//
//   bagel_quantize_rows_mxfp4   bf16 weights [N, K] -> codes [N, K/2] (element 2i in the low nibble of byte i) + scale bytes in the
//                               device order of oracle/mxfp4.py permute_scales (a lane's four next k-steps in one dword); offline
//   bagel_gemv_w4_bf16          C[M <= 4, N] = epilogue(A W^T), optional fused Qwen2RMSNorm of the A rows (modeling_qwen2.py:54-59),
//                               bias / SwiGLU16 / residual epilogues with the rounding points of gemm.hip
//
// Kernel = the skinny MFMA GEMM of skinny.hip with 4-bit weight fragments: a wave owns 16 weight rows (SwiGLU16: 16 gate + 16 up) and
// a K range; per 128-deep step a lane loads 16 bytes of its weight row straight from HBM (row-major codes ARE the fragment layout:
// row lane%16, elements 32 (lane/16) ..) and one scale dword per four steps; the activation rows are normalised and quantised ONCE per
// workgroup into LDS in the prologue (no extra launch: a launch costs ~4.7 us on this path, a whole 4-bit projection streams in 1-12).
#include "common.h"
#include <stdlib.h>

#define EPI_NONE 0
#define EPI_SWIGLU16 3

typedef __attribute__((ext_vector_type(8))) int i32x8_t;
#ifndef BAGEL_W4_NT
#define BAGEL_W4_NT 0      // a wave-wide fragment load covers 16 rows x 64 B: the half lines of two consecutive k-steps meet in L1/L2 -- nt measured slower (26.6 vs 23.6 us gate+up)
#endif
__device__ __forceinline__ u32x4_t w4_ld(const void* p) {
#if BAGEL_W4_NT
    return __builtin_nontemporal_load((const u32x4_t*)p);
#else
    return *(const u32x4_t*)p;
#endif
}

// ---- quantiser: one wave per row, a lane per 32-element block ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void mxfp4_quantize_rows_kernel(const bf16_t* __restrict__ w, long ldw, unsigned char* __restrict__ q,
                                                                   long ldq, unsigned char* __restrict__ s, long lds, int rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nb = cols >> 5;
    for (int b = lane; b < nb; b += 64) {
        const bf16_t* src = w + (long)row * ldw + (long)b * 32;
        u32x4_t v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = *(const u32x4_t*)(src + 8 * i);
        float x[32];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) { x[8 * i + 2 * e] = lo2f(v[i][e]); x[8 * i + 2 * e + 1] = hi2f(v[i][e]); }
        float amax = 0.f;
#pragma unroll
        for (int e = 0; e < 32; ++e) amax = fmaxf(amax, fabsf(x[e]));
        int sb = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 2;
        sb = sb < 0 ? 0 : sb;
        const float rx = __uint_as_float((unsigned)(254 - sb) << 23);      // 1 / X = 2^(127 - sb), exact (sb <= 253: amax is finite)
        unsigned words[4];
#pragma unroll
        for (int wd = 0; wd < 4; ++wd) {
            unsigned acc = 0u;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xv = x[8 * wd + e];
                const float a = fabsf(xv) * rx;
                unsigned c = (a > 0.25f) + (a >= 0.75f) + (a > 1.25f) + (a >= 1.75f) + (a > 2.5f) + (a >= 3.5f) + (a > 5.0f);
                c |= (xv < 0.f) ? 8u : 0u;
                acc |= c << (4 * e);
            }
            words[wd] = acc;
        }
        *(u32x4_t*)(q + (long)row * ldq + (long)b * 16) = u32x4_t{words[0], words[1], words[2], words[3]};
        // block b = k-step b/4, lane group b%4  ->  group (b/4)/4, byte 4*(b%4) + (b/4)%4
        const int kstep = b >> 2, qq = b & 3;
        s[(long)row * lds + (kstep >> 2) * 16 + qq * 4 + (kstep & 3)] = (unsigned char)sb;
    }
    // pad bytes of the last group (k-steps past K/128): 2^0
    const int nk = nb >> 2, ng = (nk + 3) >> 2;
    for (int i = lane; i < ng * 16; i += 64) {
        const int kstep = (i >> 4) * 4 + (i & 3);
        if (kstep >= nk) s[(long)row * lds + i] = 127;
    }
}

extern "C" int bagel_quantize_rows_mxfp4(const void* W, int64_t ldw, void* q, int64_t ldq_bytes, void* scales, int64_t lds_bytes,
                                         int32_t rows, int32_t cols, hipStream_t stream) {
    BAGEL_REQUIRE(W && q && scales, "quantize_rows_mxfp4: null pointer");
    BAGEL_REQUIRE(cols > 0 && (cols % 128) == 0 && (ldw % 8) == 0, "quantize_rows_mxfp4: cols %% 128 == 0 and 16-byte rows required");
    BAGEL_REQUIRE(ldq_bytes >= cols / 2 && (ldq_bytes % 16) == 0, "quantize_rows_mxfp4: ldq_bytes must be >= cols/2 and a multiple of 16");
    BAGEL_REQUIRE(lds_bytes >= ((cols / 128 + 3) / 4) * 16 && (lds_bytes % 4) == 0, "quantize_rows_mxfp4: lds_bytes too small");
    BAGEL_REQUIRE((((uintptr_t)W | (uintptr_t)q) & 15) == 0 && (((uintptr_t)scales) & 3) == 0, "quantize_rows_mxfp4: alignment");
    if (rows <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(mxfp4_quantize_rows_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, stream, (const bf16_t*)W, (long)ldw,
                       (unsigned char*)q, (long)ldq_bytes, (unsigned char*)scales, (long)lds_bytes, rows, cols);
    return bagel_check_launch("mxfp4_quantize_rows_kernel");
}

// ---- the projection ----------------------------------------------------------------------------------------------------------------
struct W4Params {
    const bf16_t* A; long lda;
    const unsigned char* Wq; long ldq;      // codes, bytes per row
    const unsigned char* Ws; long lds;      // permuted scale bytes, bytes per row
    const bf16_t* bias;
    const bf16_t* R; long ldr;
    bf16_t* C; long ldc;
    const bf16_t* norm_w; float eps;
    int M, N, K, epi;
    int split;
};

#define W4_MAX_M 4

// One activation row -> (Qwen2RMSNorm, und cast points) -> FP8 e4m3 with one scale, by the whole workgroup.  IN_REGS: a thread keeps its
// chunks of the row in registers (K / 8 chunks over the workgroup: 2 per thread at K = 3584 / 256 threads, 5 at K = 18944 / 512): one
// load round and two workgroup reductions; otherwise the row is re-read from L2 per pass.  Ends with a barrier.
template <int W4_CPT>
__device__ __forceinline__ void w4_load_row(const bf16_t* __restrict__ ar, const bf16_t* __restrict__ norm_w, int K, int tid, int nthr,
                                            u32x4_t (&xv)[W4_CPT], u32x4_t (&gv)[W4_CPT]) {
    const int nch = K >> 3;
    const u32x4_t zero = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < W4_CPT; ++i) {
        const int c = tid + i * nthr;
        xv[i] = c < nch ? *(const u32x4_t*)(ar + (long)c * 8) : zero;
    }
    if (norm_w) {
#pragma unroll
        for (int i = 0; i < W4_CPT; ++i) {
            const int c = tid + i * nthr;
            gv[i] = c < nch ? *(const u32x4_t*)(norm_w + (long)c * 8) : zero;
        }
    }
}
template <bool IN_REGS, int W4_CPT>
__device__ __forceinline__ void w4_quantise_row(const bf16_t* __restrict__ ar, const bf16_t* __restrict__ norm_w, float eps, int K,
                                                unsigned char* __restrict__ dst, float* __restrict__ scale_out, float* __restrict__ red,
                                                int tid, int nthr, u32x4_t (&xv)[W4_CPT], u32x4_t (&gv)[W4_CPT], bool preloaded) {
    const int lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
    const int nch = K >> 3;
    const int niter = IN_REGS ? W4_CPT : (nch + nthr - 1) / nthr;
    const u32x4_t zero = {0u, 0u, 0u, 0u};
    if (IN_REGS && !preloaded) w4_load_row<W4_CPT>(ar, norm_w, K, tid, nthr, xv, gv);
    float inv = 1.f;
    if (norm_w) {
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < niter; ++i) {
            const int c = tid + i * nthr;
            const u32x4_t v = IN_REGS ? xv[IN_REGS ? i : 0] : (c < nch ? *(const u32x4_t*)(ar + (long)c * 8) : zero);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float a = lo2f(v[e]), b = hi2f(v[e]); ss += a * a + b * b; }
        }
        ss = wave_sum(ss);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        float tot = 0.f;
        for (int i = 0; i < nw; ++i) tot += red[i];
        inv = rsqrtf(tot / (float)K + eps);
        __syncthreads();
    }
    auto normed = [&](u32x4_t v, const u32x4_t g) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = pack2bf(bfround(lo2f(v[e]) * inv) * lo2f(g[e]), bfround(hi2f(v[e]) * inv) * hi2f(g[e]));
        return v;
    };
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < niter; ++i) {
        const int c = tid + i * nthr;
        const bool ok = c < nch;
        u32x4_t v = IN_REGS ? xv[IN_REGS ? i : 0] : (ok ? *(const u32x4_t*)(ar + (long)c * 8) : zero);
        if (norm_w) {
            v = normed(v, IN_REGS ? gv[IN_REGS ? i : 0] : (ok ? *(const u32x4_t*)(norm_w + (long)c * 8) : zero));
            if (IN_REGS) xv[IN_REGS ? i : 0] = v;        // the normalised row replaces the raw one (bf16 pairs)
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fmaxf(fabsf(lo2f(v[e])), fabsf(hi2f(v[e]))));
    }
    amax = wave_max(amax);
    if (lane == 0) red[wave] = amax;
    __syncthreads();
    float am = 0.f;
    for (int i = 0; i < nw; ++i) am = fmaxf(am, red[i]);
    const float sq = am > 0.f ? am / 448.0f : 1.0f;
    const float qinv = 1.0f / sq;
    if (tid == 0) *scale_out = sq;
#pragma unroll
    for (int i = 0; i < niter; ++i) {
        const int c = tid + i * nthr;
        const bool ok = c < nch;
        u32x4_t v = IN_REGS ? xv[IN_REGS ? i : 0] : (ok ? *(const u32x4_t*)(ar + (long)c * 8) : zero);
        if (norm_w && !IN_REGS) v = normed(v, ok ? *(const u32x4_t*)(norm_w + (long)c * 8) : zero);
        u32x2_t o;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int wd = 0;
            wd = __builtin_amdgcn_cvt_pk_fp8_f32(lo2f(v[2 * h]) * qinv, hi2f(v[2 * h]) * qinv, wd, false);
            wd = __builtin_amdgcn_cvt_pk_fp8_f32(lo2f(v[2 * h + 1]) * qinv, hi2f(v[2 * h + 1]) * qinv, wd, true);
            o[h] = (unsigned)wd;
        }
        if (ok) *(u32x2_t*)(dst + (long)c * 8) = o;
    }
    __syncthreads();
}

// W4_CPT: 16-byte activation chunks a thread holds (2: K <= 16 x threads, the K = 3584 projections; 5: the down projection)
template <bool SWIGLU, int W4_CPT>
__global__ __launch_bounds__(512) void gemv_w4_kernel(W4Params p) {
    constexpr int KU = 8;                       // 128-deep steps per batch (8 x 16 B of codes per lane in flight), two scale dwords
    constexpr int NACC = SWIGLU ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char w4_smem[];
    // LDS: [M][K] fp8 activations | per-wave partial accumulators (K splits) | reduction scratch
    unsigned char* xq = w4_smem;
    const int xq_bytes = (p.M * p.K + 15) & ~15;
    f32x4_t (*part)[NACC][64] = (f32x4_t (*)[NACC][64])(w4_smem + xq_bytes);      // [wave][acc][lane]
    __shared__ float red[8], sx[W4_MAX_M];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nthr = blockDim.x, nw = nthr >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int ncb = SWIGLU ? p.N / 32 : p.N / 16;
    const int nk = p.K >> 7;                    // 128-deep steps
    const int S = p.split;
    const int cb = blockIdx.x * (nw / S) + wave / S;
    const int ks = wave % S;
    const bool live = cb < ncb;
    int k_lo = 0, k_hi = 0;
    if (live) {
        const int per = (((nk + S - 1) / S) + 3) & ~3;          // whole scale groups per split
        k_lo = ks * per;
        k_hi = (k_lo + per < nk) ? k_lo + per : nk;
        if (k_lo > k_hi) k_lo = k_hi;
    }
    const int cbc = live ? cb : ncb - 1;
    const int wrow = SWIGLU ? cbc * 32 : cbc * 16;
    const unsigned char* wg = p.Wq + (long)(wrow + r) * p.ldq + q * 16;
    const unsigned char* wu = wg + (long)16 * p.ldq;
    const unsigned char* sg = p.Ws + (long)(wrow + r) * p.lds + q * 4;
    const unsigned char* su = sg + (long)16 * p.lds;

    // ---- issue order = the latency chain (as in gemv_body): (1) the activation chunks and norm weights of row 0, (2) the wave's first
    //      weight batch, (3) the reductions / quantisation, which wait only for (1) -- loads return in order: had the weights gone
    //      first, the prologue would have waited for HBM (measured: 23.5 us = 10 + 13.5, purely additive) ----
    const bool in_regs = (p.K >> 3) <= W4_CPT * nthr;
    u32x4_t xv[W4_CPT], gv[W4_CPT];
    if (in_regs) w4_load_row<W4_CPT>(p.A, p.norm_w, p.K, tid, nthr, xv, gv);
    u32x4_t wf[NACC][KU];
    unsigned sc[NACC][KU / 4];
    auto load_batch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int k = (k0 + u < k_hi) ? k0 + u : (k_hi > 0 ? k_hi - 1 : 0);
            wf[0][u] = w4_ld(wg + (long)k * 64);
            if (SWIGLU) wf[NACC - 1][u] = w4_ld(wu + (long)k * 64);
        }
#pragma unroll
        for (int g4 = 0; g4 < KU / 4; ++g4) {
            const int k = (k0 + 4 * g4 < k_hi) ? k0 + 4 * g4 : (k_hi > 0 ? ((k_hi - 1) & ~3) : 0);
            sc[0][g4] = *(const unsigned*)(sg + (long)(k >> 2) * 16);
            if (SWIGLU) sc[NACC - 1][g4] = *(const unsigned*)(su + (long)(k >> 2) * 16);
        }
    };
    if (k_lo < k_hi) load_batch(k_lo);

    // ---- prologue: the M activation rows -> (RMSNorm) -> FP8 with one scale per row, into LDS ----
    for (int m = 0; m < p.M; ++m) {
        const bf16_t* ar = p.A + (long)m * p.lda;
        unsigned char* dst = xq + (long)m * p.K;
        if (in_regs) w4_quantise_row<true, W4_CPT>(ar, p.norm_w, p.eps, p.K, dst, &sx[m], red, tid, nthr, xv, gv, m == 0);
        else w4_quantise_row<false, W4_CPT>(ar, p.norm_w, p.eps, p.K, dst, &sx[m], red, tid, nthr, xv, gv, false);
    }

    // ---- the weight stream ----
    f32x4_t acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[a] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // Operand layouts of the 16x16x128 instruction, probed on the hardware (tools/mxfp4_probe.py): an FP4 operand lane (row, q) holds the
    // 32 consecutive elements 32q .. 32q+31 in its 4 registers; an FP8 operand lane holds 16q .. 16q+15 in registers 0-3 and
    // 64+16q .. 64+16q+15 in registers 4-7 (two stacked K = 64 halves).  Lane q's block-scale byte applies to lane q's 32 FP4 elements.
    const unsigned char* xrow = xq + (long)(r < p.M ? r : 0) * p.K + q * 16;
    const bool xlive = r < p.M;
    for (int k0 = k_lo; k0 < k_hi; k0 += KU) {
        if (k0 != k_lo) load_batch(k0);
        // the byte selector of the block scale is an immediate of the instruction: the eight steps are written out
#define W4_STEP(U)                                                                                                                      \
        if (k0 + U < k_hi) {                                                                                                            \
            u32x4_t x0 = {0u, 0u, 0u, 0u}, x1 = {0u, 0u, 0u, 0u};                                                                        \
            if (xlive) {                                                                                                                \
                x0 = *(const u32x4_t*)(xrow + (long)(k0 + U) * 128);                                                                    \
                x1 = *(const u32x4_t*)(xrow + (long)(k0 + U) * 128 + 64);                                                               \
            }                                                                                                                           \
            const i32x8_t xb = {(int)x0[0], (int)x0[1], (int)x0[2], (int)x0[3], (int)x1[0], (int)x1[1], (int)x1[2], (int)x1[3]};        \
            const i32x8_t wa = {(int)wf[0][U][0], (int)wf[0][U][1], (int)wf[0][U][2], (int)wf[0][U][3], 0, 0, 0, 0};                    \
            /* A = weights (FP4: cbsz 4), B = activations (e4m3: blgp 0); block scale of A = byte U % 4 of the group's dword */          \
            acc[0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wa, xb, acc[0], 4, 0, U & 3, (int)sc[0][U >> 2], 0, 127);         \
            if (SWIGLU) {                                                                                                               \
                const i32x8_t wb = {(int)wf[NACC - 1][U][0], (int)wf[NACC - 1][U][1], (int)wf[NACC - 1][U][2], (int)wf[NACC - 1][U][3], 0, 0, 0, 0}; \
                acc[NACC - 1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wb, xb, acc[NACC - 1], 4, 0, U & 3, (int)sc[NACC - 1][U >> 2], 0, 127); \
            }                                                                                                                           \
        }
        W4_STEP(0) W4_STEP(1) W4_STEP(2) W4_STEP(3) W4_STEP(4) W4_STEP(5) W4_STEP(6) W4_STEP(7)
#undef W4_STEP
    }

    if (S > 1) {                                // the K splits of a column block meet in LDS; split 0 finishes
        if (ks > 0) {
#pragma unroll
            for (int a = 0; a < NACC; ++a) part[wave][a][lane] = acc[a];
        }
        __syncthreads();
        if (ks > 0 || !live) return;
        for (int o = 1; o < S; ++o)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = acc[a] + part[wave + o][a][lane];
    } else if (!live) {
        return;
    }

    // ---- epilogue: lane owns C[m = r][n .. n+3], n = cb*16 + 4*q; activation scale first, then gemm.hip's rounding points ----
    if (r >= p.M) return;
    const int n = cb * 16 + q * 4;
    const float sm = sx[r];
    float o[4];
    if (SWIGLU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float g = bfround(acc[0][e] * sm);
            const float uu = bfround(acc[NACC - 1][e] * sm);
            o[e] = bfround(silu_f(g)) * uu;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = acc[0][e] * sm;
        if (p.bias) {
            const u32x2_t bv = *(const u32x2_t*)(p.bias + n);
            o[0] += lo2f(bv[0]); o[1] += hi2f(bv[0]); o[2] += lo2f(bv[1]); o[3] += hi2f(bv[1]);
        }
        if (p.R) {
            const u32x2_t rv = *(const u32x2_t*)(p.R + (long)r * p.ldr + n);
            o[0] = bfround(o[0]) + lo2f(rv[0]); o[1] = bfround(o[1]) + hi2f(rv[0]);
            o[2] = bfround(o[2]) + lo2f(rv[1]); o[3] = bfround(o[3]) + hi2f(rv[1]);
        }
    }
    u32x2_t v = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
    *(u32x2_t*)(p.C + (long)r * p.ldc + n) = v;
}

extern "C" int bagel_gemv_w4_bf16(const void* A, int64_t lda, const void* Wq, int64_t ldq_bytes, const void* Ws, int64_t lds_bytes,
                                  const void* bias, const void* R, int64_t ldr, void* C, int64_t ldc, const void* norm_w, float eps,
                                  int32_t M, int32_t N, int32_t K, int32_t epilogue, hipStream_t stream) {
    BAGEL_REQUIRE(A && Wq && Ws && C, "gemv_w4: null pointer");
    BAGEL_REQUIRE(M >= 1 && M <= W4_MAX_M, "gemv_w4: M=%d not in [1,%d]", M, W4_MAX_M);
    BAGEL_REQUIRE(K > 0 && (K % 128) == 0 && (lda % 8) == 0, "gemv_w4: K %% 128 == 0 and 16-byte activation rows required");
    BAGEL_REQUIRE(ldq_bytes >= K / 2 && (ldq_bytes % 16) == 0 && lds_bytes >= ((K / 128 + 3) / 4) * 16 && (lds_bytes % 4) == 0, "gemv_w4: ldq/lds");
    BAGEL_REQUIRE(epilogue == EPI_NONE || epilogue == EPI_SWIGLU16, "gemv_w4: epilogue %d not in {none, swiglu16}", epilogue);
    BAGEL_REQUIRE(epilogue == EPI_SWIGLU16 ? ((N % 32) == 0 && !bias && !R) : (N % 16) == 0, "gemv_w4: N %% 16 (SwiGLU: N %% 32, no bias/residual)");
    BAGEL_REQUIRE((ldc % 4) == 0 && (ldr % 4) == 0, "gemv_w4: ldc/ldr must be multiples of 4");
    BAGEL_REQUIRE((((uintptr_t)A | (uintptr_t)Wq | (uintptr_t)norm_w) & 15) == 0 && (((uintptr_t)Ws) & 3) == 0 &&
                  (((uintptr_t)C | (uintptr_t)R | (uintptr_t)bias) & 7) == 0, "gemv_w4: alignment");
    W4Params p;
    p.A = (const bf16_t*)A; p.lda = lda; p.Wq = (const unsigned char*)Wq; p.ldq = ldq_bytes; p.Ws = (const unsigned char*)Ws; p.lds = lds_bytes;
    p.bias = (const bf16_t*)bias; p.R = (const bf16_t*)R; p.ldr = ldr; p.C = (bf16_t*)C; p.ldc = ldc;
    p.norm_w = (const bf16_t*)norm_w; p.eps = eps; p.M = M; p.N = N; p.K = K; p.epi = epilogue;
    const bool sw = epilogue == EPI_SWIGLU16;
    const int ncb = sw ? N / 32 : N / 16;
    const int nk = K / 128;
    // K splits: enough waves to cover the chip (~2 000) while a split keeps at least one scale group (4 steps)
    // K splits (waves sharing one column block): as few as give every wave ONE 8-step batch -- it is issued ahead of the prologue and
    // nothing else of the weight stream has to wait for a reduction -- at most 8 (the down projection: 148 steps = 3 batches per wave)
    static int force = -1;
    if (force < 0) {
        const char* e = getenv("BAGEL_W4_SPLIT");     // tuning knob; the default is what bench.py measures
        force = (e && atoi(e) > 0) ? atoi(e) : 0;
    }
    int S = 1;
    while (S < 8 && nk > 8 * S) S *= 2;
    if (force) S = force;
    while (S > 1 && nk / S < 1) S /= 2;
    p.split = S;
    const int nw = S > 4 ? S : 4;
    const int grid = ceil_div(ncb, nw / S);
    const size_t xq_bytes = ((size_t)M * K + 15) & ~(size_t)15;
    const size_t smem = xq_bytes + (S > 1 ? (size_t)nw * (sw ? 2 : 1) * 64 * sizeof(f32x4_t) : 0);
    BAGEL_REQUIRE(smem <= 150 * 1024, "gemv_w4: M*K = %d bytes of activations do not fit the LDS", M * K);
    const bool few = (K / 8) <= 2 * 64 * nw;            // the row fits 2 chunks per thread
#define W4_GO(SWV, CPTV)                                                                                                              \
    do {                                                                                                                              \
        if (smem > 48 * 1024)                                                                                                         \
            if (int rc = bagel_enable_lds((const void*)gemv_w4_kernel<SWV, CPTV>, 150 * 1024, "gemv_w4_kernel")) return rc;          \
        hipLaunchKernelGGL((gemv_w4_kernel<SWV, CPTV>), dim3(grid), dim3(64 * nw), smem, stream, p);                                   \
    } while (0)
    if (sw) { if (few) W4_GO(true, 2); else W4_GO(true, 5); }
    else { if (few) W4_GO(false, 2); else W4_GO(false, 5); }
#undef W4_GO
    return bagel_check_launch("gemv_w4_kernel");
}
