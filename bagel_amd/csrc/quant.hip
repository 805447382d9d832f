// Weight-only INT8 for the autoregressive decode (the MI355X analogue of the reference's quantised inference modes,
// app.py:114-131: bitsandbytes NF4 / LLM.int8 -- an un-vendored dependency whose kernels are CUDA-only).
//
// Decode at batch 1 is a pure weight stream (decode.hip), so halving the bytes per weight is the one lever left on its
// roofline: W[n, :] is stored as  u8 = round(W / s_n) + 128,  s_n = max|W[n, :]| / 127  (row-wise absmax, the scheme of
// LLM.int8's weight side) and de-quantised on the fly:  y_n = s_n * (sum_k u8[n,k] x_k  -  128 * sum_k x_k), fp32
// accumulation, activations stay bf16 ("W8A16").  One v_cvt_f32_ubyteN + one FMA per weight: 2 VALU ops per byte, well
// under the HBM time.  This CHANGES results (~0.3 % relative error per weight) and is therefore an option the caller
// selects, like the reference's modes; the bf16 path is the default and the one the parity tests pin.
//
//   bagel_quantize_rows_i8   bf16 [N, K] -> u8 [N, K] + fp32 scale [N]
//   bagel_gemv_w8_bf16       C[M <= 4.., N] = A (dequant W)^T with the epilogues / fused RMSNorm of bagel_gemv_bf16
#include "common.h"
#include <stdlib.h>

#define EPI_NONE 0
#define EPI_GELU_TANH 1
#define EPI_SILU 2
#define EPI_SWIGLU16 3

__global__ __launch_bounds__(256) void quantize_rows_i8_kernel(const bf16_t* __restrict__ w, long ldw, unsigned char* __restrict__ q,
                                                               long ldq, float* __restrict__ scale, int rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16_t* wr = w + (long)row * ldw;
    float amax = 0.f;
    for (int c = lane; c < cols; c += 64) amax = fmaxf(amax, fabsf(bf2f(wr[c])));
    amax = wave_max(amax);
    const float s = amax > 0.f ? amax / 127.0f : 1.0f;
    if (lane == 0) scale[row] = s;
    unsigned char* qr = q + (long)row * ldq;
    for (int c = lane; c < cols; c += 64) {
        float v = rintf(__fdiv_rn(bf2f(wr[c]), s));          // round to nearest even, like torch.round
        v = fminf(fmaxf(v, -127.f), 127.f);
        qr[c] = (unsigned char)((int)v + 128);
    }
}

extern "C" int bagel_quantize_rows_i8(const void* w, int64_t ldw, void* q, int64_t ldq, float* scale, int32_t rows, int32_t cols,
                                      hipStream_t stream) {
    BAGEL_REQUIRE(w && q && scale, "quantize_rows_i8: null pointer");
    if (rows <= 0 || cols <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(quantize_rows_i8_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, stream, (const bf16_t*)w, (long)ldw,
                       (unsigned char*)q, (long)ldq, scale, rows, cols);
    return bagel_check_launch("quantize_rows_i8_kernel");
}

struct GemvW8Params {
    const bf16_t* A; long lda;
    const unsigned char* W; long ldw;      // u8 [N, K]
    const float* scale;                    // [N]
    const bf16_t* bias;
    const bf16_t* R; long ldr;
    bf16_t* C; long ldc;
    const bf16_t* norm_w; float eps;
    int M, N, K, epi;
};

// Same skeleton as gemv_body<MR, 1> of decode.hip: a wave owns a pair of weight rows, activations (optionally RMS-normalised)
// staged once per workgroup in LDS; here a 16-byte lane chunk carries 16 weights, so a row of K = 3584 is 3.5 chunk groups.
template <int MR>
__global__ __launch_bounds__(256) void gemv_w8_kernel(GemvW8Params p) {
    constexpr int U = 4;                    // chunk groups (1 KB of weights per row each) in flight per row
    extern __shared__ __attribute__((aligned(16))) unsigned char w8_smem[];
    bf16_t* xs = (bf16_t*)w8_smem;          // [MR][K]
    __shared__ float red[MR][4], xsum[MR][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = p.K, nch8 = K >> 3, nch = K >> 4;       // 8-element bf16 chunks (staging), 16-weight chunks (streaming)
    const int NP = p.N >> 1;
    const bool swiglu = p.epi == EPI_SWIGLU16;
    const int ngr = (nch + 63) >> 6;
    const int pp = blockIdx.x * 4 + wave;
    const bool live = pp < NP;
    const int r0 = swiglu ? ((pp >> 4) << 5) + (pp & 15) : 2 * pp;
    const int r1 = swiglu ? r0 + 16 : r0 + 1;

    // first weight batch before the staging (see decode.hip); activations first in program order
    u32x4_t xr[MR][2], gw[2];
    const bool small = nch8 <= 512;
    if (small) {
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const bf16_t* ar = p.A + (long)(m < p.M ? m : p.M - 1) * p.lda;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = tid + 256 * i;
                xr[m][i] = c < nch8 ? *(const u32x4_t*)(ar + (long)c * 8) : u32x4_t{0u, 0u, 0u, 0u};
            }
        }
        if (p.norm_w) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = tid + 256 * i;
                gw[i] = c < nch8 ? *(const u32x4_t*)(p.norm_w + (long)c * 8) : u32x4_t{0u, 0u, 0u, 0u};
            }
        }
    }
    u32x4_t wa[U], wb[U];
    auto load_batch = [&](int g) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ch = (g + u) * 64 + lane;
            const long off = (long)(ch < nch ? ch : 0) * 16;
            wa[u] = ld_stream<u32x4_t>(p.W + (long)r0 * p.ldw + off);
            wb[u] = ld_stream<u32x4_t>(p.W + (long)r1 * p.ldw + off);
        }
    };
    if (live) load_batch(0);

    // ---- staging: RMSNorm (optional), bf16 rows into LDS, and sum_k x_k per row (for the +128 offset of the u8 weights) ----
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
                for (int c = tid; c < nch8; c += 256) {
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
        float sx = 0.f;
        auto put = [&](int c, u32x4_t v, const u32x4_t g) {
            if (p.norm_w) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v[e] = pack2bf(bfround(lo2f(v[e]) * inv) * lo2f(g[e]), bfround(hi2f(v[e]) * inv) * hi2f(g[e]));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) sx += lo2f(v[e]) + hi2f(v[e]);
            *(u32x4_t*)(xs + (long)m * K + (long)c * 8) = v;
        };
        if (small) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = tid + 256 * i;
                if (c < nch8) put(c, xr[m][i], gw[i]);
            }
        } else {
            for (int c = tid; c < nch8; c += 256) {
                const u32x4_t g = p.norm_w ? *(const u32x4_t*)(p.norm_w + (long)c * 8) : u32x4_t{0u, 0u, 0u, 0u};
                put(c, *(const u32x4_t*)(ar + (long)c * 8), g);
            }
        }
        sx = wave_sum(sx);
        if (lane == 0) xsum[m][wave] = sx;
    }
    __syncthreads();
    if (!live) return;

    float a0[MR], a1[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) a0[m] = a1[m] = 0.f;
    for (int g = 0; g < ngr; g += U) {
        if (g != 0) load_batch(g);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ch = (g + u) * 64 + lane;
            const bool ok = ch < nch;
            const long xo = (long)(ok ? ch : 0) * 16;
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                u32x4_t x0 = *(const u32x4_t*)(xs + (long)m * K + xo);
                u32x4_t x1 = *(const u32x4_t*)(xs + (long)m * K + xo + 8);
                if (!ok) { x0 = u32x4_t{0u, 0u, 0u, 0u}; x1 = x0; }
#pragma unroll
                for (int d = 0; d < 4; ++d) {                       // dword d of the weight chunk = weights 4d .. 4d+3
                    const unsigned xa = d < 2 ? x0[2 * d] : x1[2 * d - 4], xb = d < 2 ? x0[2 * d + 1] : x1[2 * d - 3];
                    const float f0 = lo2f(xa), f1 = hi2f(xa), f2 = lo2f(xb), f3 = hi2f(xb);
                    const unsigned ua = wa[u][d], ub = wb[u][d];
                    a0[m] = fmaf((float)(ua & 0xffu), f0, a0[m]);
                    a0[m] = fmaf((float)((ua >> 8) & 0xffu), f1, a0[m]);
                    a0[m] = fmaf((float)((ua >> 16) & 0xffu), f2, a0[m]);
                    a0[m] = fmaf((float)(ua >> 24), f3, a0[m]);
                    a1[m] = fmaf((float)(ub & 0xffu), f0, a1[m]);
                    a1[m] = fmaf((float)((ub >> 8) & 0xffu), f1, a1[m]);
                    a1[m] = fmaf((float)((ub >> 16) & 0xffu), f2, a1[m]);
                    a1[m] = fmaf((float)(ub >> 24), f3, a1[m]);
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        const float t0 = wave_sum(a0[m]), t1 = wave_sum(a1[m]);
        if (lane == 0 && m < p.M) {
            const float off = 128.0f * ((xsum[m][0] + xsum[m][1]) + (xsum[m][2] + xsum[m][3]));
            const float s0 = p.scale[r0] * (t0 - off), s1 = p.scale[r1] * (t1 - off);
            if (swiglu) {
                const float gg = bfround(s0), uu = bfround(s1);
                p.C[(long)m * p.ldc + pp] = f2bf(bfround(silu_f(gg)) * uu);
            } else {
                float o[2] = {s0, s1};
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int n = t ? r1 : r0;
                    if (p.bias) o[t] += bf2f(p.bias[n]);
                    if (p.epi == EPI_GELU_TANH) o[t] = gelu_tanh_f(bfround(o[t]));
                    else if (p.epi == EPI_SILU) o[t] = silu_f(bfround(o[t]));
                    if (p.R) o[t] = bfround(o[t]) + bf2f(p.R[(long)m * p.ldr + n]);
                }
                *(unsigned*)(p.C + (long)m * p.ldc + r0) = pack2bf(o[0], o[1]);
            }
        }
    }
}

#define W8_MAX_LDS (144 * 1024)

template <int MR>
static int launch_gemv_w8(const GemvW8Params& p, hipStream_t stream) {
    const size_t smem = (size_t)MR * p.K * sizeof(bf16_t);
    if (smem > 48 * 1024)
        if (int rc = bagel_enable_lds((const void*)gemv_w8_kernel<MR>, (int)W8_MAX_LDS, "gemv_w8_kernel")) return rc;
    hipLaunchKernelGGL((gemv_w8_kernel<MR>), dim3(ceil_div(p.N / 2, 4)), dim3(256), smem, stream, p);
    return bagel_check_launch("gemv_w8_kernel");
}

extern "C" int bagel_gemv_w8_bf16(const void* A, int64_t lda, const void* Wq, int64_t ldw, const float* scale, const void* bias,
                                  const void* R, int64_t ldr, void* C, int64_t ldc, const void* norm_w, float eps, int32_t M,
                                  int32_t N, int32_t K, int32_t epilogue, hipStream_t stream) {
    BAGEL_REQUIRE(A && Wq && scale && C, "gemv_w8: null pointer");
    BAGEL_REQUIRE(K > 0 && (K % 16) == 0 && (lda % 8) == 0 && (ldw % 16) == 0, "gemv_w8: K and ldw must be multiples of 16, lda of 8");
    BAGEL_REQUIRE(N > 0 && (N % 2) == 0, "gemv_w8: N=%d must be even", N);
    BAGEL_REQUIRE(epilogue >= 0 && epilogue <= 3, "gemv_w8: unknown epilogue %d", epilogue);
    BAGEL_REQUIRE(epilogue != EPI_SWIGLU16 || ((N % 32) == 0 && !bias && !R), "gemv_w8: swiglu needs N%%32==0, no bias/residual");
    BAGEL_REQUIRE((((uintptr_t)A | (uintptr_t)Wq | (uintptr_t)norm_w) & 15) == 0, "gemv_w8: A/W/norm_w must be 16-byte aligned");
    BAGEL_REQUIRE((ldc % 2) == 0 && (ldr % 2) == 0 && (((uintptr_t)C | (uintptr_t)R) & 3) == 0, "gemv_w8: C/R rows must be 4-byte aligned");
    BAGEL_REQUIRE((size_t)K * sizeof(bf16_t) <= W8_MAX_LDS, "gemv_w8: K=%d does not fit the LDS staging buffer", K);
    if (M <= 0) return BAGEL_OK;
    int m0 = 0;
    while (m0 < M) {
        int mr = (M - m0 >= 4) ? 4 : (M - m0 >= 2 ? 2 : 1);
        while (mr > 1 && (size_t)mr * K * sizeof(bf16_t) > W8_MAX_LDS) mr >>= 1;
        GemvW8Params p;
        p.A = (const bf16_t*)A + (long)m0 * lda; p.lda = lda;
        p.W = (const unsigned char*)Wq; p.ldw = ldw; p.scale = scale;
        p.bias = (const bf16_t*)bias;
        p.R = R ? (const bf16_t*)R + (long)m0 * ldr : nullptr; p.ldr = ldr;
        p.C = (bf16_t*)C + (long)m0 * ldc; p.ldc = ldc;
        p.norm_w = (const bf16_t*)norm_w; p.eps = eps;
        p.M = mr; p.N = N; p.K = K; p.epi = epilogue;
        int rc = mr == 4 ? launch_gemv_w8<4>(p, stream) : (mr == 2 ? launch_gemv_w8<2>(p, stream) : launch_gemv_w8<1>(p, stream));
        if (rc != BAGEL_OK) return rc;
        m0 += mr;
    }
    return BAGEL_OK;
}

// ---------------------------------------------------------------------------------------------------------
// FP8 (OCP e4m3) row quantisation for the gen-expert GEMMs (bagel_gemm_fp8_bf16, gemm.hip): the MI355X-native counterpart of the
// reference's quantised load modes (app.py:114-131) on the DENOISE side, where the time goes.
//   q[r, k] = e4m3(x[r, k] / s_r)  (round to nearest even, v_cvt_pk_fp8_f32),   s_r = max_k |x[r, k]| / 448   (448 = largest e4m3)
// Used once per weight matrix at pack time and once per activation matrix per GEMM input.  One wave per row, 16-byte lanes,
// two passes over the row (the second one hits L2): HBM-bound, 3 bytes moved per element.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void quantize_rows_fp8_kernel(const bf16_t* __restrict__ x, long ldx, unsigned char* __restrict__ q,
                                                                long ldq, float* __restrict__ scale, int rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16_t* xr = x + (long)row * ldx;
    const int nch = cols >> 3;                       // 8-element chunks (16 bytes in, 8 bytes out)
    float amax = 0.f;
    for (int c = lane; c < nch; c += 64) {
        const u32x4_t v = *(const u32x4_t*)(xr + 8 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fmaxf(fabsf(lo2f(v[e])), fabsf(hi2f(v[e]))));
    }
    amax = wave_max(amax);
    const float s = amax > 0.f ? amax / 448.0f : 1.0f;
    const float inv = 1.0f / s;
    if (lane == 0) scale[row] = s;
    unsigned char* qr = q + (long)row * ldq;
    for (int c = lane; c < nch; c += 64) {
        const u32x4_t v = *(const u32x4_t*)(xr + 8 * c);
        u32x2_t o;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int w = 0;
            w = __builtin_amdgcn_cvt_pk_fp8_f32(lo2f(v[2 * h]) * inv, hi2f(v[2 * h]) * inv, w, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32(lo2f(v[2 * h + 1]) * inv, hi2f(v[2 * h + 1]) * inv, w, true);
            o[h] = (unsigned)w;
        }
        *(u32x2_t*)(qr + 8 * c) = o;
    }
}

extern "C" int bagel_quantize_rows_fp8(const void* x, int64_t ldx, void* q, int64_t ldq_bytes, float* scale, int32_t rows, int32_t cols,
                                       hipStream_t stream) {
    BAGEL_REQUIRE(x && q && scale, "quantize_rows_fp8: null pointer");
    BAGEL_REQUIRE((cols % 8) == 0 && (ldx % 8) == 0 && (ldq_bytes % 8) == 0, "quantize_rows_fp8: cols / leading dims must be multiples of 8");
    if (rows <= 0 || cols <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(quantize_rows_fp8_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, stream, (const bf16_t*)x, (long)ldx,
                       (unsigned char*)q, (long)ldq_bytes, scale, rows, cols);
    return bagel_check_launch("quantize_rows_fp8_kernel");
}
