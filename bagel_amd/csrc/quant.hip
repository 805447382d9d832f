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

// The same arithmetic with the WHOLE ROW in registers (rows of <= 64 * MAXC chunks): one wave per row issues all of its 16-byte loads before the first
// use, so a row is read from HBM ONCE with MAXC loads in flight per lane, instead of twice (the second pass from L2) with one load in flight -- the
// two-pass kernel ran at 3.1-4.1 TB/s of its own 3 bytes per element on the activation matrices of the FP8 denoise path.  Bit-identical.
template <int MAXC>
__global__ __launch_bounds__(256) void quantize_rows_fp8_reg_kernel(const bf16_t* __restrict__ x, long ldx, unsigned char* __restrict__ q,
                                                                    long ldq, float* __restrict__ scale, int rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16_t* xr = x + (long)row * ldx;
    const int nch = cols >> 3;
    u32x4_t v[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < nch ? *(const u32x4_t*)(xr + 8 * c) : u32x4_t{0u, 0u, 0u, 0u};
    }
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fmaxf(fabsf(lo2f(v[i][e])), fabsf(hi2f(v[i][e]))));
    amax = wave_max(amax);
    const float s = amax > 0.f ? amax / 448.0f : 1.0f;
    const float inv = 1.0f / s;
    if (lane == 0) scale[row] = s;
    unsigned char* qr = q + (long)row * ldq;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            u32x2_t o;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int w = 0;
                w = __builtin_amdgcn_cvt_pk_fp8_f32(lo2f(v[i][2 * h]) * inv, hi2f(v[i][2 * h]) * inv, w, false);
                w = __builtin_amdgcn_cvt_pk_fp8_f32(lo2f(v[i][2 * h + 1]) * inv, hi2f(v[i][2 * h + 1]) * inv, w, true);
                o[h] = (unsigned)w;
            }
            *(u32x2_t*)(qr + 8 * c) = o;
        }
    }
}

// Delayed scaling of the FP8 gen expert's SwiGLU output (bagel_gemm_fp8_swiglu_q8): turn the row maxima one denoise step collected into the scales the next
// step quantises with -- scale = margin * amax / 448 (margin 2: one binade of headroom for a row that grows between two steps; 1.0 where nothing was seen) --
// and clear the maxima for the step that follows.  `rows`: the physical rows in use (NULL = 0..n-1).
__global__ __launch_bounds__(256) void fp8_delayed_scales_kernel(unsigned* __restrict__ amax, float* __restrict__ scale, const int* __restrict__ rows, int n, float k) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int r = rows ? rows[i] : i;
    const float a = __uint_as_float(amax[r]);
    scale[r] = a > 0.f ? a * k : 1.0f;
    amax[r] = 0u;
}
extern "C" int bagel_fp8_delayed_scales(void* amax, float* scale, const int32_t* rows, int32_t n, float margin, hipStream_t stream) {
    BAGEL_REQUIRE(amax && scale, "fp8_delayed_scales: null pointer");
    BAGEL_REQUIRE(margin >= 1.0f, "fp8_delayed_scales: margin %g < 1 would clip the values the scale was measured on", (double)margin);
    if (n <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(fp8_delayed_scales_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, stream, (unsigned*)amax, scale, (const int*)rows, n, margin / 448.0f);
    return bagel_check_launch("fp8_delayed_scales_kernel");
}

extern "C" int bagel_quantize_rows_fp8(const void* x, int64_t ldx, void* q, int64_t ldq_bytes, float* scale, int32_t rows, int32_t cols,
                                       hipStream_t stream) {
    BAGEL_REQUIRE(x && q && scale, "quantize_rows_fp8: null pointer");
    BAGEL_REQUIRE((cols % 8) == 0 && (ldx % 8) == 0 && (ldq_bytes % 8) == 0, "quantize_rows_fp8: cols / leading dims must be multiples of 8");
    if (rows <= 0 || cols <= 0) return BAGEL_OK;
    // A/B knob, read once per process (thread-safe static): 1 = the two-pass kernel for every row length
    static const int two_pass = [] { const char* e = getenv("BAGEL_FP8_QUANT_TWO_PASS"); return (e && atoi(e) > 0) ? 1 : 0; }();
    const int nch = cols >> 3;
    if (!two_pass && nch <= 64 * 8)
        hipLaunchKernelGGL(quantize_rows_fp8_reg_kernel<8>, dim3(ceil_div(rows, 4)), dim3(256), 0, stream, (const bf16_t*)x, (long)ldx,
                           (unsigned char*)q, (long)ldq_bytes, scale, rows, cols);
    else if (!two_pass && nch <= 64 * 40)
        hipLaunchKernelGGL(quantize_rows_fp8_reg_kernel<40>, dim3(ceil_div(rows, 4)), dim3(256), 0, stream, (const bf16_t*)x, (long)ldx,
                           (unsigned char*)q, (long)ldq_bytes, scale, rows, cols);
    else
        hipLaunchKernelGGL(quantize_rows_fp8_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, stream, (const bf16_t*)x, (long)ldx,
                           (unsigned char*)q, (long)ldq_bytes, scale, rows, cols);
    return bagel_check_launch("quantize_rows_fp8_kernel");
}

// ---------------------------------------------------------------------------------------------------------
// NF4 -- the 4-bit load mode the reference ships (app.py:114-125: bitsandbytes, quant_type "nf4", blocksize 64, fp32 absmax, no double
// quantisation, bf16 compute), restated in oracle/nf4.py from the library's published algorithm:
//   code book NF4_CODE[16], blocks of 64 consecutive weights with an fp32 absmax, x = w * (1 / absmax), code = number of midpoint
//   thresholds below x, two codes per byte with the EVEN element in the HIGH nibble, w' = NF4_CODE[code] * absmax.
//   bagel_quantize_nf4     bf16 [N, K] (K % 64 == 0) -> u8 [N, K / 2] + fp32 absmax [N, K / 64]      (bit-exact vs the restatement)
//   bagel_gemv_nf4_bf16    C[M <= 4.., N] = A (dequant W)^T with the epilogues / fused RMSNorm of bagel_gemv_bf16 ("W4A16")
// The projection keeps w' in fp32 (code * absmax is not rounded to bf16 before the multiply, which the library's de-quantise-then-matmul
// path does): a relative 2^-9 per weight with random sign, far below the bf16 rounding of the output it is tested against.
// A 16-byte lane chunk carries 32 weights = half a block; the code book lives in LDS (16 floats: 64 lanes hit 16 different banks, equal
// indices broadcast), one ds_read_b32 per weight beside ~2.75 VALU (index extraction, activation unpack, FMA) -- VALU-bound like the INT8
// option, at a quarter of the bf16 weight bytes.
// ---------------------------------------------------------------------------------------------------------
__constant__ float NF4_CODE_DEV[16] = {-1.0f, -0.6961928009986877f, -0.5250730514526367f, -0.39491748809814453f, -0.28444138169288635f,
                                       -0.18477343022823334f, -0.09105003625154495f, 0.0f, 0.07958029955625534f, 0.16093020141124725f,
                                       0.24611230194568634f, 0.33791524171829224f, 0.44070982933044434f, 0.5626170039176941f,
                                       0.7229568362236023f, 1.0f};
__constant__ float NF4_THRESH_DEV[15] = {-0.8480964004993439f, -0.6106329262256622f, -0.4599952697753906f, -0.33967943489551544f,
                                         -0.23460740596055984f, -0.13791173323988914f, -0.045525018125772476f, 0.03979014977812767f,
                                         0.1202552504837513f, 0.2035212516784668f, 0.2920137718319893f, 0.3893125355243683f,
                                         0.5016634166240692f, 0.6427869200706482f, 0.8614784181118011f};

// one thread per 64-weight block
__global__ __launch_bounds__(256) void quantize_nf4_kernel(const bf16_t* __restrict__ w, long ldw, unsigned char* __restrict__ q, long ldq,
                                                           float* __restrict__ absmax, int rows, int nblk) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)rows * nblk) return;
    const int row = (int)(i / nblk), b = (int)(i % nblk);
    const bf16_t* src = w + (long)row * ldw + (long)b * 64;
    u32x4_t v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = *(const u32x4_t*)(src + 8 * c);
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fmaxf(fabsf(lo2f(v[c][e])), fabsf(hi2f(v[c][e]))));
    absmax[i] = amax;
    const float inv = __fdiv_rn(1.0f, amax);           // inf for an all-zero block: x = NaN below, every comparison false, code 0
    auto code = [&](float x) {
        unsigned c = 0;
#pragma unroll
        for (int t = 0; t < 15; ++t) c += (x * inv > NF4_THRESH_DEV[t]) ? 1u : 0u;
        return c;
    };
    unsigned char* dst = q + (long)row * ldq + (long)b * 32;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        unsigned out = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) out |= ((code(lo2f(v[c][e])) << 4) | code(hi2f(v[c][e]))) << (8 * e);   // even element -> high nibble
        *(unsigned*)(dst + 4 * c) = out;
    }
}

extern "C" int bagel_quantize_nf4(const void* w, int64_t ldw, void* q, int64_t ldq_bytes, float* absmax, int32_t rows, int32_t cols,
                                  hipStream_t stream) {
    BAGEL_REQUIRE(w && q && absmax, "quantize_nf4: null pointer");
    BAGEL_REQUIRE(cols > 0 && (cols % 64) == 0 && (ldw % 8) == 0 && (ldq_bytes % 4) == 0, "quantize_nf4: cols must be a multiple of the 64-weight block (ldw % 8, ldq % 4)");
    BAGEL_REQUIRE((((uintptr_t)w) & 15) == 0 && (((uintptr_t)q) & 3) == 0, "quantize_nf4: w must be 16-byte, q 4-byte aligned");
    if (rows <= 0) return BAGEL_OK;
    const int nblk = cols / 64;
    hipLaunchKernelGGL(quantize_nf4_kernel, dim3(ceil_div((long)rows * nblk, 256)), dim3(256), 0, stream, (const bf16_t*)w, (long)ldw,
                       (unsigned char*)q, (long)ldq_bytes, absmax, rows, nblk);
    return bagel_check_launch("quantize_nf4_kernel");
}

// De-quantise NF4 codes to bf16 weights: w = bf16(code_book[code] * absmax[block]) -- what bitsandbytes' matmul_4bit does in front of F.linear whenever
// more than one activation row is multiplied (app.py:114-125's model at prefill / denoise sizes): the WHOLE-MODEL 4-bit mode of the engines
// (MoTEngine weight_store="nf4") keeps the codes resident and materialises one layer's bf16 matrices at a time with this kernel.
// One thread per 16 codes (8 bytes in, 32 bytes out).
__global__ __launch_bounds__(256) void dequantize_nf4_kernel(const unsigned char* __restrict__ q, long ldq, const float* __restrict__ absmax,
                                                             bf16_t* __restrict__ out, long ldo, int rows, int cols) {
    const int per_row = cols >> 4;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)rows * per_row) return;
    const int row = (int)(i / per_row), c16 = (int)(i % per_row);
    const float am = absmax[(long)row * (cols >> 6) + (c16 >> 2)];
    const u32x2_t v = *(const u32x2_t*)(q + (long)row * ldq + (long)c16 * 8);
    unsigned o[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const unsigned byte = (v[b >> 2] >> (8 * (b & 3))) & 0xffu;
        o[b] = pack2bf(NF4_CODE_DEV[byte >> 4] * am, NF4_CODE_DEV[byte & 15u] * am);          // even element in the high nibble
    }
    bf16_t* dst = out + (long)row * ldo + (long)c16 * 16;
    *(u32x4_t*)dst = (u32x4_t){o[0], o[1], o[2], o[3]};
    *(u32x4_t*)(dst + 8) = (u32x4_t){o[4], o[5], o[6], o[7]};
}

extern "C" int bagel_dequantize_nf4_bf16(const void* q, int64_t ldq_bytes, const float* absmax, void* out, int64_t ld_out, int32_t rows,
                                         int32_t cols, hipStream_t stream) {
    BAGEL_REQUIRE(q && absmax && out, "dequantize_nf4: null pointer");
    BAGEL_REQUIRE(cols > 0 && (cols % 64) == 0 && (ldq_bytes % 8) == 0 && (ld_out % 8) == 0, "dequantize_nf4: cols %% 64, ldq %% 8 bytes, ld_out %% 8");
    BAGEL_REQUIRE((((uintptr_t)q) & 7) == 0 && (((uintptr_t)out) & 15) == 0, "dequantize_nf4: q must be 8-byte, out 16-byte aligned");
    if (rows <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(dequantize_nf4_kernel, dim3(ceil_div((long)rows * (cols / 16), 256)), dim3(256), 0, stream, (const unsigned char*)q,
                       (long)ldq_bytes, absmax, (bf16_t*)out, (long)ld_out, rows, cols);
    return bagel_check_launch("dequantize_nf4_kernel");
}

// De-quantise row-wise absmax INT8 (bagel_quantize_rows_i8) to bf16: w = bf16((q - 128) * scale[row]); one thread per 8 weights.
__global__ __launch_bounds__(256) void dequantize_rows_i8_kernel(const unsigned char* __restrict__ q, long ldq, const float* __restrict__ scale,
                                                                 bf16_t* __restrict__ out, long ldo, int rows, int cols) {
    const int per_row = cols >> 3;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)rows * per_row) return;
    const int row = (int)(i / per_row), c8 = (int)(i % per_row);
    const float sc = scale[row];
    const u32x2_t v = *(const u32x2_t*)(q + (long)row * ldq + (long)c8 * 8);
    unsigned o[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int lo = (int)((v[b >> 1] >> (16 * (b & 1))) & 0xffu) - 128, hi = (int)((v[b >> 1] >> (16 * (b & 1) + 8)) & 0xffu) - 128;
        o[b] = pack2bf((float)lo * sc, (float)hi * sc);
    }
    *(u32x4_t*)(out + (long)row * ldo + (long)c8 * 8) = (u32x4_t){o[0], o[1], o[2], o[3]};
}

extern "C" int bagel_dequantize_rows_i8_bf16(const void* q, int64_t ldq, const float* scale, void* out, int64_t ld_out, int32_t rows, int32_t cols,
                                             hipStream_t stream) {
    BAGEL_REQUIRE(q && scale && out, "dequantize_rows_i8: null pointer");
    BAGEL_REQUIRE(cols > 0 && (cols % 8) == 0 && (ldq % 8) == 0 && (ld_out % 8) == 0, "dequantize_rows_i8: cols %% 8, ldq %% 8, ld_out %% 8");
    BAGEL_REQUIRE((((uintptr_t)q) & 7) == 0 && (((uintptr_t)out) & 15) == 0, "dequantize_rows_i8: q must be 8-byte, out 16-byte aligned");
    if (rows <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(dequantize_rows_i8_kernel, dim3(ceil_div((long)rows * (cols / 8), 256)), dim3(256), 0, stream, (const unsigned char*)q, (long)ldq,
                       scale, (bf16_t*)out, (long)ld_out, rows, cols);
    return bagel_check_launch("dequantize_rows_i8_kernel");
}

struct GemvNf4Params {
    const bf16_t* A; long lda;
    const unsigned char* W; long ldw;      // packed codes [N, K / 2]
    const float* absmax;                   // [N, K / 64]
    const bf16_t* bias;
    const bf16_t* R; long ldr;
    bf16_t* C; long ldc;
    const bf16_t* norm_w; float eps;
    int M, N, K, epi;
};

// Skeleton of gemv_w8_kernel: a wave owns a pair of weight rows (SwiGLU16: a gate row and its up row), the (optionally RMS-normalised)
// activations are staged once per workgroup in LDS; a 16-byte lane chunk = 32 weights of each row.
template <int MR>
__global__ __launch_bounds__(256) void gemv_nf4_kernel(GemvNf4Params p) {
    constexpr int U = 2;                    // chunk groups (1 KB of codes = 2048 weights per row each) in flight per row
    extern __shared__ __attribute__((aligned(16))) unsigned char nf4_smem[];
    bf16_t* xs = (bf16_t*)nf4_smem;         // [MR][K]
    __shared__ float red[MR][4];
    __shared__ __attribute__((aligned(64))) float lut[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = p.K, nch8 = K >> 3, nch = K >> 5;       // 8-element bf16 chunks (staging), 32-weight chunks (streaming)
    const int nb = K >> 6;                                // absmax entries per row
    const int NP = p.N >> 1;
    const bool swiglu = p.epi == EPI_SWIGLU16;
    const int ngr = (nch + 63) >> 6;
    const int pp = blockIdx.x * 4 + wave;
    const bool live = pp < NP;
    const int r0 = swiglu ? ((pp >> 4) << 5) + (pp & 15) : 2 * pp;
    const int r1 = swiglu ? r0 + 16 : r0 + 1;
    if (tid < 16) lut[tid] = NF4_CODE_DEV[tid];

    u32x4_t wa[U], wb[U];
    float sa[U], sb[U];
    auto load_batch = [&](int g) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ch = (g + u) * 64 + lane;
            const int cc = ch < nch ? ch : 0;
            wa[u] = ld_stream<u32x4_t>(p.W + (long)r0 * p.ldw + (long)cc * 16);
            wb[u] = ld_stream<u32x4_t>(p.W + (long)r1 * p.ldw + (long)cc * 16);
            sa[u] = p.absmax[(long)r0 * nb + (cc >> 1)];
            sb[u] = p.absmax[(long)r1 * nb + (cc >> 1)];
        }
    };
    if (live) load_batch(0);

    // ---- staging: RMSNorm (optional) and the bf16 activation rows into LDS ----
    if (p.norm_w) {
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const bf16_t* ar = p.A + (long)(m < p.M ? m : p.M - 1) * p.lda;
            float ss = 0.f;
            for (int c = tid; c < nch8; c += 256) {
                const u32x4_t v = *(const u32x4_t*)(ar + (long)c * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = lo2f(v[e]), b = hi2f(v[e]);
                    ss += a * a + b * b;
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
        for (int c = tid; c < nch8; c += 256) {
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
    __syncthreads();
    if (!live) return;

    float a0[MR], a1[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) a0[m] = a1[m] = 0.f;
    const char* lutb = (const char*)lut;
    for (int g = 0; g < ngr; g += U) {
        if (g != 0) load_batch(g);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ch = (g + u) * 64 + lane;
            const bool ok = ch < nch;
            const long xo = (long)(ok ? ch : 0) * 32;
            float s0[MR], s1[MR];
#pragma unroll
            for (int m = 0; m < MR; ++m) s0[m] = s1[m] = 0.f;
#pragma unroll
            for (int d = 0; d < 4; ++d) {                       // dword d of the code chunk = weights 8d .. 8d+7
                const unsigned ua = wa[u][d], ub = wb[u][d];
                const unsigned ea = (ua >> 2) & 0x3C3C3C3Cu, oa = (ua << 2) & 0x3C3C3C3Cu;     // 4 * code of the even / odd elements, one per byte
                const unsigned eb = (ub >> 2) & 0x3C3C3C3Cu, ob = (ub << 2) & 0x3C3C3C3Cu;
                u32x4_t xq[MR];
#pragma unroll
                for (int m = 0; m < MR; ++m) xq[m] = *(const u32x4_t*)(xs + (long)m * K + xo + 8 * d);
#pragma unroll
                for (int b = 0; b < 4; ++b) {                   // byte b = elements 8d + 2b (high nibble) and 8d + 2b + 1 (low nibble)
                    const float ca0 = *(const float*)(lutb + ((ea >> (8 * b)) & 0xFFu)), ca1 = *(const float*)(lutb + ((oa >> (8 * b)) & 0xFFu));
                    const float cb0 = *(const float*)(lutb + ((eb >> (8 * b)) & 0xFFu)), cb1 = *(const float*)(lutb + ((ob >> (8 * b)) & 0xFFu));
#pragma unroll
                    for (int m = 0; m < MR; ++m) {
                        const float f0 = lo2f(xq[m][b]), f1 = hi2f(xq[m][b]);
                        s0[m] = fmaf(ca0, f0, s0[m]); s0[m] = fmaf(ca1, f1, s0[m]);
                        s1[m] = fmaf(cb0, f0, s1[m]); s1[m] = fmaf(cb1, f1, s1[m]);
                    }
                }
            }
            if (ok) {
#pragma unroll
                for (int m = 0; m < MR; ++m) { a0[m] = fmaf(sa[u], s0[m], a0[m]); a1[m] = fmaf(sb[u], s1[m], a1[m]); }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        const float t0 = wave_sum(a0[m]), t1 = wave_sum(a1[m]);
        if (lane == 0 && m < p.M) {
            if (swiglu) {
                const float gg = bfround(t0), uu = bfround(t1);
                p.C[(long)m * p.ldc + pp] = f2bf(bfround(silu_f(gg)) * uu);
            } else {
                float o[2] = {t0, t1};
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

template <int MR>
static int launch_gemv_nf4(const GemvNf4Params& p, hipStream_t stream) {
    const size_t smem = (size_t)MR * p.K * sizeof(bf16_t);
    if (smem > 48 * 1024)
        if (int rc = bagel_enable_lds((const void*)gemv_nf4_kernel<MR>, (int)W8_MAX_LDS, "gemv_nf4_kernel")) return rc;
    hipLaunchKernelGGL((gemv_nf4_kernel<MR>), dim3(ceil_div(p.N / 2, 4)), dim3(256), smem, stream, p);
    return bagel_check_launch("gemv_nf4_kernel");
}

extern "C" int bagel_gemv_nf4_bf16(const void* A, int64_t lda, const void* Wq, int64_t ldw_bytes, const float* absmax, const void* bias,
                                   const void* R, int64_t ldr, void* C, int64_t ldc, const void* norm_w, float eps, int32_t M, int32_t N,
                                   int32_t K, int32_t epilogue, hipStream_t stream) {
    BAGEL_REQUIRE(A && Wq && absmax && C, "gemv_nf4: null pointer");
    BAGEL_REQUIRE(K > 0 && (K % 64) == 0 && (lda % 8) == 0 && (ldw_bytes % 16) == 0, "gemv_nf4: K must be a multiple of the 64-weight block, lda of 8, ldw of 16 bytes");
    BAGEL_REQUIRE(N > 0 && (N % 2) == 0, "gemv_nf4: N=%d must be even", N);
    BAGEL_REQUIRE(epilogue >= 0 && epilogue <= 3, "gemv_nf4: unknown epilogue %d", epilogue);
    BAGEL_REQUIRE(epilogue != EPI_SWIGLU16 || ((N % 32) == 0 && !bias && !R), "gemv_nf4: swiglu needs N%%32==0, no bias/residual");
    BAGEL_REQUIRE((((uintptr_t)A | (uintptr_t)Wq | (uintptr_t)norm_w) & 15) == 0, "gemv_nf4: A/W/norm_w must be 16-byte aligned");
    BAGEL_REQUIRE((ldc % 2) == 0 && (ldr % 2) == 0 && (((uintptr_t)C | (uintptr_t)R) & 3) == 0, "gemv_nf4: C/R rows must be 4-byte aligned");
    BAGEL_REQUIRE((size_t)K * sizeof(bf16_t) <= W8_MAX_LDS, "gemv_nf4: K=%d does not fit the LDS staging buffer", K);
    if (M <= 0) return BAGEL_OK;
    int m0 = 0;
    while (m0 < M) {
        int mr = (M - m0 >= 4) ? 4 : (M - m0 >= 2 ? 2 : 1);
        while (mr > 1 && (size_t)mr * K * sizeof(bf16_t) > W8_MAX_LDS) mr >>= 1;
        GemvNf4Params p;
        p.A = (const bf16_t*)A + (long)m0 * lda; p.lda = lda;
        p.W = (const unsigned char*)Wq; p.ldw = ldw_bytes; p.absmax = absmax;
        p.bias = (const bf16_t*)bias;
        p.R = R ? (const bf16_t*)R + (long)m0 * ldr : nullptr; p.ldr = ldr;
        p.C = (bf16_t*)C + (long)m0 * ldc; p.ldc = ldc;
        p.norm_w = (const bf16_t*)norm_w; p.eps = eps;
        p.M = mr; p.N = N; p.K = K; p.epi = epilogue;
        int rc = mr == 4 ? launch_gemv_nf4<4>(p, stream) : (mr == 2 ? launch_gemv_nf4<2>(p, stream) : launch_gemv_nf4<1>(p, stream));
        if (rc != BAGEL_OK) return rc;
        m0 += mr;
    }
    return BAGEL_OK;
}
