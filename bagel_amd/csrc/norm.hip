// Row-wise normalisation kernels + the fused QK-RMSNorm/RoPE pass (HBM-bound; wave-reduction kernels).
//
//   bagel_rmsnorm_bf16      Qwen2RMSNorm (modeling_qwen2.py:54-59) with MoT weight routing by row
//                           (qwen2_navit.py:784-787, 812-815, 1079-1082)
//   bagel_layernorm_bf16    nn.LayerNorm of the SigLIP encoder (siglip_navit.py:266-269,342)
//   bagel_rope_table        Qwen2RotaryEmbedding.forward (modeling_qwen2.py:130-150): cos/sin rounded to bf16
//   bagel_qknorm_rope_bf16  q_norm/k_norm + apply_rotary_pos_emb + the bf16 casts of
//                           PackedAttentionMoT.forward_inference (qwen2_navit.py:518-557), in place on the
//                           fused QKV projection buffer.  Cast points are the reference's:
//                             und: norm -> bf16, * w -> bf16, rope with bf16 products and bf16 sum
//                             gen: everything fp32 on the bf16-rounded projection, ONE final bf16 rounding
#include "common.h"

// One wave per row; each lane streams 16-byte (8 x bf16) chunks.  Rows of up to 4096 columns (every hidden size on the path) are
// held in registers: all of a lane's loads are issued back to back (one latency per row instead of one per chunk) and the second
// pass re-reads nothing.  Same summation order as the streaming form, so the two are bit-identical.
__global__ __launch_bounds__(256) void rmsnorm_kernel(const bf16_t* __restrict__ x, long ldx, const bf16_t* __restrict__ w0,
                                                      const bf16_t* __restrict__ w1, const int* __restrict__ expert,
                                                      bf16_t* __restrict__ y, long ldy, int rows, int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16_t* xr = x + (long)row * ldx;
    const bf16_t* w = (expert && expert[row]) ? w1 : w0;
    const int nch = cols >> 3;
    bf16_t* yr = y + (long)row * ldy;
    if (nch <= 512) {
        u32x4_t v[8], g[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = lane + 64 * i;
            v[i] = c < nch ? *(const u32x4_t*)(xr + c * 8) : u32x4_t{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = lane + 64 * i;
            g[i] = c < nch ? *(const u32x4_t*)(w + c * 8) : u32x4_t{0u, 0u, 0u, 0u};
        }
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = lo2f(v[i][e]), b = hi2f(v[i][e]);
                ss += a * a + b * b;
            }
        ss = wave_sum(ss);
        const float inv = rsqrtf(ss / (float)cols + eps);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = lane + 64 * i;
            u32x4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = bfround(lo2f(v[i][e]) * inv) * lo2f(g[i][e]);
                const float b = bfround(hi2f(v[i][e]) * inv) * hi2f(g[i][e]);
                o[e] = pack2bf(a, b);
            }
            if (c < nch) *(u32x4_t*)(yr + c * 8) = o;
        }
        return;
    }
    float ss = 0.f;
    for (int c = lane; c < nch; c += 64) {
        const u32x4_t v = *(const u32x4_t*)(xr + c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = lo2f(v[e]), b = hi2f(v[e]);
            ss += a * a + b * b;
        }
    }
    ss = wave_sum(ss);
    const float inv = rsqrtf(ss / (float)cols + eps);
    for (int c = lane; c < nch; c += 64) {
        const u32x4_t v = *(const u32x4_t*)(xr + c * 8);   // L2/L1 hit: the row was just streamed
        const u32x4_t g = *(const u32x4_t*)(w + c * 8);
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = bfround(lo2f(v[e]) * inv) * lo2f(g[e]);
            const float b = bfround(hi2f(v[e]) * inv) * hi2f(g[e]);
            o[e] = pack2bf(a, b);
        }
        *(u32x4_t*)(yr + c * 8) = o;
    }
}

extern "C" int bagel_rmsnorm_bf16(const void* x, int64_t ldx, const void* w0, const void* w1, const int32_t* expert_of_row,
                                  void* y, int64_t ldy, int32_t rows, int32_t cols, float eps, hipStream_t stream) {
    BAGEL_REQUIRE(x && y && w0, "rmsnorm: null pointer");
    BAGEL_REQUIRE(cols > 0 && cols % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "rmsnorm: cols/ld must be multiples of 8");
    BAGEL_REQUIRE(!expert_of_row || w1, "rmsnorm: expert routing needs w1");
    if (rows <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(rmsnorm_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, stream, (const bf16_t*)x, (long)ldx,
                       (const bf16_t*)w0, (const bf16_t*)w1, expert_of_row, (bf16_t*)y, (long)ldy, rows, cols, eps);
    return bagel_check_launch("rmsnorm_kernel");
}

// Qwen2RMSNorm whose output goes straight to the FP8 (OCP e4m3) operand of the following gen-expert GEMM (bagel_gemm_fp8_bf16):
// q[r, :] = e4m3(y[r, :] / s_r), s_r = max |y[r, :]| / 448, y = the bf16 result of rmsnorm_kernel -- bit-identical to running
// rmsnorm_kernel and bagel_quantize_rows_fp8 one after the other, with 3 instead of 7 bytes of HBM traffic per element.
template <bool REG>
__device__ __forceinline__ void rmsnorm_fp8_row(const bf16_t* __restrict__ xr, const bf16_t* __restrict__ w, unsigned char* __restrict__ qr,
                                                float* __restrict__ scale_out, int nch, int cols, float eps, int lane) {
    // REG: the row (<= 4096 columns) lives in registers across the three passes; otherwise it is re-read from L2
    constexpr int NI = 8;
    u32x4_t vr[NI], gr[NI];
    const u32x4_t zero = u32x4_t{0u, 0u, 0u, 0u};
    if (REG) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = lane + 64 * i;
            vr[i] = c < nch ? *(const u32x4_t*)(xr + c * 8) : zero;
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = lane + 64 * i;
            gr[i] = c < nch ? *(const u32x4_t*)(w + c * 8) : zero;
        }
    }
    const int niter = REG ? NI : (nch + 63) >> 6;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < niter; ++i) {
        const int c = lane + 64 * i;
        const u32x4_t v = REG ? vr[REG ? i : 0] : (c < nch ? *(const u32x4_t*)(xr + c * 8) : zero);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = lo2f(v[e]), b = hi2f(v[e]);
            ss += a * a + b * b;
        }
    }
    ss = wave_sum(ss);
    const float inv = rsqrtf(ss / (float)cols + eps);
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < niter; ++i) {
        const int c = lane + 64 * i;
        const bool ok = c < nch;
        const u32x4_t v = REG ? vr[REG ? i : 0] : (ok ? *(const u32x4_t*)(xr + c * 8) : zero);
        const u32x4_t g = REG ? gr[REG ? i : 0] : (ok ? *(const u32x4_t*)(w + c * 8) : zero);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = bfround(bfround(lo2f(v[e]) * inv) * lo2f(g[e]));
            const float b = bfround(bfround(hi2f(v[e]) * inv) * hi2f(g[e]));
            amax = fmaxf(amax, fmaxf(fabsf(a), fabsf(b)));
        }
    }
    amax = wave_max(amax);
    const float s = amax > 0.f ? amax / 448.0f : 1.0f;
    const float qinv = 1.0f / s;
    if (lane == 0) *scale_out = s;
#pragma unroll
    for (int i = 0; i < niter; ++i) {
        const int c = lane + 64 * i;
        const bool ok = c < nch;
        const u32x4_t v = REG ? vr[REG ? i : 0] : (ok ? *(const u32x4_t*)(xr + c * 8) : zero);
        const u32x4_t g = REG ? gr[REG ? i : 0] : (ok ? *(const u32x4_t*)(w + c * 8) : zero);
        u32x2_t o;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float y[4];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                y[2 * e] = bfround(bfround(lo2f(v[2 * h + e]) * inv) * lo2f(g[2 * h + e]));
                y[2 * e + 1] = bfround(bfround(hi2f(v[2 * h + e]) * inv) * hi2f(g[2 * h + e]));
            }
            int wd = 0;
            wd = __builtin_amdgcn_cvt_pk_fp8_f32(y[0] * qinv, y[1] * qinv, wd, false);
            wd = __builtin_amdgcn_cvt_pk_fp8_f32(y[2] * qinv, y[3] * qinv, wd, true);
            o[h] = (unsigned)wd;
        }
        if (ok) *(u32x2_t*)(qr + 8 * c) = o;
    }
}

__global__ __launch_bounds__(256) void rmsnorm_fp8_kernel(const bf16_t* __restrict__ x, long ldx, const bf16_t* __restrict__ w,
                                                          unsigned char* __restrict__ q, long ldq, float* __restrict__ scale, int rows,
                                                          int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = cols >> 3;
    if (nch <= 512) rmsnorm_fp8_row<true>(x + (long)row * ldx, w, q + (long)row * ldq, scale + row, nch, cols, eps, lane);
    else rmsnorm_fp8_row<false>(x + (long)row * ldx, w, q + (long)row * ldq, scale + row, nch, cols, eps, lane);
}

extern "C" int bagel_rmsnorm_fp8(const void* x, int64_t ldx, const void* w, void* q, int64_t ldq_bytes, float* scale, int32_t rows,
                                 int32_t cols, float eps, hipStream_t stream) {
    BAGEL_REQUIRE(x && q && w && scale, "rmsnorm_fp8: null pointer");
    BAGEL_REQUIRE(cols > 0 && cols % 8 == 0 && ldx % 8 == 0 && ldq_bytes % 8 == 0, "rmsnorm_fp8: cols/ld must be multiples of 8");
    if (rows <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(rmsnorm_fp8_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, stream, (const bf16_t*)x, (long)ldx, (const bf16_t*)w,
                       (unsigned char*)q, (long)ldq_bytes, scale, rows, cols, eps);
    return bagel_check_launch("rmsnorm_fp8_kernel");
}

// LayerNorm, bf16 in/out, fp32 statistics (two-pass: mean, then centred variance -- matches ATen's CPU kernel
// to rounding).  Affine in fp32, one rounding to bf16.
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x, long ldx, const bf16_t* __restrict__ w,
                                                        const bf16_t* __restrict__ b, bf16_t* __restrict__ y, long ldy,
                                                        int rows, int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16_t* xr = x + (long)row * ldx;
    const int nch = cols >> 3;
    float s = 0.f;
    for (int c = lane; c < nch; c += 64) {
        const u32x4_t v = *(const u32x4_t*)(xr + c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) s += lo2f(v[e]) + hi2f(v[e]);
    }
    const float mean = wave_sum(s) / (float)cols;
    float ss = 0.f;
    for (int c = lane; c < nch; c += 64) {
        const u32x4_t v = *(const u32x4_t*)(xr + c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = lo2f(v[e]) - mean, bb = hi2f(v[e]) - mean;
            ss += a * a + bb * bb;
        }
    }
    const float inv = rsqrtf(wave_sum(ss) / (float)cols + eps);
    bf16_t* yr = y + (long)row * ldy;
    for (int c = lane; c < nch; c += 64) {
        const u32x4_t v = *(const u32x4_t*)(xr + c * 8);
        const u32x4_t g = *(const u32x4_t*)(w + c * 8);
        const u32x4_t bb = *(const u32x4_t*)(b + c * 8);
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            o[e] = pack2bf((lo2f(v[e]) - mean) * inv * lo2f(g[e]) + lo2f(bb[e]), (hi2f(v[e]) - mean) * inv * hi2f(g[e]) + hi2f(bb[e]));
        *(u32x4_t*)(yr + c * 8) = o;
    }
}

extern "C" int bagel_layernorm_bf16(const void* x, int64_t ldx, const void* w, const void* b, void* y, int64_t ldy,
                                    int32_t rows, int32_t cols, float eps, hipStream_t stream) {
    BAGEL_REQUIRE(x && y && w && b, "layernorm: null pointer");
    BAGEL_REQUIRE(cols > 0 && cols % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "layernorm: cols/ld must be multiples of 8");
    if (rows <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(layernorm_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, stream, (const bf16_t*)x, (long)ldx,
                       (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, (long)ldy, rows, cols, eps);
    return bagel_check_launch("layernorm_kernel");
}

// cos/sin(pos * inv_freq) in fp32, rounded to bf16 (the hidden dtype, modeling_qwen2.py:150). [rows, half]
__global__ void rope_table_kernel(const long* __restrict__ pos, const float* __restrict__ inv_freq, bf16_t* __restrict__ cosb,
                                  bf16_t* __restrict__ sinb, int rows, int half) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * half) return;
    const int r = i / half, c = i - r * half;
    const float ang = __fmul_rn((float)pos[r], inv_freq[c]);
    cosb[i] = f2bf(cosf(ang));
    sinb[i] = f2bf(sinf(ang));
}

extern "C" int bagel_rope_table(const int64_t* position_ids, const float* inv_freq, void* cos_out, void* sin_out,
                                int32_t rows, int32_t half_dim, hipStream_t stream) {
    BAGEL_REQUIRE(position_ids && inv_freq && cos_out && sin_out, "rope_table: null pointer");
    if (rows <= 0) return BAGEL_OK;
    const int n = rows * half_dim;
    hipLaunchKernelGGL(rope_table_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, stream, (const long*)position_ids, inv_freq,
                       (bf16_t*)cos_out, (bf16_t*)sin_out, rows, half_dim);
    return bagel_check_launch("rope_table_kernel");
}

// N dwords (2N bf16) vector load/store
template <int N> __device__ __forceinline__ void vload(unsigned (&d)[N], const bf16_t* p);
template <> __device__ __forceinline__ void vload<4>(unsigned (&d)[4], const bf16_t* p) { const u32x4_t v = *(const u32x4_t*)p; d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3]; }
template <> __device__ __forceinline__ void vload<2>(unsigned (&d)[2], const bf16_t* p) { const u32x2_t v = *(const u32x2_t*)p; d[0] = v[0]; d[1] = v[1]; }
template <> __device__ __forceinline__ void vload<1>(unsigned (&d)[1], const bf16_t* p) { d[0] = *(const unsigned*)p; }
template <int N> __device__ __forceinline__ void vstore(bf16_t* p, const unsigned (&d)[N]);
template <> __device__ __forceinline__ void vstore<4>(bf16_t* p, const unsigned (&d)[4]) { u32x4_t v = {d[0], d[1], d[2], d[3]}; *(u32x4_t*)p = v; }
template <> __device__ __forceinline__ void vstore<2>(bf16_t* p, const unsigned (&d)[2]) { u32x2_t v = {d[0], d[1]}; *(u32x2_t*)p = v; }
template <> __device__ __forceinline__ void vstore<1>(bf16_t* p, const unsigned (&d)[1]) { *(unsigned*)p = d[0]; }

// Fused per-head RMSNorm + RoPE, in place on q and k inside the fused QKV buffer.
//   qkv row layout: [nq * DP | nkv * DP | nkv * DP]; DP = padded head dim (storage), HD = true head dim (<= DP,
//   pad lanes are kept zero).  16 lanes own one head (8 elements each when DP == 128; HD/16 elements otherwise),
//   so a wave covers 4 heads per step and the rotate-half partner (i +- HD/2) sits 8 lanes away.
//   GEN=false: reference "und" cast points; GEN=true: fp32 pipeline with per-row expert weights.
template <int EPL>   // elements per lane (HD = 16 * EPL): 8 -> HD 128, 4 -> HD 64, 2 -> HD 32
__global__ __launch_bounds__(256) void qknorm_rope_kernel(bf16_t* __restrict__ qkv, long ld, const bf16_t* __restrict__ cosb,
                                                          const bf16_t* __restrict__ sinb, const bf16_t* __restrict__ qw0,
                                                          const bf16_t* __restrict__ kw0, const bf16_t* __restrict__ qw1,
                                                          const bf16_t* __restrict__ kw1, const int* __restrict__ expert,
                                                          int rows, int nq, int nkv, int dp, float eps, int gen, int use_norm) {
    constexpr int HD = 16 * EPL;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int sub = lane & 15;           // position inside the head
    const int hq = lane >> 4;            // which of the 4 heads of this step
    const int e0 = sub * EPL;            // first element owned by this lane
    const bool upper = sub >= 8;         // second half of the head (x2 of rotate_half)
    const int ex = (expert && expert[row]) ? 1 : 0;
    const bf16_t* qw = ex ? qw1 : qw0;
    const bf16_t* kw = ex ? kw1 : kw0;
    // cos/sin for this lane's elements: table is [rows, HD/2]; element e uses column e mod HD/2
    float cs[EPL], sn[EPL];
    {
        const int c0 = e0 & (HD / 2 - 1);
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            cs[e] = bf2f(cosb[(long)row * (HD / 2) + c0 + e]);
            sn[e] = bf2f(sinb[(long)row * (HD / 2) + c0 + e]);
        }
    }
    const int nheads = nq + nkv;
    bf16_t* base = qkv + (long)row * ld;
    for (int h0 = 0; h0 < nheads; h0 += 4) {
        const int h = h0 + hq;
        const bool act = h < nheads;
        const int hh = act ? h : nheads - 1;
        bf16_t* p = base + (long)hh * dp + e0;
        const bf16_t* wv = (hh < nq ? qw : kw) + e0;
        float x[EPL], wf[EPL];
        {
            unsigned xr[EPL / 2], wr[EPL / 2];
            vload<EPL / 2>(xr, p);
            if (use_norm) vload<EPL / 2>(wr, wv);
            else {
#pragma unroll
                for (int e = 0; e < EPL / 2; ++e) wr[e] = 0;
            }
#pragma unroll
            for (int e = 0; e < EPL / 2; ++e) {
                x[2 * e] = lo2f(xr[e]);
                x[2 * e + 1] = hi2f(xr[e]);
                wf[2 * e] = use_norm ? lo2f(wr[e]) : 1.0f;
                wf[2 * e + 1] = use_norm ? hi2f(wr[e]) : 1.0f;
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
            for (int e = 0; e < EPL; ++e) {
                if (gen) nrm[e] = __fmul_rn(wf[e], __fmul_rn(x[e], inv));                 // fp32: w * (x * rsqrt)
                else     nrm[e] = bfround(wf[e] * bfround(x[e] * inv));                   // bf16(w * bf16(x * rsqrt))
            }
        } else {
#pragma unroll
            for (int e = 0; e < EPL; ++e) nrm[e] = x[e];
        }
        // rotate_half: lower half gets -x2, upper half gets +x1; partner is 8 lanes away
        float out[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const float partner = __shfl_xor(nrm[e], 8, 64);
            const float rot = upper ? partner : -partner;
            if (gen) out[e] = __fadd_rn(__fmul_rn(nrm[e], cs[e]), __fmul_rn(rot, sn[e]));
            else     out[e] = bfround(nrm[e] * cs[e]) + bfround(rot * sn[e]);
        }
        if (act) {
            unsigned orr[EPL / 2];
#pragma unroll
            for (int e = 0; e < EPL / 2; ++e) orr[e] = pack2bf(out[2 * e], out[2 * e + 1]);
            vstore<EPL / 2>(p, orr);
        }
    }
}

extern "C" int bagel_qknorm_rope_bf16(void* qkv, int64_t ld, const void* cos_tab, const void* sin_tab, const void* q_w0,
                                      const void* k_w0, const void* q_w1, const void* k_w1, const int32_t* expert_of_row,
                                      int32_t rows, int32_t nq, int32_t nkv, int32_t head_dim, int32_t head_dim_padded,
                                      float eps, int32_t gen_mode, int32_t use_norm, hipStream_t stream) {
    BAGEL_REQUIRE(qkv && cos_tab && sin_tab, "qknorm_rope: null pointer");
    BAGEL_REQUIRE(!use_norm || (q_w0 && k_w0), "qknorm_rope: norm weights missing");
    BAGEL_REQUIRE(!expert_of_row || !use_norm || (q_w1 && k_w1), "qknorm_rope: expert routing needs the second weight set");
    BAGEL_REQUIRE(head_dim_padded >= head_dim && ld % 2 == 0, "qknorm_rope: bad head_dim_padded/ld");
    if (rows <= 0) return BAGEL_OK;
    const dim3 grid(ceil_div(rows, 4)), block(256);
#define QKR_LAUNCH(EPL)                                                                                                     \
    hipLaunchKernelGGL(qknorm_rope_kernel<EPL>, grid, block, 0, stream, (bf16_t*)qkv, (long)ld, (const bf16_t*)cos_tab,     \
                       (const bf16_t*)sin_tab, (const bf16_t*)q_w0, (const bf16_t*)k_w0, (const bf16_t*)q_w1,                \
                       (const bf16_t*)k_w1, expert_of_row, rows, nq, nkv, head_dim_padded, eps, gen_mode, use_norm)
    switch (head_dim) {
        case 128: QKR_LAUNCH(8); break;
        case 64: QKR_LAUNCH(4); break;
        case 32: QKR_LAUNCH(2); break;
        default: return bagel_set_error(BAGEL_ERR_UNSUPPORTED, "qknorm_rope: head_dim %d not in {32,64,128}", head_dim);
    }
#undef QKR_LAUNCH
    return bagel_check_launch("qknorm_rope_kernel");
}

// SigLIP 2-D RoPE (siglip_navit.py:102-142,224-230): every head is split in halves, the first rotated by the token's
// row position, the second by its column position; rotate_half pairs element j with j + hd/4 inside a half.  In place on
// the q and k heads of the fused projection rows, bf16 products and bf16 sum like the eager reference ops.
__global__ __launch_bounds__(256) void rope2d_kernel(bf16_t* __restrict__ qkv, long ld, const bf16_t* __restrict__ cos_h,
                                                     const bf16_t* __restrict__ sin_h, const bf16_t* __restrict__ cos_w,
                                                     const bf16_t* __restrict__ sin_w, const long* __restrict__ pos_ids, long n,
                                                     int nheads, int hd, int dp) {
    const int q4 = hd >> 2, h2 = hd >> 1;
    const long total = n * nheads * 2 * q4;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int j = (int)(t % q4);
    const int half = (int)((t / q4) & 1);
    const int head = (int)((t / (2 * q4)) % nheads);
    const long row = t / ((long)2 * q4 * nheads);
    const long pos = pos_ids[row];
    const bf16_t* ct = (half ? cos_w : cos_h) + pos * h2;
    const bf16_t* st = (half ? sin_w : sin_h) + pos * h2;
    bf16_t* p = qkv + row * ld + (long)head * dp + half * h2;
    const float x1 = bf2f(p[j]), x2 = bf2f(p[j + q4]);
    const float o1 = bfround(x1 * bf2f(ct[j])) + bfround(-x2 * bf2f(st[j]));
    const float o2 = bfround(x2 * bf2f(ct[j + q4])) + bfround(x1 * bf2f(st[j + q4]));
    p[j] = f2bf(o1);
    p[j + q4] = f2bf(o2);
}

extern "C" int bagel_rope2d_bf16(void* qkv, int64_t ld, const void* cos_h, const void* sin_h, const void* cos_w, const void* sin_w,
                                 const int64_t* pos_ids, int64_t rows, int32_t nheads, int32_t head_dim, int32_t head_dim_padded,
                                 hipStream_t stream) {
    BAGEL_REQUIRE(qkv && cos_h && sin_h && cos_w && sin_w && pos_ids, "rope2d: null pointer");
    BAGEL_REQUIRE(head_dim > 0 && head_dim % 4 == 0 && head_dim <= head_dim_padded, "rope2d: head_dim %d must be a multiple of 4", head_dim);
    if (rows <= 0 || nheads <= 0) return BAGEL_OK;
    const long total = rows * nheads * (head_dim / 2);
    hipLaunchKernelGGL(rope2d_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, stream, (bf16_t*)qkv, (long)ld, (const bf16_t*)cos_h,
                       (const bf16_t*)sin_h, (const bf16_t*)cos_w, (const bf16_t*)sin_w, (const long*)pos_ids, (long)rows, nheads,
                       head_dim, head_dim_padded);
    return bagel_check_launch("rope2d_kernel");
}
