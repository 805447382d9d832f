// HBM-bound glue kernels of the flow path (row copies, flow-token assembly, CFG combine + renorm + Euler step,
// argmax).  Every rounding point follows the reference's eager bf16 op sequence (bagel.py:796-806, 873-905, 746).
#include "common.h"
#include <string.h>

// ---------------------------------------------------------------------------------------------------------
// Row gather/scatter copy:  dst[dst_rows ? dst_rows[i] : i][0:cols] = src[src_rows ? src_rows[i] : i][0:cols]
// Used for: embed_tokens lookup (bagel.py:796), frozen position-table lookup, KV-cache merge/append
// (qwen2_navit.py:563-570).  One wave per row, 16-byte chunks.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void copy_rows_kernel(const bf16_t* __restrict__ src, long ld_src, const int* __restrict__ src_rows,
                                                        bf16_t* __restrict__ dst, long ld_dst, const int* __restrict__ dst_rows,
                                                        int n, int cols) {
    const int lane = threadIdx.x & 63;
    const int nch = cols >> 3;
    for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += gridDim.x * 4) {
        const long sr = src_rows ? src_rows[i] : i;
        const long dr = dst_rows ? dst_rows[i] : i;
        const bf16_t* s = src + sr * ld_src;
        bf16_t* d = dst + dr * ld_dst;
        for (int c = lane; c < nch; c += 64) *(u32x4_t*)(d + c * 8) = *(const u32x4_t*)(s + c * 8);
    }
}

extern "C" int bagel_copy_rows_bf16(const void* src, int64_t ld_src, const int32_t* src_rows, void* dst, int64_t ld_dst,
                                    const int32_t* dst_rows, int32_t n, int32_t cols, hipStream_t stream) {
    BAGEL_REQUIRE(src && dst, "copy_rows: null pointer");
    BAGEL_REQUIRE(cols % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0, "copy_rows: cols/ld must be multiples of 8");
    if (n <= 0) return BAGEL_OK;
    const int blocks = min(ceil_div(n, 4), 4096);
    hipLaunchKernelGGL(copy_rows_kernel, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)src, (long)ld_src, src_rows,
                       (bf16_t*)dst, (long)ld_dst, dst_rows, n, cols);
    return bagel_check_launch("copy_rows_kernel");
}

// fp32 -> bf16 (the autocast input cast of F.linear; x_t is fp32, bagel.py:803), 2-D with strides; destination columns
// [cols, cols_padded) are zero-filled (K padding for the GEMM, e.g. the 588-wide ViT patch vectors).
__global__ void f32_to_bf16_kernel(const float* __restrict__ s, long ld_src, bf16_t* __restrict__ d, long ld_dst, int rows,
                                   int cols, int cols_padded) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cp2 = cols_padded >> 1;
    if (i >= (long)rows * cp2) return;
    const int r = (int)(i / cp2), c = (int)(i % cp2) * 2;
    const float a = c < cols ? s[(long)r * ld_src + c] : 0.f;
    const float b = c + 1 < cols ? s[(long)r * ld_src + c + 1] : 0.f;
    *(unsigned*)(d + (long)r * ld_dst + c) = pack2bf(a, b);
}

extern "C" int bagel_f32_to_bf16(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, int32_t rows, int32_t cols,
                                 int32_t cols_padded, hipStream_t stream) {
    BAGEL_REQUIRE(src && dst, "f32_to_bf16: null pointer");
    BAGEL_REQUIRE(cols_padded >= cols && cols_padded % 2 == 0 && ld_dst % 2 == 0, "f32_to_bf16: padded width must be even");
    if (rows <= 0 || cols <= 0) return BAGEL_OK;
    const long n = (long)rows * (cols_padded / 2);
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, stream, src, (long)ld_src, (bf16_t*)dst,
                       (long)ld_dst, rows, cols, cols_padded);
    return bagel_check_launch("f32_to_bf16_kernel");
}

// TimestepEmbedder.timestep_embedding (modeling_utils.py:88-104) for ONE timestep: [cos(t f_k) | sin(t f_k)] -> bf16
__global__ void timestep_sinusoid_kernel(float t, const float* __restrict__ freqs, bf16_t* __restrict__ out, int half) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= half) return;
    const float a = __fmul_rn(t, freqs[k]);
    out[k] = f2bf(cosf(a));
    out[half + k] = f2bf(sinf(a));
}

extern "C" int bagel_timestep_sinusoid(float t, const float* freqs, void* out, int32_t half, hipStream_t stream) {
    BAGEL_REQUIRE(freqs && out && half > 0, "timestep_sinusoid: bad arguments");
    hipLaunchKernelGGL(timestep_sinusoid_kernel, dim3(ceil_div(half, 128)), dim3(128), 0, stream, t, freqs, (bf16_t*)out, half);
    return bagel_check_launch("timestep_sinusoid_kernel");
}

// seq[rows[i]] = bf16( bf16(seq[rows[i]] + temb) + pos_table[pos_ids[i]] )     (bagel.py:803-806)
// seq rows already hold vae2llm(x_t) (bf16, bias included) written by the GEMM.
__global__ __launch_bounds__(256) void flow_add_kernel(bf16_t* __restrict__ seq, long ld, const int* __restrict__ rows,
                                                       const bf16_t* __restrict__ temb, const bf16_t* __restrict__ pos_table,
                                                       long ld_pos, const long* __restrict__ pos_ids, int n, int cols) {
    const int lane = threadIdx.x & 63;
    const int nch = cols >> 3;
    for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += gridDim.x * 4) {
        bf16_t* s = seq + (long)rows[i] * ld;
        const bf16_t* pr = pos_table + pos_ids[i] * ld_pos;
        for (int c = lane; c < nch; c += 64) {
            const u32x4_t a = *(const u32x4_t*)(s + c * 8);
            const u32x4_t t = *(const u32x4_t*)(temb + c * 8);
            const u32x4_t q = *(const u32x4_t*)(pr + c * 8);
            u32x4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                o[e] = pack2bf(bfround(lo2f(a[e]) + lo2f(t[e])) + lo2f(q[e]), bfround(hi2f(a[e]) + hi2f(t[e])) + hi2f(q[e]));
            *(u32x4_t*)(s + c * 8) = o;
        }
    }
}

extern "C" int bagel_flow_add_bf16(void* seq, int64_t ld, const int32_t* rows, const void* temb, const void* pos_table,
                                   int64_t ld_pos, const int64_t* pos_ids, int32_t n, int32_t cols, hipStream_t stream) {
    BAGEL_REQUIRE(seq && rows && temb && pos_table && pos_ids, "flow_add: null pointer");
    BAGEL_REQUIRE(cols % 8 == 0 && ld % 8 == 0 && ld_pos % 8 == 0, "flow_add: cols/ld must be multiples of 8");
    if (n <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(flow_add_kernel, dim3(min(ceil_div(n, 4), 4096)), dim3(256), 0, stream, (bf16_t*)seq, (long)ld, rows,
                       (const bf16_t*)temb, (const bf16_t*)pos_table, (long)ld_pos, (const long*)pos_ids, n, cols);
    return bagel_check_launch("flow_add_kernel");
}

// dst[i] = bf16(a[i] + b[rows? ...])  -- generic "x + table[ids]" used by the ViT path (siglip_navit.py:192, bagel.py:391-392)
__global__ __launch_bounds__(256) void add_rows_kernel(bf16_t* __restrict__ x, long ld, const bf16_t* __restrict__ table, long ld_t,
                                                       const long* __restrict__ ids, int n, int cols) {
    const int lane = threadIdx.x & 63;
    const int nch = cols >> 3;
    for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += gridDim.x * 4) {
        bf16_t* s = x + (long)i * ld;
        const bf16_t* pr = table + ids[i] * ld_t;
        for (int c = lane; c < nch; c += 64) {
            const u32x4_t a = *(const u32x4_t*)(s + c * 8);
            const u32x4_t q = *(const u32x4_t*)(pr + c * 8);
            u32x4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = pack2bf(lo2f(a[e]) + lo2f(q[e]), hi2f(a[e]) + hi2f(q[e]));
            *(u32x4_t*)(s + c * 8) = o;
        }
    }
}

extern "C" int bagel_add_table_rows_bf16(void* x, int64_t ld, const void* table, int64_t ld_table, const int64_t* ids, int32_t n,
                                         int32_t cols, hipStream_t stream) {
    BAGEL_REQUIRE(x && table && ids, "add_table_rows: null pointer");
    BAGEL_REQUIRE(cols % 8 == 0 && ld % 8 == 0 && ld_table % 8 == 0, "add_table_rows: cols/ld must be multiples of 8");
    if (n <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(add_rows_kernel, dim3(min(ceil_div(n, 4), 4096)), dim3(256), 0, stream, (bf16_t*)x, (long)ld,
                       (const bf16_t*)table, (long)ld_table, (const long*)ids, n, cols);
    return bagel_check_launch("add_rows_kernel");
}

// ---------------------------------------------------------------------------------------------------------
// Classifier-free-guidance combine + renorm + Euler update (bagel.py:873-905, 746).  bf16 eager semantics:
// every binary op rounds to bf16; torch.norm accumulates fp32 and returns bf16; `+1e-8`, the division and the
// clamp bounds are bf16; v*dt is a bf16 product (0-dim fp32 dt does not promote), x_t stays fp32.
//   mode 0 "global": one scale for the whole LOCAL batch (two launches: partial sums, then apply)
//   mode 1 "channel": per-token scale;   mode 2 "text_channel": per-token scale after the text stage only
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float cfg_mix(float base, float x, float s) {   // base + s * (x - base), bf16 at every step
    return bfround(base + bfround(s * bfround(x - base)));
}
__device__ __forceinline__ float renorm_scale(float ss0, float ss1, float mn_bf) {
    const float n0 = bfround(sqrtf(ss0)), n1 = bfround(sqrtf(ss1));
    const float r = bfround(n0 / bfround(n1 + 1e-8f));
    return fminf(fmaxf(r, mn_bf), 1.0f);
}

#define CFG_MAX_PER_LANE 4   // cols <= 256

__global__ __launch_bounds__(256) void cfg_stage1_kernel(const bf16_t* __restrict__ v, const bf16_t* __restrict__ vct,
                                                         const bf16_t* __restrict__ vci, bf16_t* __restrict__ tmp,
                                                         float* __restrict__ partials, int n, int cols, float s_text,
                                                         float s_img, float mn_bf, int mode) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float acc0 = 0.f, acc1 = 0.f;
    for (int i = blockIdx.x * 4 + wv; i < n; i += gridDim.x * 4) {
        float a[CFG_MAX_PER_LANE], t[CFG_MAX_PER_LANE], im[CFG_MAX_PER_LANE];
        float ss0 = 0.f, ss1 = 0.f;
#pragma unroll
        for (int k = 0; k < CFG_MAX_PER_LANE; ++k) {
            const int c = lane + 64 * k;
            a[k] = t[k] = im[k] = 0.f;
            if (c < cols) {
                const long idx = (long)i * cols + c;
                a[k] = bf2f(v[idx]);
                const float vt_ = cfg_mix(bf2f(vct[idx]), a[k], s_text);           // v_t_text_
                im[k] = vci ? bf2f(vci[idx]) : 0.f;
                if (mode == 2) t[k] = vt_;
                else t[k] = vci ? cfg_mix(im[k], vt_, s_img) : vt_;                 // v_t_
                ss0 += a[k] * a[k];
                ss1 += t[k] * t[k];
            }
        }
        if (mode == 0) {
            acc0 += ss0; acc1 += ss1;
#pragma unroll
            for (int k = 0; k < CFG_MAX_PER_LANE; ++k) {
                const int c = lane + 64 * k;
                if (c < cols) tmp[(long)i * cols + c] = f2bf(t[k]);
            }
        } else {
            const float sc = renorm_scale(wave_sum(ss0), wave_sum(ss1), mn_bf);
#pragma unroll
            for (int k = 0; k < CFG_MAX_PER_LANE; ++k) {
                const int c = lane + 64 * k;
                if (c < cols) {
                    float o = bfround(t[k] * sc);
                    if (mode == 2 && vci) o = cfg_mix(im[k], o, s_img);
                    tmp[(long)i * cols + c] = f2bf(o);
                }
            }
        }
    }
    if (mode == 0) {
        __shared__ float red[2][4];
        acc0 = wave_sum(acc0); acc1 = wave_sum(acc1);
        if (lane == 0) { red[0][wv] = acc0; red[1][wv] = acc1; }
        __syncthreads();
        if (threadIdx.x == 0) {
            partials[2 * blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
            partials[2 * blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        }
    }
}

// x_t -= float(bf16(v_t * dt)),  v_t = use_scale ? bf16(tmp * scale) : tmp, scale from the stage-1 partials.
__global__ __launch_bounds__(256) void cfg_stage2_euler_kernel(float* __restrict__ x, const bf16_t* __restrict__ tmp,
                                                               const float* __restrict__ partials, int nparts, float mn_bf,
                                                               float dt, long n, int use_scale) {
    float sc = 1.0f;
    if (use_scale) {
        // fixed-order reduction of the partial sums: every block computes the identical value
        __shared__ float red[2][256];
        float a0 = 0.f, a1 = 0.f;
        for (int i = threadIdx.x; i < nparts; i += 256) { a0 += partials[2 * i]; a1 += partials[2 * i + 1]; }
        red[0][threadIdx.x] = a0; red[1][threadIdx.x] = a1;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (threadIdx.x < s) { red[0][threadIdx.x] += red[0][threadIdx.x + s]; red[1][threadIdx.x] += red[1][threadIdx.x + s]; }
            __syncthreads();
        }
        sc = renorm_scale(red[0][0], red[1][0], mn_bf);
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float vt = bf2f(tmp[i]);
        if (use_scale) vt = bfround(vt * sc);
        x[i] = x[i] - bfround(vt * dt);
    }
}

extern "C" int bagel_cfg_stage1(const void* v, const void* v_cfg_text, const void* v_cfg_img, void* tmp, float* partials,
                                int32_t max_partials, int32_t n_rows, int32_t cols, float text_scale, float img_scale,
                                float renorm_min, int32_t mode, int32_t* nparts_out, hipStream_t stream) {
    BAGEL_REQUIRE(v && v_cfg_text && tmp && partials && nparts_out, "cfg_stage1: null pointer");
    BAGEL_REQUIRE(cols > 0 && cols <= 64 * CFG_MAX_PER_LANE, "cfg_stage1: cols=%d unsupported", cols);
    BAGEL_REQUIRE(mode >= 0 && mode <= 2, "cfg_stage1: bad renorm mode %d", mode);
    const int blocks = min(min(ceil_div(n_rows, 4), 256), max_partials);
    BAGEL_REQUIRE(blocks > 0, "cfg_stage1: empty input");
    *nparts_out = blocks;
    // clamp(min=...) casts the python float to the tensor dtype (bf16)
    unsigned u; float mnf = renorm_min; memcpy(&u, &mnf, 4);
    u += 0x7fffu + ((u >> 16) & 1u); u &= 0xffff0000u; memcpy(&mnf, &u, 4);
    hipLaunchKernelGGL(cfg_stage1_kernel, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)v, (const bf16_t*)v_cfg_text,
                       (const bf16_t*)v_cfg_img, (bf16_t*)tmp, partials, n_rows, cols, text_scale, img_scale, mnf, mode);
    return bagel_check_launch("cfg_stage1_kernel");
}

extern "C" int bagel_cfg_stage2_euler(float* x_t, const void* v_or_tmp, const float* partials, int32_t nparts, float renorm_min,
                                      float dt, int64_t n_elems, int32_t use_global_scale, hipStream_t stream) {
    BAGEL_REQUIRE(x_t && v_or_tmp, "cfg_stage2: null pointer");
    BAGEL_REQUIRE(!use_global_scale || (partials && nparts > 0), "cfg_stage2: partials missing");
    if (n_elems <= 0) return BAGEL_OK;
    unsigned u; float mnf = renorm_min; memcpy(&u, &mnf, 4);
    u += 0x7fffu + ((u >> 16) & 1u); u &= 0xffff0000u; memcpy(&mnf, &u, 4);
    const int blocks = (int)min((long)ceil_div(n_elems, 256), 2048L);
    hipLaunchKernelGGL(cfg_stage2_euler_kernel, dim3(blocks), dim3(256), 0, stream, x_t, (const bf16_t*)v_or_tmp, partials,
                       nparts, mnf, dt, (long)n_elems, use_global_scale);
    return bagel_check_launch("cfg_stage2_euler_kernel");
}

// ---------------------------------------------------------------------------------------------------------
// argmax over bf16 logits rows (bagel.py:984); ties -> lowest index (torch.argmax).  One block per row.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void argmax_kernel(const bf16_t* __restrict__ x, long ld, long* __restrict__ out, int cols) {
    const bf16_t* r = x + (long)blockIdx.x * ld;
    const int tid = threadIdx.x;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    // 16-byte lanes when the row allows it (the lm_head logits: 152064 columns = 19008 chunks, ~19 per thread)
    const bool vec = ((((uintptr_t)r) & 15) == 0);
    const int nvec = vec ? (cols >> 3) : 0;
    for (int c = tid; c < nvec; c += 1024) {
        const u32x4_t v = *(const u32x4_t*)(r + (long)c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float lo = lo2f(v[e]), hi = hi2f(v[e]);
            const int i0 = c * 8 + 2 * e;
            if (lo > best) { best = lo; bi = i0; }          // within a thread the indices only grow: strict > keeps the lowest
            if (hi > best) { best = hi; bi = i0 + 1; }
        }
    }
    for (int c = nvec * 8 + tid; c < cols; c += 1024) {
        const float f = bf2f(r[c]);
        if (f > best || (f == best && c < bi)) { best = f; bi = c; }
    }
    __shared__ float sv[1024];
    __shared__ int si[1024];
    sv[tid] = best; si[tid] = bi;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) {
            const float f = sv[tid + s]; const int j = si[tid + s];
            if (f > sv[tid] || (f == sv[tid] && j < si[tid])) { sv[tid] = f; si[tid] = j; }
        }
        __syncthreads();
    }
    if (tid == 0) out[blockIdx.x] = si[0] == 0x7fffffff ? 0 : si[0];   // a row of -inf: index 0, like torch
}

extern "C" int bagel_argmax_bf16(const void* logits, int64_t ld, int64_t* out, int32_t rows, int32_t cols, hipStream_t stream) {
    BAGEL_REQUIRE(logits && out && cols > 0, "argmax: bad arguments");
    if (rows <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(argmax_kernel, dim3(rows), dim3(1024), 0, stream, (const bf16_t*)logits, (long)ld, (long*)out, cols);
    return bagel_check_launch("argmax_kernel");
}

// ---------------------------------------------------------------------------------------------------------
// Sampling of the next token ON THE DEVICE (bagel.py:980-983: probs = softmax(pred_logits / temperature); curr_tokens = multinomial(probs, 1)).
// torch.multinomial draws from torch's generator on the host side of the step, which keeps a sampled decode out of the hipGraph (one eager
// launch sequence per token).  The Gumbel-max form draws from EXACTLY the same categorical distribution with one pass over the logits:
//     token = argmax_i ( z_i + g_i ),   z_i = bf16(logit_i / temperature)  (the reference divides bf16 logits in bf16),   g_i = -log(-log(u_i)),
// u_i uniform in (0, 1) from Philox4x32-10 keyed by the call's 64-bit seed with the counter (i / 4, row, step, 0) -- `step` is read from the
// decode session's device-side step counter, so every replay of the captured step draws fresh numbers.  The RNG STREAM is not torch's (no two
// devices share one anyway, as the reference's own comment at bagel.py:980 notes); the seed is drawn from torch's generator once per call, so
// torch.manual_seed still makes a run reproducible.  oracle/sampling.py restates it; tests pin ids and the empirical distribution.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0], p1 = (unsigned long long)0xCD9E8D57u * c[2];
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1;
        c[0] = n0; c[1] = (unsigned)p1; c[2] = n2; c[3] = (unsigned)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ float gumbel_of(unsigned x) {
    // 23 random bits + 0.5: (2k + 1) / 2 with 2k + 1 < 2^24 is exact in fp32, so u lies in [2^-24, 1 - 2^-24] and both logarithms are finite.
    // (24 bits + 0.5 is NOT exact: 0xFFFFFF + 0.5 rounds to 2^24, u == 1 and the Gumbel value is +inf -- that column would win regardless of its logit.)
    const float u = ((float)(x >> 9) + 0.5f) * 1.1920928955078125e-7f;
    return -logf(-logf(u));
}

// test hook: the uniform -> Gumbel map of the sampler on caller-chosen 32-bit draws (the edge values 0 and 0xFFFFFFFF cannot be reached through a seed)
__global__ void gumbel_of_kernel(const unsigned* __restrict__ x, float* __restrict__ g, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) g[i] = gumbel_of(x[i]);
}
extern "C" int bagel_debug_gumbel_of_u32(const uint32_t* x, float* g, int32_t n, hipStream_t stream) {
    BAGEL_REQUIRE(x && g, "gumbel_of: bad arguments");
    if (n <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(gumbel_of_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, (const unsigned*)x, g, n);
    return bagel_check_launch("gumbel_of_kernel");
}

__global__ __launch_bounds__(1024) void sample_gumbel_kernel(const bf16_t* __restrict__ x, long ld, long* __restrict__ out, int cols, float temperature,
                                                             unsigned seed_lo, unsigned seed_hi, const int* __restrict__ step_ctr) {
    const bf16_t* r = x + (long)blockIdx.x * ld;
    const int tid = threadIdx.x;
    const unsigned step = step_ctr ? (unsigned)step_ctr[0] : 0u;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int q = tid; q * 4 < cols; q += 1024) {                                   // one Philox call = four consecutive columns
        unsigned c[4] = {(unsigned)q, (unsigned)blockIdx.x, step, 0u};
        philox4x32_10(c, seed_lo, seed_hi);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = q * 4 + e;
            if (i < cols) {
                const float z = bfround(bf2f(r[i]) / temperature) + gumbel_of(c[e]);
                if (z > best) { best = z; bi = i; }                                 // indices only grow inside a thread: strict > keeps the lowest
            }
        }
    }
    __shared__ float sv[1024];
    __shared__ int si[1024];
    sv[tid] = best; si[tid] = bi;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) {
            const float f = sv[tid + s]; const int j = si[tid + s];
            if (f > sv[tid] || (f == sv[tid] && j < si[tid])) { sv[tid] = f; si[tid] = j; }
        }
        __syncthreads();
    }
    if (tid == 0) out[blockIdx.x] = si[0] == 0x7fffffff ? 0 : si[0];
}

extern "C" int bagel_sample_gumbel_bf16(const void* logits, int64_t ld, int64_t* out, int32_t rows, int32_t cols, float temperature, int64_t seed,
                                        const int32_t* step_ctr, hipStream_t stream) {
    BAGEL_REQUIRE(logits && out && cols > 0, "sample_gumbel: bad arguments");
    BAGEL_REQUIRE(temperature > 0.f, "sample_gumbel: temperature must be positive (got %g)", (double)temperature);
    if (rows <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(sample_gumbel_kernel, dim3(rows), dim3(1024), 0, stream, (const bf16_t*)logits, (long)ld, (long*)out, cols, temperature,
                       (unsigned)((uint64_t)seed & 0xffffffffu), (unsigned)((uint64_t)seed >> 32), (const int*)step_ctr);
    return bagel_check_launch("sample_gumbel_kernel");
}
