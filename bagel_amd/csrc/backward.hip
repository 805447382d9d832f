// Reverse kernels of the training step (loss.backward() over Bagel.forward, bagel.py:101-229; the reference gets these from torch
// autograd, train/pretrain_unified_navit.py:683-735).  Everything here is HBM-bound row / element work in fp32 arithmetic on bf16
// tensors; the matrix products of the backward (dX = dY W, dW = dY^T X) run on the forward's GEMM kernel over the transposed operand
// images written by bagel_transpose_bf16, and the attention reverse lives in attention_bwd.hip.
//
//   bagel_transpose_bf16          dst[c][j] = src[rows[j]][c]: 64 x 64 tiles through LDS, 16-byte loads and stores on both sides
//   bagel_rmsnorm_bwd_bf16        Qwen2RMSNorm reverse + residual-gradient add, per-expert weight gradients (64-row partial sums,
//                                 then a column pass: deterministic, no atomics in HBM)
//   bagel_layernorm_bwd_bf16      nn.LayerNorm reverse (SigLIP), same two passes, weight and bias gradients
//   bagel_qknorm_rope_bwd_bf16    inverse rotation + per-head RMSNorm reverse on the [q | k | v] gradient rows, q_norm / k_norm gradients
//   bagel_swiglu_bwd_bf16         SiLU-gate reverse in the interleaved [16 gate | 16 up] layout of the fused gate/up GEMM
//   bagel_act_bwd_bf16            GELU-tanh / SiLU reverse (connector, time embedder)
//   bagel_cross_entropy_bwd_bf16  (softmax - onehot) * upstream, in place on the logits
//   bagel_mse_rows_bwd_bf16       2 (pred - target) * upstream
//   bagel_rows_segment_sum_bf16   embedding-gather reverse: rows that share an id are summed by ONE workgroup in index order
//   bagel_colsum_bf16             bias gradients
#include "common.h"

// ------------------------------------------------------------------------------------------------------------------------------
// transpose
// ------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ src, long ld_src, const int* __restrict__ src_rows,
                                                        int rows, int cols, bf16_t* __restrict__ dst, long ld_dst, int rows_padded) {
    __shared__ bf16_t tile[64][72];                          // [j][c], 144-byte rows: the column reads below spread over the banks
    const int j0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tid = threadIdx.x;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = tid + it * 256;                      // 64 rows x 8 chunks
        const int j = idx >> 3, ch = idx & 7;
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (j0 + j < rows && c0 + ch * 8 < cols) {
            const long r = src_rows ? src_rows[j0 + j] : (j0 + j);
            v = *(const u32x4_t*)(src + r * ld_src + c0 + ch * 8);
        }
        *(u32x4_t*)&tile[j][ch * 8] = v;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = tid + it * 256;                      // 64 columns x 8 chunks of 8 rows
        const int c = idx >> 3, jc = idx & 7;
        if (c0 + c >= cols || j0 + jc * 8 >= rows_padded) continue;
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            o[e] = (unsigned)tile[jc * 8 + 2 * e][c] | ((unsigned)tile[jc * 8 + 2 * e + 1][c] << 16);
        *(u32x4_t*)(dst + (long)(c0 + c) * ld_dst + j0 + jc * 8) = o;
    }
}

extern "C" int bagel_transpose_bf16(const void* src, int64_t ld_src, const int32_t* src_rows, int32_t rows, int32_t cols, void* dst,
                                    int64_t ld_dst, int32_t rows_padded, hipStream_t stream) {
    BAGEL_REQUIRE(src && dst, "transpose: null pointer");
    BAGEL_REQUIRE(cols % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0 && rows_padded % 8 == 0, "transpose: cols / ld / rows_padded must be multiples of 8");
    BAGEL_REQUIRE(rows >= 0 && rows_padded >= rows && ld_dst >= rows_padded, "transpose: rows <= rows_padded <= ld_dst expected");
    BAGEL_REQUIRE(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0, "transpose: 16-byte aligned operands expected");
    if (rows_padded <= 0 || cols <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(transpose_kernel, dim3(ceil_div(rows_padded, 64), ceil_div(cols, 64)), dim3(256), 0, stream, (const bf16_t*)src,
                       (long)ld_src, src_rows, rows, cols, (bf16_t*)dst, (long)ld_dst, rows_padded);
    return bagel_check_launch("transpose_kernel");
}

// ------------------------------------------------------------------------------------------------------------------------------
// column sums: 64-row partials (fp32) + a column pass.  partial[blk][width]; the column pass writes up to four bf16 vectors of
// `seg` columns each (out[c / seg][c % seg]).
// ------------------------------------------------------------------------------------------------------------------------------
struct ColsumOut { bf16_t* p[4]; };

__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ partial, int nblk, int width, int seg, ColsumOut out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= width) return;
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) s += partial[(long)b * width + c];
    bf16_t* o = out.p[c / seg];
    if (o) o[c % seg] = f2bf(s);
}

static int colsum_finish(const float* partial, int nblk, int width, int seg, bf16_t* o0, bf16_t* o1, bf16_t* o2, bf16_t* o3, hipStream_t stream) {
    ColsumOut out = {{o0, o1, o2, o3}};
    hipLaunchKernelGGL(colsum_finish_kernel, dim3(ceil_div(width, 256)), dim3(256), 0, stream, partial, nblk, width, seg, out);
    return bagel_check_launch("colsum_finish_kernel");
}

__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ src, long ld, const int* __restrict__ rows, int n_rows, int cols,
                                                     float* __restrict__ partial) {
    const int ch = blockIdx.y * 256 + threadIdx.x;
    if (ch * 8 >= cols) return;
    const int r0 = blockIdx.x * 64, r1 = min(r0 + 64, n_rows);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int j = r0; j < r1; ++j) {
        const long r = rows ? rows[j] : j;
        const u32x4_t v = *(const u32x4_t*)(src + r * ld + ch * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[2 * e] += lo2f(v[e]); acc[2 * e + 1] += hi2f(v[e]); }
    }
    float* p = partial + (long)blockIdx.x * cols + ch * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) p[e] = acc[e];
}

extern "C" int bagel_colsum_bf16(const void* src, int64_t ld, const int32_t* rows, int32_t n_rows, int32_t cols, float* partial_ws, void* out,
                                 hipStream_t stream) {
    BAGEL_REQUIRE(src && partial_ws && out, "colsum: null pointer");
    BAGEL_REQUIRE(cols % 8 == 0 && ld % 8 == 0 && cols > 0, "colsum: cols / ld must be multiples of 8");
    const int nblk = max(ceil_div(n_rows, 64), 0);
    if (nblk > 0) {
        hipLaunchKernelGGL(colsum_kernel, dim3(nblk, ceil_div(cols / 8, 256)), dim3(256), 0, stream, (const bf16_t*)src, (long)ld, rows, n_rows, cols,
                           partial_ws);
        const int rc = bagel_check_launch("colsum_kernel");
        if (rc) return rc;
    }
    return colsum_finish(partial_ws, nblk, cols, cols, (bf16_t*)out, nullptr, nullptr, nullptr, stream);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Qwen2RMSNorm reverse (modeling_qwen2.py:54-59: y = w * bf16(x * rsqrt(mean(x^2) + eps))), two passes, both deterministic:
//   rows     one wave per row (8 rows per wave, 64 per workgroup of 8 waves): rsqrt and the dot product, dx, g += dx; the row's rsqrt
//            goes to a small fp32 side buffer.  No weight-gradient state in registers, so several waves per SIMD hide the latencies.
//   columns  dw_e[c] = sum over the rows of expert e of dy * bf16(x * rsqrt): a thread owns 8 columns and walks 64 rows (no cross-thread
//            reduction), 64-row partial sums, then the shared column pass.  Re-reads x and dy (6 instead of 4 passes over [rows, cols]).
// ------------------------------------------------------------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(512) void rmsnorm_bwd_rows_kernel(const bf16_t* __restrict__ x, long ldx, const bf16_t* __restrict__ dy, long lddy,
                                                               const bf16_t* __restrict__ w0, const bf16_t* __restrict__ w1,
                                                               const int* __restrict__ expert, bf16_t* __restrict__ g, long ldg, int accumulate,
                                                               float* __restrict__ rinv_out, int rows, int cols, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = cols >> 3;
    const int r0 = blockIdx.x * 64 + wave * 8;
    for (int r = r0; r < min(r0 + 8, rows); ++r) {
        const int ex = (expert && w1) ? expert[r] : 0;
        const bf16_t* wr = ex ? w1 : w0;
        float xv[NV][8], dv[NV][8];
        float ss = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = lane + 64 * v;
            u32x4_t a = {0u, 0u, 0u, 0u}, b = {0u, 0u, 0u, 0u}, ww = {0u, 0u, 0u, 0u};
            if (c < nch) {
                a = *(const u32x4_t*)(x + (long)r * ldx + c * 8);
                b = *(const u32x4_t*)(dy + (long)r * lddy + c * 8);
                ww = *(const u32x4_t*)(wr + c * 8);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xv[v][2 * e] = lo2f(a[e]); xv[v][2 * e + 1] = hi2f(a[e]);
                dv[v][2 * e] = lo2f(b[e]) * lo2f(ww[e]); dv[v][2 * e + 1] = hi2f(b[e]) * hi2f(ww[e]);      // dy * w
                ss += xv[v][2 * e] * xv[v][2 * e] + xv[v][2 * e + 1] * xv[v][2 * e + 1];
            }
        }
        ss = wave_sum(ss);
        const float rinv = rsqrtf(ss / (float)cols + eps);
        if (lane == 0) rinv_out[r] = rinv;
        float dot = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int i = 0; i < 8; ++i) { xv[v][i] *= rinv; dot += dv[v][i] * xv[v][i]; }
        dot = wave_sum(dot) / (float)cols;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = lane + 64 * v;
            if (c >= nch) continue;
            u32x4_t gin = {0u, 0u, 0u, 0u};
            if (accumulate) gin = *(const u32x4_t*)(g + (long)r * ldg + c * 8);
            u32x4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dx0 = rinv * (dv[v][2 * e] - xv[v][2 * e] * dot), dx1 = rinv * (dv[v][2 * e + 1] - xv[v][2 * e + 1] * dot);
                o[e] = pack2bf(lo2f(gin[e]) + bfround(dx0), hi2f(gin[e]) + bfround(dx1));
            }
            *(u32x4_t*)(g + (long)r * ldg + c * 8) = o;
        }
    }
}

__global__ __launch_bounds__(256) void rmsnorm_bwd_cols_kernel(const bf16_t* __restrict__ x, long ldx, const bf16_t* __restrict__ dy, long lddy,
                                                               const int* __restrict__ expert, const float* __restrict__ rinv,
                                                               float* __restrict__ partial, int rows, int cols) {
    const int ch = blockIdx.y * 256 + threadIdx.x;
    if (ch * 8 >= cols) return;
    const int r0 = blockIdx.x * 64, r1 = min(r0 + 64, rows);
    float acc[2][8];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[e][i] = 0.f;
    for (int r = r0; r < r1; ++r) {
        const int ex = expert ? expert[r] : 0;
        const float ri = rinv[r];
        const u32x4_t a = *(const u32x4_t*)(x + (long)r * ldx + ch * 8), b = *(const u32x4_t*)(dy + (long)r * lddy + ch * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d0 = lo2f(b[e]) * bfround(lo2f(a[e]) * ri), d1 = hi2f(b[e]) * bfround(hi2f(a[e]) * ri);
            if (ex) { acc[1][2 * e] += d0; acc[1][2 * e + 1] += d1; }
            else    { acc[0][2 * e] += d0; acc[0][2 * e + 1] += d1; }
        }
    }
    float* p = partial + (long)blockIdx.x * 2 * cols + ch * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) { p[i] = acc[0][i]; p[cols + i] = acc[1][i]; }
}

extern "C" int bagel_rmsnorm_bwd_bf16(const void* x, int64_t ldx, const void* dy, int64_t lddy, const void* w0, const void* w1,
                                      const int32_t* expert_of_row, void* g, int64_t ldg, int32_t accumulate, void* dw0, void* dw1,
                                      float* partial_ws, int32_t rows, int32_t cols, float eps, hipStream_t stream) {
    BAGEL_REQUIRE(x && dy && w0 && g && dw0 && partial_ws, "rmsnorm_bwd: null pointer");
    BAGEL_REQUIRE(cols % 8 == 0 && cols > 0 && cols <= 4096, "rmsnorm_bwd: cols must be a multiple of 8, at most 4096");
    BAGEL_REQUIRE(ldx % 8 == 0 && lddy % 8 == 0 && ldg % 8 == 0, "rmsnorm_bwd: leading dimensions must be multiples of 8");
    BAGEL_REQUIRE((w1 == nullptr) == (dw1 == nullptr), "rmsnorm_bwd: w1 and dw1 go together");
    const int nblk = max(ceil_div(rows, 64), 0);
    if (nblk > 0) {
        float* rinv = partial_ws + (long)nblk * 2 * cols;            // the rows' rsqrt values, behind the partial sums
        const int nv = ceil_div(cols / 8, 64);
#define BAGEL_RMSBWD(NV)                                                                                                                 \
        hipLaunchKernelGGL(rmsnorm_bwd_rows_kernel<NV>, dim3(nblk), dim3(512), 0, stream, (const bf16_t*)x, (long)ldx, (const bf16_t*)dy,  \
                           (long)lddy, (const bf16_t*)w0, (const bf16_t*)w1, expert_of_row, (bf16_t*)g, (long)ldg, (int)accumulate, rinv,   \
                           rows, cols, eps)
        if (nv <= 1) BAGEL_RMSBWD(1);
        else if (nv <= 2) BAGEL_RMSBWD(2);
        else if (nv <= 4) BAGEL_RMSBWD(4);
        else BAGEL_RMSBWD(8);
#undef BAGEL_RMSBWD
        if (int rc = bagel_check_launch("rmsnorm_bwd_rows_kernel")) return rc;
        hipLaunchKernelGGL(rmsnorm_bwd_cols_kernel, dim3(nblk, ceil_div(cols / 8, 256)), dim3(256), 0, stream, (const bf16_t*)x, (long)ldx,
                           (const bf16_t*)dy, (long)lddy, w1 ? expert_of_row : nullptr, rinv, partial_ws, rows, cols);
        if (int rc = bagel_check_launch("rmsnorm_bwd_cols_kernel")) return rc;
    }
    return colsum_finish(partial_ws, nblk, 2 * cols, cols, (bf16_t*)dw0, (bf16_t*)dw1, nullptr, nullptr, stream);
}

// ------------------------------------------------------------------------------------------------------------------------------
// nn.LayerNorm reverse (SigLIP, siglip_navit.py:266-269,342: y = bf16((x - mean) * rsqrt(var + eps) * w + b), fp32 statistics), the
// same two deterministic passes as the RMSNorm reverse: rows -> dx (+ residual gradient), mean and rsqrt to a side buffer;
// columns -> dw[c] = sum dy * xh, db[c] = sum dy (partial layout [block][dw | db]).
// ------------------------------------------------------------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(512) void layernorm_bwd_rows_kernel(const bf16_t* __restrict__ x, long ldx, const bf16_t* __restrict__ dy, long lddy,
                                                                 const bf16_t* __restrict__ w, bf16_t* __restrict__ g, long ldg, int accumulate,
                                                                 float* __restrict__ stats, int rows, int cols, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = cols >> 3;
    const int r0 = blockIdx.x * 64 + wave * 8;
    for (int r = r0; r < min(r0 + 8, rows); ++r) {
        float xv[NV][8], dv[NV][8];
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = lane + 64 * v;
            u32x4_t a = {0u, 0u, 0u, 0u}, b = {0u, 0u, 0u, 0u}, ww = {0u, 0u, 0u, 0u};
            if (c < nch) {
                a = *(const u32x4_t*)(x + (long)r * ldx + c * 8);
                b = *(const u32x4_t*)(dy + (long)r * lddy + c * 8);
                ww = *(const u32x4_t*)(w + c * 8);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xv[v][2 * e] = lo2f(a[e]); xv[v][2 * e + 1] = hi2f(a[e]);
                dv[v][2 * e] = lo2f(b[e]) * lo2f(ww[e]); dv[v][2 * e + 1] = hi2f(b[e]) * hi2f(ww[e]);      // dy * w
                s += xv[v][2 * e] + xv[v][2 * e + 1];
            }
        }
        const float mean = wave_sum(s) / (float)cols;
        float ss = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const bool on = lane + 64 * v < nch;
#pragma unroll
            for (int i = 0; i < 8; ++i) { xv[v][i] = on ? xv[v][i] - mean : 0.f; ss += xv[v][i] * xv[v][i]; }
        }
        const float rinv = rsqrtf(wave_sum(ss) / (float)cols + eps);
        if (lane == 0) { stats[2 * r] = mean; stats[2 * r + 1] = rinv; }
        float d1 = 0.f, d2 = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int i = 0; i < 8; ++i) { xv[v][i] *= rinv; d1 += dv[v][i]; d2 += dv[v][i] * xv[v][i]; }
        d1 = wave_sum(d1) / (float)cols;
        d2 = wave_sum(d2) / (float)cols;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = lane + 64 * v;
            if (c >= nch) continue;
            u32x4_t gin = {0u, 0u, 0u, 0u};
            if (accumulate) gin = *(const u32x4_t*)(g + (long)r * ldg + c * 8);
            u32x4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dx0 = rinv * (dv[v][2 * e] - d1 - xv[v][2 * e] * d2), dx1 = rinv * (dv[v][2 * e + 1] - d1 - xv[v][2 * e + 1] * d2);
                o[e] = pack2bf(lo2f(gin[e]) + bfround(dx0), hi2f(gin[e]) + bfround(dx1));
            }
            *(u32x4_t*)(g + (long)r * ldg + c * 8) = o;
        }
    }
}

__global__ __launch_bounds__(256) void layernorm_bwd_cols_kernel(const bf16_t* __restrict__ x, long ldx, const bf16_t* __restrict__ dy, long lddy,
                                                                 const float* __restrict__ stats, float* __restrict__ partial, int rows, int cols) {
    const int ch = blockIdx.y * 256 + threadIdx.x;
    if (ch * 8 >= cols) return;
    const int r0 = blockIdx.x * 64, r1 = min(r0 + 64, rows);
    float dw[8], db[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { dw[i] = 0.f; db[i] = 0.f; }
    for (int r = r0; r < r1; ++r) {
        const float mean = stats[2 * r], ri = stats[2 * r + 1];
        const u32x4_t a = *(const u32x4_t*)(x + (long)r * ldx + ch * 8), b = *(const u32x4_t*)(dy + (long)r * lddy + ch * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float y0 = lo2f(b[e]), y1 = hi2f(b[e]);
            dw[2 * e] += y0 * ((lo2f(a[e]) - mean) * ri); dw[2 * e + 1] += y1 * ((hi2f(a[e]) - mean) * ri);
            db[2 * e] += y0; db[2 * e + 1] += y1;
        }
    }
    float* p = partial + (long)blockIdx.x * 2 * cols + ch * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) { p[i] = dw[i]; p[cols + i] = db[i]; }
}

extern "C" int bagel_layernorm_bwd_bf16(const void* x, int64_t ldx, const void* dy, int64_t lddy, const void* w, void* g, int64_t ldg,
                                        int32_t accumulate, void* dw, void* db, float* partial_ws, int32_t rows, int32_t cols, float eps,
                                        hipStream_t stream) {
    BAGEL_REQUIRE(x && dy && w && g && dw && db && partial_ws, "layernorm_bwd: null pointer");
    BAGEL_REQUIRE(cols % 8 == 0 && cols > 0 && cols <= 4096, "layernorm_bwd: cols must be a multiple of 8, at most 4096");
    BAGEL_REQUIRE(ldx % 8 == 0 && lddy % 8 == 0 && ldg % 8 == 0, "layernorm_bwd: leading dimensions must be multiples of 8");
    const int nblk = max(ceil_div(rows, 64), 0);
    if (nblk > 0) {
        float* stats = partial_ws + (long)nblk * 2 * cols;           // (mean, rsqrt) per row, behind the partial sums
        const int nv = ceil_div(cols / 8, 64);
#define BAGEL_LNBWD(NV)                                                                                                                    \
        hipLaunchKernelGGL(layernorm_bwd_rows_kernel<NV>, dim3(nblk), dim3(512), 0, stream, (const bf16_t*)x, (long)ldx, (const bf16_t*)dy,  \
                           (long)lddy, (const bf16_t*)w, (bf16_t*)g, (long)ldg, (int)accumulate, stats, rows, cols, eps)
        if (nv <= 1) BAGEL_LNBWD(1);
        else if (nv <= 2) BAGEL_LNBWD(2);
        else if (nv <= 4) BAGEL_LNBWD(4);
        else BAGEL_LNBWD(8);
#undef BAGEL_LNBWD
        if (int rc = bagel_check_launch("layernorm_bwd_rows_kernel")) return rc;
        hipLaunchKernelGGL(layernorm_bwd_cols_kernel, dim3(nblk, ceil_div(cols / 8, 256)), dim3(256), 0, stream, (const bf16_t*)x, (long)ldx,
                           (const bf16_t*)dy, (long)lddy, stats, partial_ws, rows, cols);
        if (int rc = bagel_check_launch("layernorm_bwd_cols_kernel")) return rc;
    }
    return colsum_finish(partial_ws, nblk, 2 * cols, cols, (bf16_t*)dw, (bf16_t*)db, nullptr, nullptr, stream);
}

// ------------------------------------------------------------------------------------------------------------------------------
// QK-norm + RoPE reverse (training form of PackedAttentionMoT.forward_train, qwen2_navit.py:430-455: both experts in the bf16
// pipeline).  Forward per head:  n = w * bf16(x * rsqrt(mean(x^2) + eps));  y1 = n1 c - n2 s,  y2 = n2 c + n1 s.
// ------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sum16(float v) {          // sum over the 16 lanes of a head group
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// A wave works on 4 heads at a time: 16 lanes per head, lane `sub` owns the elements 4 sub .. 4 sub + 3 of BOTH halves of the head (the
// rotation pairs element i with i + head_dim / 2), 8-byte loads and stores.
__global__ __launch_bounds__(256) void qknorm_rope_bwd_kernel(bf16_t* __restrict__ dqkv, long ld, const bf16_t* __restrict__ raw, long ld_raw,
                                                              const bf16_t* __restrict__ cos_t, const bf16_t* __restrict__ sin_t,
                                                              const bf16_t* __restrict__ qw0, const bf16_t* __restrict__ kw0,
                                                              const bf16_t* __restrict__ qw1, const bf16_t* __restrict__ kw1,
                                                              const int* __restrict__ expert, float* __restrict__ partial, int rows, int nq,
                                                              int nkv, int hd, int dp, float eps, int use_norm) {
    __shared__ float red[4 * 128];                           // [expert][q | k][hd]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = hd >> 1;
    const int sub = lane & 15, grp = lane >> 4;
    const bool on = 4 * sub < half;
    float dw[2][2][8];                                       // [expert][q | k][4 of the first half, 4 of the second]
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) dw[e][k][i] = 0.f;
    for (int c = threadIdx.x; c < 4 * hd; c += 256) red[c] = 0.f;
    const int r0 = blockIdx.x * 64 + wave * 16;
    auto ld4 = [&](const bf16_t* p_, float* out) {
        const u32x2_t v = *(const u32x2_t*)p_;
        out[0] = lo2f(v[0]); out[1] = hi2f(v[0]); out[2] = lo2f(v[1]); out[3] = hi2f(v[1]);
    };
    for (int r = r0; r < min(r0 + 16, rows); ++r) {
        const int ex = (expert && qw1) ? expert[r] : 0;
        float c[4] = {0.f, 0.f, 0.f, 0.f}, s[4] = {0.f, 0.f, 0.f, 0.f};
        if (on) { ld4(cos_t + (long)r * half + 4 * sub, c); ld4(sin_t + (long)r * half + 4 * sub, s); }
        for (int h0 = 0; h0 < nq + nkv; h0 += 4) {
            const int h = h0 + grp;
            const bool live = on && h < nq + nkv;
            const int isk = h >= nq;
            bf16_t* d = dqkv + (long)r * ld + (long)h * dp + 4 * sub;
            float d1[4] = {0.f, 0.f, 0.f, 0.f}, d2[4] = {0.f, 0.f, 0.f, 0.f}, n1[4], n2[4];
            if (live) { ld4(d, d1); ld4(d + half, d2); }
#pragma unroll
            for (int i = 0; i < 4; ++i) { n1[i] = d1[i] * c[i] + d2[i] * s[i]; n2[i] = d2[i] * c[i] - d1[i] * s[i]; }     // gradient of the normalised head
            if (use_norm) {
                const bf16_t* xr = raw + (long)r * ld_raw + (long)h * dp + 4 * sub;
                const bf16_t* w = (isk ? (ex ? kw1 : kw0) : (ex ? qw1 : qw0)) + 4 * sub;
                float x1[4] = {0.f, 0.f, 0.f, 0.f}, x2[4] = {0.f, 0.f, 0.f, 0.f}, w1_[4] = {0.f, 0.f, 0.f, 0.f}, w2_[4] = {0.f, 0.f, 0.f, 0.f};
                if (live) { ld4(xr, x1); ld4(xr + half, x2); ld4(w, w1_); ld4(w + half, w2_); }
                float ss = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) ss += x1[i] * x1[i] + x2[i] * x2[i];
                const float rinv = rsqrtf(sum16(ss) / (float)hd + eps);
                float dot = 0.f, a1[4], a2[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    x1[i] *= rinv; x2[i] *= rinv;
                    n1[i] = bfround(n1[i]); n2[i] = bfround(n2[i]);              // the bf16 tensor between the two reverse ops
                    a1[i] = n1[i] * w1_[i]; a2[i] = n2[i] * w2_[i];
                    dot += a1[i] * x1[i] + a2[i] * x2[i];
                }
                dot = sum16(dot) / (float)hd;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float g1 = n1[i] * bfround(x1[i]), g2 = n2[i] * bfround(x2[i]);
                    if (ex) { if (isk) { dw[1][1][i] += g1; dw[1][1][4 + i] += g2; } else { dw[1][0][i] += g1; dw[1][0][4 + i] += g2; } }
                    else    { if (isk) { dw[0][1][i] += g1; dw[0][1][4 + i] += g2; } else { dw[0][0][i] += g1; dw[0][0][4 + i] += g2; } }
                    n1[i] = rinv * (a1[i] - x1[i] * dot);
                    n2[i] = rinv * (a2[i] - x2[i] * dot);
                }
            }
            if (live) {
                u32x2_t o1 = {pack2bf(n1[0], n1[1]), pack2bf(n1[2], n1[3])}, o2 = {pack2bf(n2[0], n2[1]), pack2bf(n2[2], n2[3])};
                *(u32x2_t*)d = o1;
                *(u32x2_t*)(d + half) = o2;
            }
        }
    }
    __syncthreads();
    if (on && use_norm) {
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    atomicAdd(&red[(e * 2 + k) * hd + 4 * sub + i], dw[e][k][i]);
                    atomicAdd(&red[(e * 2 + k) * hd + half + 4 * sub + i], dw[e][k][4 + i]);
                }
    }
    __syncthreads();
    float* p = partial + (long)blockIdx.x * 4 * hd;
    for (int c = threadIdx.x; c < 4 * hd; c += 256) p[c] = red[c];
}

extern "C" int bagel_qknorm_rope_bwd_bf16(void* dqkv, int64_t ld, const void* qkv_raw, int64_t ld_raw, const void* cos_tab, const void* sin_tab,
                                          const void* q_w0, const void* k_w0, const void* q_w1, const void* k_w1, const int32_t* expert_of_row,
                                          void* dqw0, void* dkw0, void* dqw1, void* dkw1, float* partial_ws, int32_t rows, int32_t nq,
                                          int32_t nkv, int32_t head_dim, int32_t head_dim_padded, float eps, int32_t use_norm,
                                          hipStream_t stream) {
    BAGEL_REQUIRE(dqkv && cos_tab && sin_tab && partial_ws, "qknorm_rope_bwd: null pointer");
    BAGEL_REQUIRE(head_dim % 8 == 0 && head_dim <= 128 && head_dim_padded >= head_dim && head_dim_padded % 4 == 0 && ld % 4 == 0 && ld_raw % 4 == 0,
                  "qknorm_rope_bwd: head_dim must be a multiple of 8, at most 128");
    BAGEL_REQUIRE(!use_norm || (qkv_raw && q_w0 && k_w0 && dqw0 && dkw0), "qknorm_rope_bwd: use_norm needs the raw projection, the weights and dqw0 / dkw0");
    BAGEL_REQUIRE((q_w1 == nullptr) == (k_w1 == nullptr) && (!use_norm || ((q_w1 == nullptr) == (dqw1 == nullptr) && (k_w1 == nullptr) == (dkw1 == nullptr))),
                  "qknorm_rope_bwd: the second expert's weights and gradient outputs go together");
    const int nblk = max(ceil_div(rows, 64), 0);
    if (nblk > 0) {
        hipLaunchKernelGGL(qknorm_rope_bwd_kernel, dim3(nblk), dim3(256), 0, stream, (bf16_t*)dqkv, (long)ld, (const bf16_t*)qkv_raw, (long)ld_raw,
                           (const bf16_t*)cos_tab, (const bf16_t*)sin_tab, (const bf16_t*)q_w0, (const bf16_t*)k_w0, (const bf16_t*)q_w1,
                           (const bf16_t*)k_w1, expert_of_row, partial_ws, rows, nq, nkv, head_dim, head_dim_padded, eps, use_norm);
        const int rc = bagel_check_launch("qknorm_rope_bwd_kernel");
        if (rc) return rc;
    }
    if (!use_norm) return BAGEL_OK;
    return colsum_finish(partial_ws, nblk, 4 * head_dim, head_dim, (bf16_t*)dqw0, (bf16_t*)dkw0, (bf16_t*)dqw1, (bf16_t*)dkw1, stream);
}

// ------------------------------------------------------------------------------------------------------------------------------
// SwiGLU reverse: act = bf16(bf16(silu(g)) * u)  (modeling_qwen2.py:201), gate / up interleaved in 16-column blocks.
// ------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(bf16_t* __restrict__ gu, long ld, const bf16_t* __restrict__ d_act, long ld_d, long rows,
                                                         int inter) {
    const int nch = inter >> 3;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * nch) return;
    const long r = i / nch;
    const int a0 = (int)(i - r * nch) * 8;
    bf16_t* gp = gu + r * ld + (a0 >> 4) * 32 + (a0 & 15);
    const u32x4_t gv = *(const u32x4_t*)gp, uv = *(const u32x4_t*)(gp + 16), dv = *(const u32x4_t*)(d_act + r * ld_d + a0);
    u32x4_t og, ou;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float dg[2], du[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float g = k ? hi2f(gv[e]) : lo2f(gv[e]), u = k ? hi2f(uv[e]) : lo2f(uv[e]), d = k ? hi2f(dv[e]) : lo2f(dv[e]);
            const float sg = 1.0f / (1.0f + __expf(-g));
            du[k] = d * bfround(g * sg);
            dg[k] = d * u * (sg * (1.0f + g * (1.0f - sg)));
        }
        og[e] = pack2bf(dg[0], dg[1]);
        ou[e] = pack2bf(du[0], du[1]);
    }
    *(u32x4_t*)gp = og;
    *(u32x4_t*)(gp + 16) = ou;
}

// The SwiGLU16 epilogue as a kernel of its own: act = bf16(bf16(silu(g)) * u) from the STORED bf16 gate/up projection -- bit-identical to
// the fused epilogue of bagel_gemm_bf16 (which rounds the accumulators to bf16 before the activation).  The training forward uses it when
// the tape keeps the un-activated projection (288 GB: 76 KB per token and layer) instead of recomputing it in the backward.
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const bf16_t* __restrict__ gu, long ld, bf16_t* __restrict__ act, long ld_a, long rows, int inter) {
    const int nch = inter >> 3;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * nch) return;
    const long r = i / nch;
    const int a0 = (int)(i - r * nch) * 8;
    const bf16_t* gp = gu + r * ld + (a0 >> 4) * 32 + (a0 & 15);
    const u32x4_t gv = *(const u32x4_t*)gp, uv = *(const u32x4_t*)(gp + 16);
    u32x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        o[e] = pack2bf(bfround(silu_f(lo2f(gv[e]))) * lo2f(uv[e]), bfround(silu_f(hi2f(gv[e]))) * hi2f(uv[e]));
    *(u32x4_t*)(act + r * ld_a + a0) = o;
}

extern "C" int bagel_swiglu_fwd_bf16(const void* gu, int64_t ld, void* act, int64_t ld_act, int64_t rows, int32_t inter, hipStream_t stream) {
    BAGEL_REQUIRE(gu && act, "swiglu_fwd: null pointer");
    BAGEL_REQUIRE(inter % 16 == 0 && ld % 8 == 0 && ld_act % 8 == 0, "swiglu_fwd: the intermediate size must be a multiple of 16");
    if (rows <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(ceil_div(rows * (inter / 8), 256)), dim3(256), 0, stream, (const bf16_t*)gu, (long)ld, (bf16_t*)act,
                       (long)ld_act, (long)rows, inter);
    return bagel_check_launch("swiglu_fwd_kernel");
}

extern "C" int bagel_swiglu_bwd_bf16(void* gu, int64_t ld, const void* d_act, int64_t ld_d, int64_t rows, int32_t inter, hipStream_t stream) {
    BAGEL_REQUIRE(gu && d_act, "swiglu_bwd: null pointer");
    BAGEL_REQUIRE(inter % 16 == 0 && ld % 8 == 0 && ld_d % 8 == 0, "swiglu_bwd: the intermediate size must be a multiple of 16");
    if (rows <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(ceil_div(rows * (inter / 8), 256)), dim3(256), 0, stream, (bf16_t*)gu, (long)ld, (const bf16_t*)d_act,
                       (long)ld_d, (long)rows, inter);
    return bagel_check_launch("swiglu_bwd_kernel");
}

// GELU-tanh (kind 1) / SiLU (kind 2) reverse: pre <- d_out * act'(pre)
__global__ __launch_bounds__(256) void act_bwd_kernel(bf16_t* __restrict__ pre, long ld, const bf16_t* __restrict__ d_out, long ld_d, long rows, int cols,
                                                      int kind) {
    const int nch = cols >> 3;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * nch) return;
    const long r = i / nch;
    const int c = (int)(i - r * nch) * 8;
    const u32x4_t xv = *(const u32x4_t*)(pre + r * ld + c), dv = *(const u32x4_t*)(d_out + r * ld_d + c);
    u32x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float res[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float x = k ? hi2f(xv[e]) : lo2f(xv[e]), d = k ? hi2f(dv[e]) : lo2f(dv[e]);
            float der;
            if (kind == 1) {
                const float k0 = 0.7978845608028654f, k1 = 0.044715f;
                const float t = tanhf(k0 * (x + k1 * x * x * x));
                der = 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * k0 * (1.0f + 3.0f * k1 * x * x);
            } else {
                const float sg = 1.0f / (1.0f + __expf(-x));
                der = sg * (1.0f + x * (1.0f - sg));
            }
            res[k] = d * der;
        }
        o[e] = pack2bf(res[0], res[1]);
    }
    *(u32x4_t*)(pre + r * ld + c) = o;
}

extern "C" int bagel_act_bwd_bf16(void* pre, int64_t ld, const void* d_out, int64_t ld_d, int64_t rows, int32_t cols, int32_t kind,
                                  hipStream_t stream) {
    BAGEL_REQUIRE(pre && d_out, "act_bwd: null pointer");
    BAGEL_REQUIRE(kind == 1 || kind == 2, "act_bwd: kind must be 1 (gelu_tanh) or 2 (silu)");
    BAGEL_REQUIRE(cols % 8 == 0 && ld % 8 == 0 && ld_d % 8 == 0, "act_bwd: cols / ld must be multiples of 8");
    if (rows <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(ceil_div(rows * (cols / 8), 256)), dim3(256), 0, stream, (bf16_t*)pre, (long)ld, (const bf16_t*)d_out,
                       (long)ld_d, (long)rows, cols, kind);
    return bagel_check_launch("act_bwd_kernel");
}

// ------------------------------------------------------------------------------------------------------------------------------
// loss heads
// ------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void cross_entropy_bwd_kernel(bf16_t* __restrict__ logits, long ld, const long* __restrict__ labels,
                                                                 const float* __restrict__ d_loss, int cols) {
    bf16_t* r = logits + (long)blockIdx.x * ld;
    const int tid = threadIdx.x;
    __shared__ float red[16];
    __shared__ float bc[2];
    float m = -INFINITY;
    for (int c = tid; c < cols; c += 1024) m = fmaxf(m, bf2f(r[c]));
    m = wave_max(m);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) m = fmaxf(m, red[w]);
    __syncthreads();
    float s = 0.f;
    for (int c = tid; c < cols; c += 1024) s += expf(bf2f(r[c]) - m);
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) tot += red[w];
        bc[0] = tot;
    }
    __syncthreads();
    const long lab = labels[blockIdx.x];
    const bool ok = lab >= 0 && lab < cols;
    const float inv = 1.0f / bc[0], dl = d_loss[blockIdx.x];
    for (int c = tid; c < cols; c += 1024) {
        const float p = expf(bf2f(r[c]) - m) * inv;
        r[c] = f2bf(ok ? (p - (c == lab ? 1.0f : 0.0f)) * dl : 0.0f);
    }
}

extern "C" int bagel_cross_entropy_bwd_bf16(void* logits, int64_t ld, const int64_t* labels, const float* d_loss, int32_t rows, int32_t cols,
                                            hipStream_t stream) {
    BAGEL_REQUIRE(logits && labels && d_loss && cols > 0, "cross_entropy_bwd: bad arguments");
    if (rows <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(cross_entropy_bwd_kernel, dim3(rows), dim3(1024), 0, stream, (bf16_t*)logits, (long)ld, (const long*)labels, d_loss, cols);
    return bagel_check_launch("cross_entropy_bwd_kernel");
}

__global__ __launch_bounds__(256) void mse_rows_bwd_kernel(const bf16_t* __restrict__ pred, long ld_pred, const float* __restrict__ noise,
                                                           const float* __restrict__ clean, const int* __restrict__ src_rows,
                                                           const float* __restrict__ d_loss, bf16_t* __restrict__ out, long ld_out, long n, int cols) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * cols) return;
    const long r = i / cols;
    const int c = (int)(i - r * cols);
    const long s = (long)src_rows[r] * cols + c;
    const float target = __fsub_rn(noise[s], clean[s]);
    out[r * ld_out + c] = f2bf(2.0f * (bf2f(pred[r * ld_pred + c]) - target) * d_loss[i]);
}

extern "C" int bagel_mse_rows_bwd_bf16(const void* pred, int64_t ld_pred, const float* noise, const float* clean, const int32_t* src_rows,
                                       const float* d_loss, void* d_pred, int64_t ld_d, int64_t n_rows, int32_t cols, hipStream_t stream) {
    BAGEL_REQUIRE(pred && noise && clean && src_rows && d_loss && d_pred, "mse_rows_bwd: null pointer");
    if (n_rows <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(mse_rows_bwd_kernel, dim3(ceil_div(n_rows * cols, 256)), dim3(256), 0, stream, (const bf16_t*)pred, (long)ld_pred, noise, clean,
                       src_rows, d_loss, (bf16_t*)d_pred, (long)ld_d, (long)n_rows, cols);
    return bagel_check_launch("mse_rows_bwd_kernel");
}

// ------------------------------------------------------------------------------------------------------------------------------
// gather reverse: dst[dst_rows[s]] = sum of src[order[i]] over segment s, in index order (deterministic)
// ------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rows_segment_sum_kernel(const bf16_t* __restrict__ src, long ld_src, const int* __restrict__ order,
                                                               const int* __restrict__ seg_off, const int* __restrict__ dst_rows,
                                                               bf16_t* __restrict__ dst, long ld_dst, int n_seg, int cols) {
    const int nch = cols >> 3;
    for (int s = blockIdx.x; s < n_seg; s += gridDim.x) {
        const int i0 = seg_off[s], i1 = seg_off[s + 1];
        bf16_t* d = dst + (long)dst_rows[s] * ld_dst;
        for (int c = threadIdx.x; c < nch; c += 256) {
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int i = i0; i < i1; ++i) {
                const u32x4_t v = *(const u32x4_t*)(src + (long)order[i] * ld_src + c * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc[2 * e] += lo2f(v[e]); acc[2 * e + 1] += hi2f(v[e]); }
            }
            u32x4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = pack2bf(acc[2 * e], acc[2 * e + 1]);
            *(u32x4_t*)(d + c * 8) = o;
        }
    }
}

extern "C" int bagel_rows_segment_sum_bf16(const void* src, int64_t ld_src, const int32_t* order, const int32_t* seg_off, const int32_t* dst_rows,
                                           void* dst, int64_t ld_dst, int32_t n_seg, int32_t cols, hipStream_t stream) {
    BAGEL_REQUIRE(src && order && seg_off && dst_rows && dst, "rows_segment_sum: null pointer");
    BAGEL_REQUIRE(cols % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0, "rows_segment_sum: cols / ld must be multiples of 8");
    if (n_seg <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(rows_segment_sum_kernel, dim3(min(n_seg, 8192)), dim3(256), 0, stream, (const bf16_t*)src, (long)ld_src, order, seg_off, dst_rows,
                       (bf16_t*)dst, (long)ld_dst, n_seg, cols);
    return bagel_check_launch("rows_segment_sum_kernel");
}
