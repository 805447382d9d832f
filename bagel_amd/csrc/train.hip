// Training-forward glue of Bagel.forward (bagel.py:101-229): the elementwise / row-reduction work around the backbone.
// All HBM-bound; fp32 arithmetic follows the reference's eager op order so that the per-token losses are reproducible.
//
//   bagel_flow_mix_bf16        x_t = (1 - t) * clean + t * noise (fp32, :187) cast to bf16 for vae2llm (autocast, :190)
//   bagel_flow_add_rows_bf16   seq[rows[i]] = bf16(bf16(seq[rows[i]] + temb[tid[i]]) + pos[pid[i]])  (:188-191; one
//                              timestep embedding per IMAGE -- the reference re-runs the time MLP for every token)
//   bagel_mse_rows_f32         (pred - (noise - clean))^2  (:214-217)
//   bagel_cross_entropy_bf16   F.cross_entropy(logits.float(), labels, reduction="none")  (:222)
#include "common.h"

__global__ __launch_bounds__(256) void flow_mix_kernel(const float* __restrict__ clean, const float* __restrict__ noise,
                                                       const float* __restrict__ t, bf16_t* __restrict__ out, long n, int cols) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * cols) return;
    const float tt = t[i / cols];
    const float v = __fadd_rn(__fmul_rn(__fsub_rn(1.0f, tt), clean[i]), __fmul_rn(tt, noise[i]));
    out[i] = f2bf(v);
}

extern "C" int bagel_flow_mix_bf16(const float* clean, const float* noise, const float* t, void* out, int64_t n_rows,
                                   int32_t cols, hipStream_t stream) {
    BAGEL_REQUIRE(clean && noise && t && out, "flow_mix: null pointer");
    if (n_rows <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(flow_mix_kernel, dim3(ceil_div(n_rows * cols, 256)), dim3(256), 0, stream, clean, noise, t, (bf16_t*)out,
                       (long)n_rows, cols);
    return bagel_check_launch("flow_mix_kernel");
}

__global__ __launch_bounds__(256) void flow_add_rows_kernel(bf16_t* __restrict__ seq, long ld, const int* __restrict__ rows,
                                                            const bf16_t* __restrict__ temb, long ld_temb, const int* __restrict__ temb_ids,
                                                            const bf16_t* __restrict__ pos_table, long ld_pos,
                                                            const long* __restrict__ pos_ids, int n, int cols) {
    const int lane = threadIdx.x & 63;
    const int nch = cols >> 3;
    for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += gridDim.x * 4) {
        bf16_t* s = seq + (long)rows[i] * ld;
        const bf16_t* tr = temb + (long)temb_ids[i] * ld_temb;
        const bf16_t* pr = pos_table + pos_ids[i] * ld_pos;
        for (int c = lane; c < nch; c += 64) {
            const u32x4_t a = *(const u32x4_t*)(s + c * 8);
            const u32x4_t t = *(const u32x4_t*)(tr + c * 8);
            const u32x4_t q = *(const u32x4_t*)(pr + c * 8);
            u32x4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                o[e] = pack2bf(bfround(lo2f(a[e]) + lo2f(t[e])) + lo2f(q[e]), bfround(hi2f(a[e]) + hi2f(t[e])) + hi2f(q[e]));
            *(u32x4_t*)(s + c * 8) = o;
        }
    }
}

extern "C" int bagel_flow_add_rows_bf16(void* seq, int64_t ld, const int32_t* rows, const void* temb, int64_t ld_temb,
                                        const int32_t* temb_ids, const void* pos_table, int64_t ld_pos, const int64_t* pos_ids,
                                        int32_t n, int32_t cols, hipStream_t stream) {
    BAGEL_REQUIRE(seq && rows && temb && temb_ids && pos_table && pos_ids, "flow_add_rows: null pointer");
    BAGEL_REQUIRE(cols % 8 == 0 && ld % 8 == 0 && ld_pos % 8 == 0 && ld_temb % 8 == 0, "flow_add_rows: cols/ld must be multiples of 8");
    if (n <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(flow_add_rows_kernel, dim3(min(ceil_div(n, 4), 4096)), dim3(256), 0, stream, (bf16_t*)seq, (long)ld, rows,
                       (const bf16_t*)temb, (long)ld_temb, temb_ids, (const bf16_t*)pos_table, (long)ld_pos, (const long*)pos_ids, n, cols);
    return bagel_check_launch("flow_add_rows_kernel");
}

__global__ __launch_bounds__(256) void mse_rows_kernel(const bf16_t* __restrict__ pred, long ld_pred, const float* __restrict__ noise,
                                                       const float* __restrict__ clean, const int* __restrict__ src_rows,
                                                       float* __restrict__ out, long n, int cols) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * cols) return;
    const long r = i / cols;
    const int c = (int)(i - r * cols);
    const long s = (long)src_rows[r] * cols + c;
    const float target = __fsub_rn(noise[s], clean[s]);
    const float d = __fsub_rn(bf2f(pred[r * ld_pred + c]), target);
    out[i] = __fmul_rn(d, d);
}

extern "C" int bagel_mse_rows_f32(const void* pred, int64_t ld_pred, const float* noise, const float* clean,
                                  const int32_t* src_rows, float* out, int64_t n_rows, int32_t cols, hipStream_t stream) {
    BAGEL_REQUIRE(pred && noise && clean && src_rows && out, "mse_rows: null pointer");
    if (n_rows <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(mse_rows_kernel, dim3(ceil_div(n_rows * cols, 256)), dim3(256), 0, stream, (const bf16_t*)pred, (long)ld_pred,
                       noise, clean, src_rows, out, (long)n_rows, cols);
    return bagel_check_launch("mse_rows_kernel");
}

// One workgroup per row: max, then sum of exp(x - max) (fp32 on the bf16 logits), loss = log(sum) + max - x[label].
__global__ __launch_bounds__(1024) void cross_entropy_kernel(const bf16_t* __restrict__ logits, long ld, const long* __restrict__ labels,
                                                             float* __restrict__ out, int cols) {
    const bf16_t* r = logits + (long)blockIdx.x * ld;
    const int tid = threadIdx.x;
    __shared__ float red[16];
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
        const long lab = labels[blockIdx.x];
        out[blockIdx.x] = (lab >= 0 && lab < cols) ? (logf(tot) + m) - bf2f(r[lab]) : 0.0f;   // out-of-range label: ignored (torch ignore_index)
    }
}

extern "C" int bagel_cross_entropy_bf16(const void* logits, int64_t ld, const int64_t* labels, float* out, int32_t rows,
                                        int32_t cols, hipStream_t stream) {
    BAGEL_REQUIRE(logits && labels && out && cols > 0, "cross_entropy: bad arguments");
    if (rows <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(cross_entropy_kernel, dim3(rows), dim3(1024), 0, stream, (const bf16_t*)logits, (long)ld, (const long*)labels, out, cols);
    return bagel_check_launch("cross_entropy_kernel");
}
