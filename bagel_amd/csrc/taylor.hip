// TaylorSeer feature cache (modeling/cache_utils/taylorseer.py:11-46; hooks qwen2_navit.py:824-829): two HBM-bound
// elementwise kernels over the [rows, cols] bf16 output of the last decoder layer.
//
//   full step    f'_0 = feature;  f'_{i+1} = bf16( bf16(f'_i - f_i) / distance )      i < n_diff   (derivative_approximation)
//   Taylor step  out  = sum_i bf16( bf16(c_i * f_i) * x^i ),  c_i = 1/i!, every partial sum rounded to bf16 (taylor_formula)
//
// The reference evaluates these as chains of eager bf16 tensor ops (one rounding per op); both kernels keep exactly those
// rounding points, so given the same cached features the result is bit-identical to the reference's, in ONE pass over
// HBM: 2 * (n_diff + 1) resp. (n + 1) row-streams of 16-byte lanes instead of ~3 n eager kernels.
#include "common.h"

#define TAYLOR_MAX_FACTORS 7   // max_order 6 (taylorseer.py:139) + the feature itself

struct TaylorBufs { bf16_t* f[TAYLOR_MAX_FACTORS]; };
struct TaylorCoef { float c[TAYLOR_MAX_FACTORS]; float xp[TAYLOR_MAX_FACTORS]; };

__global__ __launch_bounds__(256) void taylor_update_kernel(const bf16_t* __restrict__ feat, long ldf, TaylorBufs B, int n_diff,
                                                            float dist, long rows, int cols) {
    const int cpr = cols >> 3;                                   // 16-byte chunks per row
    const long nchunks = rows * cpr;
    for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < nchunks; c += (long)gridDim.x * 256) {
        const long r = c / cpr;
        const int k = (int)(c - r * cpr) * 8;
        u32x4_t cur = *(const u32x4_t*)(feat + r * ldf + k);
        const long off = r * cols + k;
        for (int i = 0; i < n_diff; ++i) {
            const u32x4_t old = *(const u32x4_t*)(B.f[i] + off);
            *(u32x4_t*)(B.f[i] + off) = cur;
            u32x4_t nxt;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dl = bfround(lo2f(cur[e]) - lo2f(old[e]));
                const float dh = bfround(hi2f(cur[e]) - hi2f(old[e]));
                nxt[e] = pack2bf(__fdiv_rn(dl, dist), __fdiv_rn(dh, dist));
            }
            cur = nxt;
        }
        *(u32x4_t*)(B.f[n_diff] + off) = cur;
    }
}

__global__ __launch_bounds__(256) void taylor_eval_kernel(TaylorBufs B, TaylorCoef C, int n, bf16_t* __restrict__ out, long ldo,
                                                          long rows, int cols) {
    const int cpr = cols >> 3;
    const long nchunks = rows * cpr;
    for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < nchunks; c += (long)gridDim.x * 256) {
        const long r = c / cpr;
        const int k = (int)(c - r * cpr) * 8;
        const long off = r * cols + k;
        const u32x4_t f0 = *(const u32x4_t*)(B.f[0] + off);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[2 * e] = lo2f(f0[e]); acc[2 * e + 1] = hi2f(f0[e]); }
        for (int i = 1; i < n; ++i) {
            const u32x4_t fi = *(const u32x4_t*)(B.f[i] + off);
            const float ci = C.c[i], xi = C.xp[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float tl = bfround(bfround(ci * lo2f(fi[e])) * xi);
                const float th = bfround(bfround(ci * hi2f(fi[e])) * xi);
                acc[2 * e] = bfround(acc[2 * e] + tl);
                acc[2 * e + 1] = bfround(acc[2 * e + 1] + th);
            }
        }
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack2bf(acc[2 * e], acc[2 * e + 1]);
        *(u32x4_t*)(out + r * ldo + k) = o;
    }
}

static int taylor_grid(long rows, int cols) {
    const long nchunks = rows * (cols >> 3);
    long g = (nchunks + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;     // grid-stride beyond 16 workgroups per CU
    return (int)(g < 1 ? 1 : g);
}

extern "C" int bagel_taylor_update_bf16(const void* feature, int64_t ld_feature, void* const* factors, int32_t n_diff,
                                        int32_t distance, int64_t rows, int32_t cols, hipStream_t stream) {
    BAGEL_REQUIRE(feature && factors, "taylor_update: null pointer");
    BAGEL_REQUIRE(n_diff >= 0 && n_diff < TAYLOR_MAX_FACTORS, "taylor_update: n_diff %d not in [0,%d)", n_diff, TAYLOR_MAX_FACTORS);
    BAGEL_REQUIRE(n_diff == 0 || distance != 0, "taylor_update: zero step distance");
    BAGEL_REQUIRE(cols > 0 && cols % 8 == 0 && ld_feature % 8 == 0, "taylor_update: cols/ld must be multiples of 8");
    TaylorBufs B;
    for (int i = 0; i < TAYLOR_MAX_FACTORS; ++i) B.f[i] = nullptr;
    for (int i = 0; i <= n_diff; ++i) {
        BAGEL_REQUIRE(factors[i] && (((uintptr_t)factors[i]) & 15) == 0, "taylor_update: factor buffer %d null or misaligned", i);
        B.f[i] = (bf16_t*)factors[i];
    }
    if (rows <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(taylor_update_kernel, dim3(taylor_grid(rows, cols)), dim3(256), 0, stream, (const bf16_t*)feature,
                       (long)ld_feature, B, n_diff, (float)distance, (long)rows, cols);
    return bagel_check_launch("taylor_update_kernel");
}

extern "C" int bagel_taylor_eval_bf16(void* const* factors, int32_t n, int32_t x, void* out, int64_t ld_out, int64_t rows,
                                      int32_t cols, hipStream_t stream) {
    BAGEL_REQUIRE(factors && out, "taylor_eval: null pointer");
    BAGEL_REQUIRE(n >= 1 && n <= TAYLOR_MAX_FACTORS, "taylor_eval: n %d not in [1,%d]", n, TAYLOR_MAX_FACTORS);
    BAGEL_REQUIRE(cols > 0 && cols % 8 == 0 && ld_out % 8 == 0, "taylor_eval: cols/ld must be multiples of 8");
    TaylorBufs B;
    TaylorCoef C;
    double fact = 1.0, xp = 1.0;
    for (int i = 0; i < TAYLOR_MAX_FACTORS; ++i) {
        B.f[i] = nullptr;
        if (i > 0) { fact *= (double)i; xp *= (double)x; }
        C.c[i] = (float)(1.0 / fact);     // python: (1 / math.factorial(i)) as the fp32 scalar of a bf16 tensor op
        C.xp[i] = (float)xp;              // python: x ** i
    }
    for (int i = 0; i < n; ++i) {
        BAGEL_REQUIRE(factors[i] && (((uintptr_t)factors[i]) & 15) == 0, "taylor_eval: factor buffer %d null or misaligned", i);
        B.f[i] = (bf16_t*)factors[i];
    }
    if (rows <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(taylor_eval_kernel, dim3(taylor_grid(rows, cols)), dim3(256), 0, stream, B, C, n, (bf16_t*)out, (long)ld_out,
                       (long)rows, cols);
    return bagel_check_launch("taylor_eval_kernel");
}
