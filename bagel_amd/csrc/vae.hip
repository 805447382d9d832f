// FLUX-style VAE kernels for gfx950, fp32 end to end (the reference keeps its VAE in fp32: app.py:48,138,
// eval/gen/gen_images_mp.py:93).  Activations are NHWC so the contraction dimension (channels) is contiguous.
//
//   bagel_conv_gemm_f32     implicit-GEMM convolution / plain GEMM on the exact-fp32 MFMA
//                           (v_mfma_f32_32x32x2_f32 == an fmaf chain, cdna_hip_programming.md section 3):
//                           replaces F.conv2d 3x3 / 1x1 (autoencoder.py:76-78,102,114,139,170,221,248), the
//                           stride-2 downsample with its asymmetric pad (:104-107), nearest-2x upsample + conv
//                           (:116-118, fused: the loader reads in[(y+dy)>>1][(x+dx)>>1]) and the q k^T / p v
//                           products of the single-head AttnBlock (:49-62).
//   bagel_groupnorm_f32     GroupNorm(32, eps 1e-6, affine) (+ swish) (autoencoder.py:43,75,77,169,247)
//   bagel_softmax_rows_f32  softmax(scale * s) in place
//   bagel_vae_reparam_f32   DiagonalGaussian sample + latent scale/shift (autoencoder.py:280-287, 315-318)
//   bagel_vae_unscale_f32   z / scale + shift (autoencoder.py:321)
#include "common.h"

__device__ __forceinline__ void glds16v(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

struct ConvParams {
    const float* in;     // [B, Hin, Win, Cin] (mode 0: [M, K] rows with stride ld_in)
    const float* w;      // [Cout, taps*Cin]
    const float* bias;   // [Cout] or null
    const float* res;    // residual, same layout as out, or null
    float* out;          // [M, ld_out]
    long ld_in, ld_w, ld_out;
    int B, Hin, Win, Cin, Hout, Wout, Cout;
    int M, K;            // M = B*Hout*Wout, K = taps*Cin
    int mode;            // 0 rows, 1 conv3 s1 p1, 2 conv3 s2 pad(0,1,0,1), 3 nearest-2x upsample + conv3 s1 p1
    int tiles_m, tiles_n;
};

// Tile: 128 (pixels) x 128 (cout) x 32 (k, fp32 = 128-byte rows), 4 waves (2x2), each wave 64x64 = 2x2 blocks of 32x32.
// Operands are swapped (D = Wfrag * Afrag): a lane owns 4 consecutive cout of one pixel -> 16-byte stores.
__global__ __launch_bounds__(256) void conv_gemm_f32_kernel(const ConvParams p) {
    constexpr int BM = 128, BN = 128, A_BYTES = BM * 128, STAGE = (BM + BN) * 128, NW = 4, LA = BM / 8 / NW, LB = BN / 8 / NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nblk = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tn = bid % p.tiles_n, tm = bid / p.tiles_n;   // consecutive blocks share the A (pixel) panel
    const int m0 = tm * BM, n0 = tn * BN;
    const char* zsrc = (const char*)bagel_zero16;

    // per-lane A rows (pixels): decode once
    int ab[LA], ay[LA], ax[LA], ach[LA];
    const char* arow[LA];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int j = wave + i * NW;
        const int row = j * 8 + (lane >> 3);
        ach[i] = ((lane & 7) ^ ((row >> 1) & 7)) * 4;   // channel offset (fp32 elements) of this lane's 16-byte chunk
        int m = m0 + row;
        m = m < p.M ? m : p.M - 1;
        if (p.mode == 0) {
            arow[i] = (const char*)(p.in + (long)m * p.ld_in);
            ab[i] = ay[i] = ax[i] = 0;
        } else {
            const int hw = p.Hout * p.Wout;
            ab[i] = m / hw;
            const int r2 = m - ab[i] * hw;
            ay[i] = r2 / p.Wout;
            ax[i] = r2 - ay[i] * p.Wout;
            arow[i] = nullptr;
        }
    }
    const char* pb[LB];
    int bch[LB];
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const int j = wave + i * NW;
        const int row = j * 8 + (lane >> 3);
        bch[i] = ((lane & 7) ^ ((row >> 1) & 7)) * 4;
        int n = n0 + row;
        n = n < p.Cout ? n : p.Cout - 1;
        pb[i] = (const char*)(p.w + (long)n * p.ld_w);
    }

    const int nk = (p.K + 31) >> 5;
    auto issue = [&](int stage, int kt) {
        char* sb = smem + stage * STAGE;
        const int k0 = kt * 32;
        int dy = 0, dx = 0, c0 = k0;
        if (p.mode != 0) {
            const int tap = k0 / p.Cin;      // Cin % 32 == 0 -> a k-tile never straddles taps
            c0 = k0 - tap * p.Cin;
            dy = tap / 3;
            dx = tap - dy * 3;
        }
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const char* src;
            if (p.mode == 0) {
                src = (k0 + ach[i] < p.K) ? arow[i] + (long)(k0 + ach[i]) * 4 : zsrc;
            } else {
                int yy, xx;
                bool ok;
                if (p.mode == 1) { yy = ay[i] + dy - 1; xx = ax[i] + dx - 1; ok = yy >= 0 && yy < p.Hin && xx >= 0 && xx < p.Win; }
                else if (p.mode == 2) { yy = 2 * ay[i] + dy; xx = 2 * ax[i] + dx; ok = yy < p.Hin && xx < p.Win; }
                else { yy = ay[i] + dy - 1; xx = ax[i] + dx - 1; ok = yy >= 0 && yy < p.Hout && xx >= 0 && xx < p.Wout; yy >>= 1; xx >>= 1; }
                src = ok ? (const char*)(p.in + (((long)ab[i] * p.Hin + yy) * p.Win + xx) * p.Cin + c0 + ach[i]) : zsrc;
            }
            glds16v(src, sb + (wave + i * NW) * 1024);
        }
#pragma unroll
        for (int i = 0; i < LB; ++i)
            glds16v((k0 + bch[i] < p.K) ? pb[i] + (long)(k0 + bch[i]) * 4 : zsrc, sb + A_BYTES + (wave + i * NW) * 1024);
    };

    // fragment reads: lane -> row (lane&31) of a 32-row block, 16-byte chunk 2*s + (lane>>5), s = 0..3
    const int fr = lane & 31, hi = lane >> 5;
    const int sw = (fr >> 1) & 7;
    const int a_off = (wm * 64 + fr) * 128;
    const int b_off = A_BYTES + (wn * 64 + fr) * 128;

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int st = kt & 1;
        if (kt + 1 < nk) {
            issue(st ^ 1, kt + 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LA + LB) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_barrier" ::: "memory");
        const char* sb = smem + st * STAGE;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int ch = (2 * s + hi) ^ sw;
            f32x4_t af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *(const f32x4_t*)(sb + a_off + i * 4096 + ch * 16);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = *(const f32x4_t*)(sb + b_off + j * 4096 + ch * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j][e], af[i][e], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    // epilogue: D row = cout (8*(r>>2) + 4*hi + (r&3)), D col = pixel (lane&31)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm * 64 + i * 32 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int n = n0 + wn * 64 + j * 32 + 8 * u + 4 * hi;
                if (n >= p.Cout) continue;
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = acc[i][j][4 * u + e];
                float* dst = p.out + (long)m * p.ld_out + n;
                if (n + 3 < p.Cout) {
                    if (p.bias) { const f32x4_t b = *(const f32x4_t*)(p.bias + n); o[0] += b[0]; o[1] += b[1]; o[2] += b[2]; o[3] += b[3]; }
                    if (p.res) { const f32x4_t r = *(const f32x4_t*)(p.res + (long)m * p.ld_out + n); o[0] += r[0]; o[1] += r[1]; o[2] += r[2]; o[3] += r[3]; }
                    *(f32x4_t*)dst = (f32x4_t){o[0], o[1], o[2], o[3]};
                } else {
                    for (int e = 0; e < 4 && n + e < p.Cout; ++e) {
                        float x = o[e];
                        if (p.bias) x += p.bias[n + e];
                        if (p.res) x += p.res[(long)m * p.ld_out + n + e];
                        dst[e] = x;
                    }
                }
            }
    }
}

extern "C" int bagel_conv_gemm_f32(const float* in, int64_t ld_in, const float* w, int64_t ld_w, const float* bias,
                                   const float* residual, float* out, int64_t ld_out, int32_t B, int32_t Hin, int32_t Win,
                                   int32_t Cin, int32_t Hout, int32_t Wout, int32_t Cout, int32_t mode, hipStream_t stream) {
    BAGEL_REQUIRE(in && w && out, "conv_gemm: null pointer");
    BAGEL_REQUIRE(mode >= 0 && mode <= 3, "conv_gemm: bad mode %d", mode);
    BAGEL_REQUIRE(Cin % 4 == 0 && ld_w % 4 == 0 && ld_out % 4 == 0 && ld_in % 4 == 0, "conv_gemm: channel counts / strides must be multiples of 4");
    BAGEL_REQUIRE(mode == 0 || Cin % 32 == 0, "conv_gemm: 3x3 modes need Cin %% 32 == 0 (pad the channels), got %d", Cin);
    ConvParams p;
    p.in = in; p.w = w; p.bias = bias; p.res = residual; p.out = out;
    p.ld_in = ld_in; p.ld_w = ld_w; p.ld_out = ld_out;
    p.B = B; p.Hin = Hin; p.Win = Win; p.Cin = Cin; p.Hout = Hout; p.Wout = Wout; p.Cout = Cout;
    p.M = B * Hout * Wout;
    p.K = (mode == 0 ? 1 : 9) * Cin;
    p.mode = mode;
    if (p.M <= 0 || Cout <= 0) return BAGEL_OK;
    p.tiles_m = ceil_div(p.M, 128);
    p.tiles_n = ceil_div(Cout, 128);
    constexpr int smem = 2 * 256 * 128;
    if (int rc = bagel_enable_lds((const void*)conv_gemm_f32_kernel, smem, "conv_gemm_f32_kernel")) return rc;
    hipLaunchKernelGGL(conv_gemm_f32_kernel, dim3(p.tiles_m * p.tiles_n), dim3(256), smem, stream, p);
    return bagel_check_launch("conv_gemm_f32_kernel");
}

// ---------------------------------------------------------------------------------------------------------
// GroupNorm over NHWC fp32: stats per (image, group) with a shifted two-moment accumulation (pivot = first element of
// the group, common to every partial -> partials add exactly; no catastrophic cancellation), fixed-order reduction.
// ---------------------------------------------------------------------------------------------------------
#define GN_SLICES 64
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, float* __restrict__ partial, int HW, int C, int G) {
    const int b = blockIdx.z, g = blockIdx.y, sl = blockIdx.x;
    const int cpg = C / G;
    const float* xb = x + (long)b * HW * C + g * cpg;
    const float pivot = xb[0];
    const long n = (long)HW * cpg;
    const long per = (n + GN_SLICES - 1) / GN_SLICES;
    const long lo = sl * per, hi = min(n, lo + per);
    float s1 = 0.f, s2 = 0.f;
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
        const long pix = i / cpg;
        const int c = (int)(i - pix * cpg);
        const float d = xb[pix * C + c] - pivot;
        s1 += d; s2 += d * d;
    }
    __shared__ float r1[256], r2[256];
    r1[threadIdx.x] = s1; r2[threadIdx.x] = s2;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) { r1[threadIdx.x] += r1[threadIdx.x + s]; r2[threadIdx.x] += r2[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float* o = partial + (((long)b * G + g) * GN_SLICES + sl) * 2;
        o[0] = r1[0]; o[1] = r2[0];
    }
}

// (mean, rstd) per (image, group) from the slice partials, fixed order.
__global__ void gn_finalize_kernel(const float* __restrict__ x, const float* __restrict__ partial, float* __restrict__ stats, int HW,
                                   int C, int G, float eps, int BG) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BG) return;
    const int b = i / G, g = i - b * G;
    const int cpg = C / G;
    const float* pp = partial + (long)i * GN_SLICES * 2;
    float s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < GN_SLICES; ++k) { s1 += pp[2 * k]; s2 += pp[2 * k + 1]; }
    const float n = (float)HW * (float)cpg;
    const float pivot = x[(long)b * HW * C + g * cpg];
    const float md = s1 / n;
    const float var = fmaxf(s2 / n - md * md, 0.f);
    stats[2 * i] = pivot + md;
    stats[2 * i + 1] = rsqrtf(var + eps);
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, int HW, int C, int G,
                                                       int swish) {
    const int b = blockIdx.y;
    const int cpg = C / G;
    const long total4 = (long)HW * C / 4;
    for (long i4 = (long)blockIdx.x * 256 + threadIdx.x; i4 < total4; i4 += (long)gridDim.x * 256) {
        const long i = i4 * 4;
        const int c = (int)(i % C);
        const long off = (long)b * HW * C + i;
        const f32x4_t v = *(const f32x4_t*)(x + off);
        f32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int g = (c + e) / cpg;
            const float mean = stats[2 * (b * G + g)], inv = stats[2 * (b * G + g) + 1];
            float t = (v[e] - mean) * inv * gamma[c + e] + beta[c + e];
            if (swish) t = t * (1.0f / (1.0f + __expf(-t)));
            o[e] = t;
        }
        *(f32x4_t*)(y + off) = o;
    }
}

extern "C" int bagel_groupnorm_f32(const float* x, float* y, float* partial_ws, const float* gamma, const float* beta, int32_t B,
                                   int32_t HW, int32_t C, int32_t groups, float eps, int32_t swish, hipStream_t stream) {
    BAGEL_REQUIRE(x && y && partial_ws && gamma && beta, "groupnorm: null pointer");
    BAGEL_REQUIRE(C % groups == 0 && C % 4 == 0, "groupnorm: C=%d must divide into %d groups and be a multiple of 4", C, groups);
    if (B <= 0 || HW <= 0) return BAGEL_OK;
    // workspace: [B*G*GN_SLICES*2] partials followed by [B*G*2] stats
    float* stats = partial_ws + (long)B * groups * GN_SLICES * 2;
    hipLaunchKernelGGL(gn_stats_kernel, dim3(GN_SLICES, groups, B), dim3(256), 0, stream, x, partial_ws, HW, C, groups);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(ceil_div(B * groups, 64)), dim3(64), 0, stream, x, partial_ws, stats, HW, C, groups, eps, B * groups);
    const long total4 = (long)HW * C / 4;
    const int blocks = (int)min((long)ceil_div(total4, 256), 2048L);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(blocks, B), dim3(256), 0, stream, x, y, stats, gamma, beta, HW, C, groups, swish);
    return bagel_check_launch("groupnorm kernels");
}

// softmax over rows, in place: x = softmax(scale * x).  One block per row.
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ x, long ld, int cols, float scale) {
    float* r = x + (long)blockIdx.x * ld;
    __shared__ float red[256];
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < cols; c += 256) mx = fmaxf(mx, r[c]);
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]); __syncthreads(); }
    mx = red[0];
    __syncthreads();
    float sum = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) { const float e = __expf((r[c] - mx) * scale); r[c] = e; sum += e; }
    red[threadIdx.x] = sum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    const float inv = 1.0f / red[0];
    for (int c = threadIdx.x; c < cols; c += 256) r[c] *= inv;
}

extern "C" int bagel_softmax_rows_f32(float* x, int64_t ld, int32_t rows, int32_t cols, float scale, hipStream_t stream) {
    BAGEL_REQUIRE(x && cols > 0, "softmax_rows: bad arguments");
    if (rows <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, stream, x, (long)ld, cols, scale);
    return bagel_check_launch("softmax_rows_kernel");
}

// z = scale * ((mean + exp(0.5 * logvar) * noise) - shift); moments NHWC [n_pix, 2*zc] (mean | logvar), noise/out [n_pix, zc]
__global__ void vae_reparam_kernel(const float* __restrict__ mom, const float* __restrict__ noise, float* __restrict__ z, long n_pix,
                                   int zc, float scale, float shift) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pix * zc) return;
    const long pix = i / zc;
    const int c = (int)(i - pix * zc);
    const float mean = mom[pix * 2 * zc + c], logvar = mom[pix * 2 * zc + zc + c];
    const float stdv = expf(0.5f * logvar);
    const float s = mean + stdv * (noise ? noise[i] : 0.f);
    z[i] = scale * (s - shift);
}

extern "C" int bagel_vae_reparam_f32(const float* moments, const float* noise, float* z, int64_t n_pix, int32_t z_channels,
                                     float scale, float shift, hipStream_t stream) {
    BAGEL_REQUIRE(moments && z, "vae_reparam: null pointer");
    const long n = n_pix * z_channels;
    if (n <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(vae_reparam_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, stream, moments, noise, z, (long)n_pix, z_channels, scale, shift);
    return bagel_check_launch("vae_reparam_kernel");
}

__global__ void vae_unscale_kernel(const float* __restrict__ z, float* __restrict__ out, long n, float scale, float shift) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = z[i] / scale + shift;
}

extern "C" int bagel_vae_unscale_f32(const float* z, float* out, int64_t n, float scale, float shift, hipStream_t stream) {
    BAGEL_REQUIRE(z && out, "vae_unscale: null pointer");
    if (n <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(vae_unscale_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, stream, z, out, (long)n, scale, shift);
    return bagel_check_launch("vae_unscale_kernel");
}

// =========================================================================================================================================
// The VAE under the inferencer's bf16 autocast (inferencer.py:233 -> decode_image :174-185; forward_cache_update_vae bagel.py:491-550):
// torch.autocast("cuda", bf16) runs every conv2d and the SDPA of the AttnBlock in bf16 (inputs, weights AND bias cast to bf16, fp32
// accumulation, bf16 result), keeps group_norm in fp32 (bf16 input cast up, fp32 result, swish on it) and adds residuals in the dtype of
// their operands (bf16).  oracle/bagel_oracle.py VAE_AUTOCAST = "cuda" states it; tests/golden/vae_full_bf16.pt pins it.
//
//   bagel_conv_gemm_bf16    NHWC bf16 implicit-GEMM convolution / plain GEMM on mfma_f32_16x16x32_bf16 (16x the fp32 MFMA's rate): same
//                           modes, tap-in-the-DMA-address loader and fused nearest-2x upsample as bagel_conv_gemm_f32; 128 x 128 x 64 tile
//                           (a 64-channel k-tile = one 128-byte LDS row; inside one tap when Cin % 64 == 0, per-chunk taps otherwise), 4 waves (2 x 2), swapped
//                           operands so a lane owns 4 consecutive cout of one pixel; epilogue bf16(acc + bias) [+ residual, rounded again],
//                           or raw fp32 (the attention scores).
//   bagel_groupnorm_bf16    GroupNorm(32) (+ swish) of a bf16 NHWC tensor in fp32, result rounded ONCE to bf16 (= the next conv's input cast)
//   bagel_softmax_rows_bf16 softmax(scale * s) of fp32 score rows -> bf16 probabilities (the P operand of the second product)
// =========================================================================================================================================
struct ConvBf16Params {
    const bf16_t* in;    // [B, Hin, Win, Cin] (mode 0: [M, K] rows with stride ld_in)
    const bf16_t* w;     // [Cout, taps*Cin]
    const bf16_t* bias;  // [Cout] or null
    const bf16_t* res;   // residual, same layout as out, or null
    void* out;           // [M, ld_out] bf16 (OUT_F32: fp32)
    long ld_in, ld_w, ld_out;
    int B, Hin, Win, Cin, Hout, Wout, Cout;
    int M, K;
    int mode;
    int tiles_m, tiles_n;
};

__device__ __forceinline__ void glds16b(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <bool OUT_F32>
__global__ __launch_bounds__(256) void conv_gemm_bf16_kernel(const ConvBf16Params p) {
    constexpr int BM = 128, BN = 128, A_BYTES = BM * 128, STAGE = (BM + BN) * 128, NW = 4, LA = BM / 8 / NW, LB = BN / 8 / NW;
    constexpr int MB = 4, NB = 4;                    // 16-row fragments per wave (64 x 64 per wave)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nblk = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tn = bid % p.tiles_n, tm = bid / p.tiles_n;   // consecutive blocks share the A (pixel) panel
    const int m0 = tm * BM, n0 = tn * BN;
    const char* zsrc = (const char*)bagel_zero16;

    int ab[LA], ay[LA], ax[LA], ach[LA];
    const char* arow[LA];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int j = wave + i * NW;
        const int row = j * 8 + (lane >> 3);
        ach[i] = ((lane & 7) ^ ((row >> 1) & 7)) * 8;   // channel offset (bf16 elements) of this lane's 16-byte chunk
        int m = m0 + row;
        m = m < p.M ? m : p.M - 1;
        if (p.mode == 0) {
            arow[i] = (const char*)(p.in + (long)m * p.ld_in);
            ab[i] = ay[i] = ax[i] = 0;
        } else {
            const int hw = p.Hout * p.Wout;
            ab[i] = m / hw;
            const int r2 = m - ab[i] * hw;
            ay[i] = r2 / p.Wout;
            ax[i] = r2 - ay[i] * p.Wout;
            arow[i] = nullptr;
        }
    }
    const char* pb[LB];
    int bch[LB];
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const int j = wave + i * NW;
        const int row = j * 8 + (lane >> 3);
        bch[i] = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
        int n = n0 + row;
        n = n < p.Cout ? n : p.Cout - 1;
        pb[i] = (const char*)(p.w + (long)n * p.ld_w);
    }

    const int nk = (p.K + 63) >> 6;
    auto issue = [&](int stage, int kt) {
        char* sb = smem + stage * STAGE;
        const int k0 = kt * 64;
        // Cin % 64 == 0 (every layer of the real VAE behind conv_in): a k-tile lies inside ONE tap, decoded once per k-tile on the scalar unit;
        // otherwise (small test models, Cin % 8 == 0) every 8-channel chunk finds its own tap
        const bool uni = (p.Cin & 63) == 0;
        int dy = 0, dx = 0, c0 = k0;
        if (p.mode != 0 && uni) {
            const int tap = k0 / p.Cin;
            c0 = k0 - tap * p.Cin;
            dy = tap / 3;
            dx = tap - dy * 3;
        }
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const char* src;
            if (p.mode == 0) {
                src = (k0 + ach[i] < p.K) ? arow[i] + (long)(k0 + ach[i]) * 2 : zsrc;
            } else {
                int yy, xx, cc = c0 + ach[i], dyi = dy, dxi = dx;
                bool ok = true;
                if (!uni) {
                    const int k = k0 + ach[i];
                    const int tap = k / p.Cin;
                    cc = k - tap * p.Cin;
                    dyi = tap / 3;
                    dxi = tap - dyi * 3;
                    ok = k < p.K;
                }
                if (p.mode == 1) { yy = ay[i] + dyi - 1; xx = ax[i] + dxi - 1; ok = ok && yy >= 0 && yy < p.Hin && xx >= 0 && xx < p.Win; }
                else if (p.mode == 2) { yy = 2 * ay[i] + dyi; xx = 2 * ax[i] + dxi; ok = ok && yy < p.Hin && xx < p.Win; }
                else { yy = ay[i] + dyi - 1; xx = ax[i] + dxi - 1; ok = ok && yy >= 0 && yy < p.Hout && xx >= 0 && xx < p.Wout; yy >>= 1; xx >>= 1; }
                src = ok ? (const char*)(p.in + (((long)ab[i] * p.Hin + yy) * p.Win + xx) * p.Cin + cc) : zsrc;
            }
            glds16b(src, sb + (wave + i * NW) * 1024);
        }
#pragma unroll
        for (int i = 0; i < LB; ++i)
            glds16b((k0 + bch[i] < p.K) ? pb[i] + (long)(k0 + bch[i]) * 2 : zsrc, sb + A_BYTES + (wave + i * NW) * 1024);
    };

    // fragment reads (gemm.hip gemm_tn_kernel): lane -> row lane & 15 of a 16-row block, 8 k-elements at chunk lane / 16 + 4 kh
    const int fr = lane & 15;
    const int sw = fr >> 1;
    const int ch0 = (lane >> 4) ^ sw;
    const int ch1 = ((lane >> 4) + 4) ^ sw;
    const int a_off = (wm * 64 + fr) * 128;
    const int b_off = A_BYTES + (wn * 64 + fr) * 128;

    f32x4_t acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int st = kt & 1;
        if (kt + 1 < nk) {
            issue(st ^ 1, kt + 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LA + LB) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_barrier" ::: "memory");
        const char* sb = smem + st * STAGE;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const int ch = kh ? ch1 : ch0;
            bf16x8_t af[MB], bfr[NB];
#pragma unroll
            for (int i = 0; i < MB; ++i) af[i] = *(const bf16x8_t*)(sb + a_off + i * 2048 + ch * 16);
#pragma unroll
            for (int j = 0; j < NB; ++j) bfr[j] = *(const bf16x8_t*)(sb + b_off + j * 2048 + ch * 16);
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    // epilogue: lane owns out[pixel m][cout n .. n + 3], n = fragment base + (lane >> 4) * 4
    const int nsub = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int m = m0 + wm * 64 + i * 16 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int n = n0 + wn * 64 + j * 16 + nsub;
            if (n >= p.Cout) continue;
            float o[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (OUT_F32) {
                float* dst = (float*)p.out + (long)m * p.ld_out + n;
                if (n + 3 < p.Cout) *(f32x4_t*)dst = (f32x4_t){o[0], o[1], o[2], o[3]};
                else for (int e = 0; e < 4 && n + e < p.Cout; ++e) dst[e] = o[e];
                continue;
            }
            bf16_t* dst = (bf16_t*)p.out + (long)m * p.ld_out + n;
            if (n + 3 < p.Cout) {
                if (p.bias) {
                    const u32x2_t bv = *(const u32x2_t*)(p.bias + n);
                    o[0] += lo2f(bv[0]); o[1] += hi2f(bv[0]); o[2] += lo2f(bv[1]); o[3] += hi2f(bv[1]);
                }
                if (p.res) {
                    const u32x2_t rv = *(const u32x2_t*)(p.res + (long)m * p.ld_out + n);
                    o[0] = bfround(o[0]) + lo2f(rv[0]); o[1] = bfround(o[1]) + hi2f(rv[0]);
                    o[2] = bfround(o[2]) + lo2f(rv[1]); o[3] = bfround(o[3]) + hi2f(rv[1]);
                }
                *(u32x2_t*)dst = (u32x2_t){pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
            } else {
                for (int e = 0; e < 4 && n + e < p.Cout; ++e) {
                    float x = o[e];
                    if (p.bias) x += bf2f(p.bias[n + e]);
                    if (p.res) x = bfround(x) + bf2f(p.res[(long)m * p.ld_out + n + e]);
                    dst[e] = f2bf(x);
                }
            }
        }
    }
}

extern "C" int bagel_conv_gemm_bf16(const void* in, int64_t ld_in, const void* w, int64_t ld_w, const void* bias, const void* residual,
                                    void* out, int64_t ld_out, int32_t out_f32, int32_t B, int32_t Hin, int32_t Win, int32_t Cin, int32_t Hout,
                                    int32_t Wout, int32_t Cout, int32_t mode, hipStream_t stream) {
    BAGEL_REQUIRE(in && w && out, "conv_gemm_bf16: null pointer");
    BAGEL_REQUIRE(mode >= 0 && mode <= 3, "conv_gemm_bf16: bad mode %d", mode);
    BAGEL_REQUIRE(Cin % 8 == 0 && ld_w % 8 == 0 && ld_in % 8 == 0 && ld_out % 4 == 0, "conv_gemm_bf16: channel counts / strides must keep 16-byte chunks (8-byte output rows)");
    BAGEL_REQUIRE(!out_f32 || (!bias && !residual), "conv_gemm_bf16: the fp32 output form takes no bias / residual");
    BAGEL_REQUIRE((((uintptr_t)in | (uintptr_t)w) & 15) == 0 && (((uintptr_t)out | (uintptr_t)bias | (uintptr_t)residual) & 7) == 0, "conv_gemm_bf16: alignment");
    ConvBf16Params p;
    p.in = (const bf16_t*)in; p.w = (const bf16_t*)w; p.bias = (const bf16_t*)bias; p.res = (const bf16_t*)residual; p.out = out;
    p.ld_in = ld_in; p.ld_w = ld_w; p.ld_out = ld_out;
    p.B = B; p.Hin = Hin; p.Win = Win; p.Cin = Cin; p.Hout = Hout; p.Wout = Wout; p.Cout = Cout;
    p.M = B * Hout * Wout;
    p.K = (mode == 0 ? 1 : 9) * Cin;
    p.mode = mode;
    if (p.M <= 0 || Cout <= 0) return BAGEL_OK;
    p.tiles_m = ceil_div(p.M, 128);
    p.tiles_n = ceil_div(Cout, 128);
    constexpr int smem = 2 * 256 * 128;
    if (out_f32) {
        if (int rc = bagel_enable_lds((const void*)conv_gemm_bf16_kernel<true>, smem, "conv_gemm_bf16_kernel")) return rc;
        hipLaunchKernelGGL(conv_gemm_bf16_kernel<true>, dim3(p.tiles_m * p.tiles_n), dim3(256), smem, stream, p);
    } else {
        if (int rc = bagel_enable_lds((const void*)conv_gemm_bf16_kernel<false>, smem, "conv_gemm_bf16_kernel")) return rc;
        hipLaunchKernelGGL(conv_gemm_bf16_kernel<false>, dim3(p.tiles_m * p.tiles_n), dim3(256), smem, stream, p);
    }
    return bagel_check_launch("conv_gemm_bf16_kernel");
}

#define GN_BF16_SLICES 1024
// ---- GroupNorm of a bf16 NHWC tensor: fp32 statistics (same shifted two-moment scheme and fixed-order reduction as the fp32 kernels) ----
__global__ __launch_bounds__(256) void gn_stats_bf16_kernel(const bf16_t* __restrict__ x, float* __restrict__ partial, int HW, int C, int G, int S) {
    const int b = blockIdx.z, g = blockIdx.y, sl = blockIdx.x;
    const int cpg = C / G;
    const bf16_t* xb = x + (long)b * HW * C + g * cpg;
    const float pivot = bf2f(xb[0]);
    const long n = (long)HW * cpg;
    const long per = (n + S - 1) / S;
    const long lo = sl * per, hi = min(n, lo + per);
    float s1 = 0.f, s2 = 0.f;
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
        const long pix = i / cpg;
        const int c = (int)(i - pix * cpg);
        const float d = bf2f(xb[pix * C + c]) - pivot;
        s1 += d; s2 += d * d;
    }
    __shared__ float r1[256], r2[256];
    r1[threadIdx.x] = s1; r2[threadIdx.x] = s2;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) { r1[threadIdx.x] += r1[threadIdx.x + s]; r2[threadIdx.x] += r2[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float* o = partial + (((long)b * G + g) * S + sl) * 2;
        o[0] = r1[0]; o[1] = r2[0];
    }
}

// one wave per (image, group): lane l sums slices l, l + 64, ...; the 64 lane sums meet in a fixed butterfly order
__global__ __launch_bounds__(64) void gn_finalize_bf16_kernel(const bf16_t* __restrict__ x, const float* __restrict__ partial, float* __restrict__ stats,
                                                              int HW, int C, int G, float eps, int BG, int S) {
    const int i = blockIdx.x;
    const int b = i / G, g = i - b * G;
    const int cpg = C / G;
    const float* pp = partial + (long)i * S * 2;
    float s1 = 0.f, s2 = 0.f;
    for (int k = threadIdx.x; k < S; k += 64) { s1 += pp[2 * k]; s2 += pp[2 * k + 1]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if (threadIdx.x) return;
    const float n = (float)HW * (float)cpg;
    const float pivot = bf2f(x[(long)b * HW * C + g * cpg]);
    const float md = s1 / n;
    const float var = fmaxf(s2 / n - md * md, 0.f);
    stats[2 * i] = pivot + md;
    stats[2 * i + 1] = rsqrtf(var + eps);
}

// Coalesced statistics pass: a thread owns one 16-byte chunk (8 channels) of a pixel row and walks the pixels of its slice, so a wave reads
// whole 128-byte lines (the per-group form above reads C / G channels out of every 2 C-byte pixel row: 8 of 256 bytes at C = 128); the
// per-channel partial sums meet in LDS and are folded into per-group sums in a fixed order.  Needs 256 % (C / 8) == 0.
__global__ __launch_bounds__(256) void gn_stats_rows_bf16_kernel(const bf16_t* __restrict__ x, float* __restrict__ partial, int HW, int C, int G, int S) {
    const int b = blockIdx.y, sl = blockIdx.x, tid = threadIdx.x;
    const int cpg = C / G, CH = C >> 3, ppi = 256 / CH;
    const int ck = tid % CH, pr = tid / CH;
    const bf16_t* xb = x + (long)b * HW * C;
    const int per = (HW + S - 1) / S;
    const int lo = sl * per, hi = min(HW, lo + per);
    float piv[8], s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        piv[e] = bf2f(xb[((ck * 8 + e) / cpg) * cpg]);
        s1[e] = s2[e] = 0.f;
    }
    for (int p = lo + pr; p < hi; p += ppi) {
        const u32x4_t v = *(const u32x4_t*)(xb + (long)p * C + ck * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = ((e & 1) ? hi2f(v[e >> 1]) : lo2f(v[e >> 1])) - piv[e];
            s1[e] += d; s2[e] += d * d;
        }
    }
    __shared__ float red[256][17];
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[tid][e] = s1[e]; red[tid][8 + e] = s2[e]; }
    __syncthreads();
    for (int g = tid; g < G; g += 256) {
        float a1 = 0.f, a2 = 0.f;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c)
            for (int r = 0; r < ppi; ++r) { a1 += red[r * CH + (c >> 3)][c & 7]; a2 += red[r * CH + (c >> 3)][8 + (c & 7)]; }
        float* o = partial + (((long)b * G + g) * S + sl) * 2;
        o[0] = a1; o[1] = a2;
    }
}

// per (image, channel): y = x * a + b with a = rstd * gamma, b = beta - mean * a  (the apply pass then needs no division and no gather)
__global__ void gn_affine_kernel(const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
                                 float* __restrict__ ab, int C, int G, int BC) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BC) return;
    const int b = i / C, c = i - b * C, g = c / (C / G);
    const float mean = stats[2 * (b * G + g)], inv = stats[2 * (b * G + g) + 1];
    const float a = inv * gamma[c];
    ab[2 * (long)i] = a;
    ab[2 * (long)i + 1] = beta[c] - mean * a;
}

__global__ __launch_bounds__(256) void gn_apply_bf16_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, const float* __restrict__ ab,
                                                            int HW, int C, int swish) {
    const int b = blockIdx.y;
    const long total8 = (long)HW * C / 8;
    const float* abb = ab + (long)b * C * 2;
    for (long i8 = (long)blockIdx.x * 256 + threadIdx.x; i8 < total8; i8 += (long)gridDim.x * 256) {
        const long i = i8 * 8;
        const int c = (int)(i % C);
        const long off = (long)b * HW * C + i;
        const u32x4_t v = *(const u32x4_t*)(x + off);
        const f32x4_t q0 = *(const f32x4_t*)(abb + 2 * c), q1 = *(const f32x4_t*)(abb + 2 * c + 4), q2 = *(const f32x4_t*)(abb + 2 * c + 8),
                      q3 = *(const f32x4_t*)(abb + 2 * c + 12);
        const float aa[8] = {q0[0], q0[2], q1[0], q1[2], q2[0], q2[2], q3[0], q3[2]}, bb[8] = {q0[1], q0[3], q1[1], q1[3], q2[1], q2[3], q3[1], q3[3]};
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xv = (e & 1) ? hi2f(v[e >> 1]) : lo2f(v[e >> 1]);
            float t = xv * aa[e] + bb[e];
            if (swish) t = t * __builtin_amdgcn_rcpf(1.0f + __expf(-t));
            o[e] = t;
        }
        *(u32x4_t*)(y + off) = (u32x4_t){pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7])};
    }
}

extern "C" int bagel_groupnorm_bf16(const void* x, void* y, float* partial_ws, const float* gamma, const float* beta, int32_t B, int32_t HW,
                                    int32_t C, int32_t groups, float eps, int32_t swish, hipStream_t stream) {
    BAGEL_REQUIRE(x && y && partial_ws && gamma && beta, "groupnorm_bf16: null pointer");
    BAGEL_REQUIRE(C % groups == 0 && C % 8 == 0, "groupnorm_bf16: C=%d must divide into %d groups and be a multiple of 8", C, groups);
    BAGEL_REQUIRE((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "groupnorm_bf16: 16-byte alignment");
    if (B <= 0 || HW <= 0) return BAGEL_OK;
    // workspace: [B*G*GN_BF16_SLICES*2] partials (S <= GN_BF16_SLICES of them used), [B*G*2] stats, [B*C*2] per-channel (scale, shift).
    // S: enough slices that B * S workgroups cover the chip several times over at the 1024^2 sizes (64 left 3/4 of the CUs idle), never
    // fewer than 64 pixels per slice.
    float* stats = partial_ws + (long)B * groups * GN_BF16_SLICES * 2;
    float* ab = stats + (long)B * groups * 2;
    const int CH = C / 8;
    const int S = (int)std::max(1L, std::min((long)GN_BF16_SLICES, (long)HW / 64));
    if (CH <= 256 && 256 % CH == 0)
        hipLaunchKernelGGL(gn_stats_rows_bf16_kernel, dim3(S, B), dim3(256), 0, stream, (const bf16_t*)x, partial_ws, HW, C, groups, S);
    else
        hipLaunchKernelGGL(gn_stats_bf16_kernel, dim3(S, groups, B), dim3(256), 0, stream, (const bf16_t*)x, partial_ws, HW, C, groups, S);
    hipLaunchKernelGGL(gn_finalize_bf16_kernel, dim3(B * groups), dim3(64), 0, stream, (const bf16_t*)x, partial_ws, stats, HW, C, groups, eps,
                       B * groups, S);
    hipLaunchKernelGGL(gn_affine_kernel, dim3(ceil_div(B * C, 256)), dim3(256), 0, stream, stats, gamma, beta, ab, C, groups, B * C);
    const long total8 = (long)HW * C / 8;
    const int blocks = (int)min((long)ceil_div(total8, 256), 4096L);
    hipLaunchKernelGGL(gn_apply_bf16_kernel, dim3(blocks, B), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, ab, HW, C, swish);
    return bagel_check_launch("groupnorm bf16 kernels");
}

// softmax over fp32 score rows -> bf16 probabilities: y = bf16(softmax(scale * x)).  One block per row.
__global__ __launch_bounds__(256) void softmax_rows_bf16_kernel(const float* __restrict__ x, long ldx, bf16_t* __restrict__ y, long ldy, int cols, float scale) {
    const float* r = x + (long)blockIdx.x * ldx;
    bf16_t* o = y + (long)blockIdx.x * ldy;
    __shared__ float red[256];
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < cols; c += 256) mx = fmaxf(mx, r[c]);
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]); __syncthreads(); }
    mx = red[0];
    __syncthreads();
    float sum = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) sum += __expf((r[c] - mx) * scale);
    red[threadIdx.x] = sum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    const float inv = 1.0f / red[0];
    for (int c = threadIdx.x; c < cols; c += 256) o[c] = f2bf(__expf((r[c] - mx) * scale) * inv);
}

extern "C" int bagel_softmax_rows_bf16(const float* x, int64_t ldx, void* y, int64_t ldy, int32_t rows, int32_t cols, float scale, hipStream_t stream) {
    BAGEL_REQUIRE(x && y && cols > 0, "softmax_rows_bf16: bad arguments");
    if (rows <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(softmax_rows_bf16_kernel, dim3(rows), dim3(256), 0, stream, x, (long)ldx, (bf16_t*)y, (long)ldy, cols, scale);
    return bagel_check_launch("softmax_rows_bf16_kernel");
}

// DiagonalGaussian sample + latent scale/shift on bf16 moments, every elementwise op rounding to bf16 as eager bf16 tensors do
// (autoencoder.py:280-287, 315-318 under autocast: the moments leave conv_out in bf16 and nothing casts them up).
__global__ void vae_reparam_bf16_kernel(const bf16_t* __restrict__ mom, long ld_mom, const bf16_t* __restrict__ noise, bf16_t* __restrict__ z, long n_pix,
                                        int zc, float scale, float shift) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pix * zc) return;
    const long pix = i / zc;
    const int c = (int)(i - pix * zc);
    const float mean = bf2f(mom[pix * ld_mom + c]), logvar = bf2f(mom[pix * ld_mom + zc + c]);
    const float stdv = bfround(expf(bfround(0.5f * logvar)));
    const float s = noise ? bfround(mean + bfround(stdv * bf2f(noise[i]))) : mean;
    z[i] = f2bf(scale * bfround(s - shift));          // python scalars enter a bf16 elementwise op as fp32 operands: not rounded first
}

extern "C" int bagel_vae_reparam_bf16(const void* moments, int64_t ld_moments, const void* noise, void* z, int64_t n_pix, int32_t z_channels,
                                      float scale, float shift, hipStream_t stream) {
    BAGEL_REQUIRE(moments && z, "vae_reparam_bf16: null pointer");
    const long n = n_pix * z_channels;
    if (n <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(vae_reparam_bf16_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, stream, (const bf16_t*)moments, (long)ld_moments, (const bf16_t*)noise,
                       (bf16_t*)z, (long)n_pix, z_channels, scale, shift);
    return bagel_check_launch("vae_reparam_bf16_kernel");
}
