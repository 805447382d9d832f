// Image pre/post-processing on the device: the byte/integer work either side of the forward path.
//
//   bagel_resample_u8      one separable pass of Pillow's 8-bit ImagingResample (src/libImaging/Resample.c), which is what
//                          torchvision's resize of a PIL image executes for data/transforms.py:88 -- fixed-point taps
//                          (22 fractional bits, computed by the host exactly as precompute_coeffs/normalize_coeffs_8bpc do),
//                          +2^21 rounding, arithmetic shift, clamp to [0,255].  Bit-exact by construction: integers only.
//   bagel_u8_to_chw_f32    ToTensor + Normalize (data/transforms.py:109-115): ((u8 / 255) - mean) / std, fp32, HWC -> CHW
//   bagel_chw_f32_to_u8    decode_image (inferencer.py:182-183): ((x * 0.5 + 0.5).clamp(0,1) * 255) truncated to uint8, CHW -> HWC
//
// All three are HBM-bound passes over a few MB; one thread per output element keeps the access rows coalesced.
#include "common.h"

#define RS_BITS 22

__device__ __forceinline__ unsigned char rs_clip8(int v) {
    v >>= RS_BITS;
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal: out[line][o][c] = sum_t in[line][first[o] + t][c] * kk[o][t]
__global__ __launch_bounds__(256) void resample_h_kernel(const unsigned char* __restrict__ in, long in_stride,
                                                         unsigned char* __restrict__ out, long out_stride, int out_len, int ch,
                                                         const int* __restrict__ bounds, const int* __restrict__ kk, int ksize) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= out_len) return;
    const long line = blockIdx.y;
    const int first = bounds[2 * o], n = bounds[2 * o + 1];
    const int* k = kk + (long)o * ksize;
    const unsigned char* src = in + line * in_stride + (long)first * ch;
    for (int c = 0; c < ch; ++c) {
        int ss = 1 << (RS_BITS - 1);
        for (int t = 0; t < n; ++t) ss += (int)src[(long)t * ch + c] * k[t];
        out[line * out_stride + (long)o * ch + c] = rs_clip8(ss);
    }
}

// vertical: out[o][b] = sum_t in[first[o] + t][b] * kk[o][t]   for every byte column b of a row
__global__ __launch_bounds__(256) void resample_v_kernel(const unsigned char* __restrict__ in, long in_stride,
                                                         unsigned char* __restrict__ out, long out_stride, int row_bytes,
                                                         const int* __restrict__ bounds, const int* __restrict__ kk, int ksize) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= row_bytes) return;
    const int o = blockIdx.y;
    const int first = bounds[2 * o], n = bounds[2 * o + 1];
    const int* k = kk + (long)o * ksize;
    int ss = 1 << (RS_BITS - 1);
    for (int t = 0; t < n; ++t) ss += (int)in[(long)(first + t) * in_stride + b] * k[t];
    out[(long)o * out_stride + b] = rs_clip8(ss);
}

extern "C" int bagel_resample_u8(const void* in, int64_t in_stride, void* out, int64_t out_stride, int32_t n_lines,
                                 int32_t out_len, int32_t channels, const int32_t* bounds, const int32_t* kk, int32_t ksize,
                                 int32_t vertical, hipStream_t stream) {
    BAGEL_REQUIRE(in && out && bounds && kk, "resample_u8: null pointer");
    BAGEL_REQUIRE(channels >= 1 && channels <= 4 && ksize >= 1, "resample_u8: channels %d / ksize %d", channels, ksize);
    if (n_lines <= 0 || out_len <= 0) return BAGEL_OK;
    if (vertical) {
        // n_lines = bytes per row (width * channels), out_len = output rows
        BAGEL_REQUIRE(out_len <= 65535, "resample_u8: more than 65535 output rows");
        hipLaunchKernelGGL(resample_v_kernel, dim3(ceil_div(n_lines, 256), out_len), dim3(256), 0, stream, (const unsigned char*)in,
                           (long)in_stride, (unsigned char*)out, (long)out_stride, n_lines, bounds, kk, ksize);
        return bagel_check_launch("resample_v_kernel");
    }
    BAGEL_REQUIRE(n_lines <= 65535, "resample_u8: more than 65535 rows");
    hipLaunchKernelGGL(resample_h_kernel, dim3(ceil_div(out_len, 256), n_lines), dim3(256), 0, stream, (const unsigned char*)in,
                       (long)in_stride, (unsigned char*)out, (long)out_stride, out_len, channels, bounds, kk, ksize);
    return bagel_check_launch("resample_h_kernel");
}

struct MeanStd { float mean[4]; float stdv[4]; };

__global__ __launch_bounds__(256) void u8_to_chw_f32_kernel(const unsigned char* __restrict__ in, long in_stride, float* __restrict__ out,
                                                            int H, int W, int C, MeanStd ms) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= W) return;
    const unsigned char* p = in + (long)y * in_stride + (long)x * C;
    for (int c = 0; c < C; ++c) {
        const float v = __fdiv_rn((float)p[c], 255.0f);                                  // ToTensor: .div(255)
        out[((long)c * H + y) * W + x] = __fdiv_rn(__fsub_rn(v, ms.mean[c]), ms.stdv[c]);   // Normalize: sub_ then div_
    }
}

extern "C" int bagel_u8_to_chw_f32(const void* in, int64_t in_stride, float* out, int32_t H, int32_t W, int32_t C,
                                   const float* mean, const float* stdv, hipStream_t stream) {
    BAGEL_REQUIRE(in && out && mean && stdv, "u8_to_chw_f32: null pointer (mean/std are HOST arrays of C floats)");
    BAGEL_REQUIRE(C >= 1 && C <= 4 && H <= 65535, "u8_to_chw_f32: C=%d H=%d", C, H);
    if (H <= 0 || W <= 0) return BAGEL_OK;
    MeanStd ms;
    for (int c = 0; c < 4; ++c) { ms.mean[c] = c < C ? mean[c] : 0.f; ms.stdv[c] = c < C ? stdv[c] : 1.f; }
    hipLaunchKernelGGL(u8_to_chw_f32_kernel, dim3(ceil_div(W, 256), H), dim3(256), 0, stream, (const unsigned char*)in, (long)in_stride,
                       out, H, W, C, ms);
    return bagel_check_launch("u8_to_chw_f32_kernel");
}

__global__ __launch_bounds__(256) void chw_f32_to_u8_kernel(const float* __restrict__ in, long chan_stride, long row_stride,
                                                            unsigned char* __restrict__ out, long out_stride, int W, int C) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= W) return;
    for (int c = 0; c < C; ++c) {
        float v = __fadd_rn(__fmul_rn(in[(long)c * chan_stride + (long)y * row_stride + x], 0.5f), 0.5f);
        v = fminf(fmaxf(v, 0.0f), 1.0f);
        out[(long)y * out_stride + (long)x * C + c] = (unsigned char)(int)__fmul_rn(v, 255.0f);   // truncation, as .to(uint8)
    }
}

extern "C" int bagel_chw_f32_to_u8(const float* in, int64_t chan_stride, int64_t row_stride, void* out, int64_t out_stride,
                                   int32_t H, int32_t W, int32_t C, hipStream_t stream) {
    BAGEL_REQUIRE(in && out, "chw_f32_to_u8: null pointer");
    BAGEL_REQUIRE(C >= 1 && C <= 4 && H <= 65535, "chw_f32_to_u8: C=%d H=%d", C, H);
    if (H <= 0 || W <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(chw_f32_to_u8_kernel, dim3(ceil_div(W, 256), H), dim3(256), 0, stream, in, (long)chan_stride, (long)row_stride,
                       (unsigned char*)out, (long)out_stride, W, C);
    return bagel_check_launch("chw_f32_to_u8_kernel");
}

// decode_image under the inferencer's bf16 autocast (inferencer.py:233 -> :182-183): the decoder output is bf16 and every elementwise op of
// ``(image * 0.5 + 0.5).clamp(0, 1) * 255`` rounds to bf16 before the truncating uint8 cast.  The source may have any element strides (the
// VAE engine hands over its NHWC buffer as a CHW view: no layout copy).
__global__ __launch_bounds__(256) void chw_bf16_to_u8_kernel(const bf16_t* __restrict__ in, long chan_stride, long row_stride, long col_stride,
                                                             unsigned char* __restrict__ out, long out_stride, int W, int C) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= W) return;
    for (int c = 0; c < C; ++c) {
        float v = bfround(bfround(bf2f(in[(long)c * chan_stride + (long)y * row_stride + (long)x * col_stride]) * 0.5f) + 0.5f);
        v = fminf(fmaxf(v, 0.0f), 1.0f);
        out[(long)y * out_stride + (long)x * C + c] = (unsigned char)(int)bfround(v * 255.0f);   // truncation, as .to(uint8)
    }
}

extern "C" int bagel_chw_bf16_to_u8(const void* in, int64_t chan_stride, int64_t row_stride, int64_t col_stride, void* out, int64_t out_stride,
                                    int32_t H, int32_t W, int32_t C, hipStream_t stream) {
    BAGEL_REQUIRE(in && out, "chw_bf16_to_u8: null pointer");
    BAGEL_REQUIRE(C >= 1 && C <= 4 && H <= 65535, "chw_bf16_to_u8: C=%d H=%d", C, H);
    if (H <= 0 || W <= 0) return BAGEL_OK;
    hipLaunchKernelGGL(chw_bf16_to_u8_kernel, dim3(ceil_div(W, 256), H), dim3(256), 0, stream, (const bf16_t*)in, (long)chan_stride, (long)row_stride,
                       (long)col_stride, (unsigned char*)out, (long)out_stride, W, C);
    return bagel_check_launch("chw_bf16_to_u8_kernel");
}
