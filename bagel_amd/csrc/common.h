// Shared device/host helpers for libbagel_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
// the public C ABI: every translation unit sees the prototypes of the entry points it defines, so a definition that drifts from include/bagel_hip.h is a
// COMPILE error ("conflicting types") instead of a ctypes call with the wrong arguments at run time
#include "../../include/bagel_hip.h"

typedef unsigned short bf16_t;   // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // MFMA A/B fragment (8 bf16 = 4 VGPR)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

#define BAGEL_OK 0
#define BAGEL_ERR_ARG (-1)
#define BAGEL_ERR_LAUNCH (-2)
#define BAGEL_ERR_UNSUPPORTED (-3)

int bagel_set_error(int code, const char* fmt, ...);
int bagel_check_launch(const char* what);
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device): a process that drives several GPUs must enable
// the > 64 KB LDS kernels on each of them; the return code is checked and reported through bagel_set_error.
int bagel_enable_lds(const void* func, int bytes, const char* what);

#define BAGEL_REQUIRE(cond, ...)                                  \
    do {                                                          \
        if (!(cond)) return bagel_set_error(BAGEL_ERR_ARG, __VA_ARGS__); \
    } while (0)

// ---- bf16 <-> f32 (round-to-nearest-even, identical to torch's CPU/GPU conversion) ----
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
// gfx950 converts in hardware (v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN) -- the same rounding torch uses.
typedef __attribute__((ext_vector_type(2))) __bf16 hwbf16x2_t;
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ float bfround(float f) { return bf2f(f2bf(f)); }
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    hwbf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float lo2f(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi2f(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// ---- wave (64 lanes) reductions ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- activations (fp32 math on bf16-rounded inputs, result rounded by the caller) ----
// x * rcp(1 + e^-x).  v_rcp_f32 is accurate to 1 ulp, so the fp32 value is within ~1.5 ulp of the correctly rounded quotient --
// invisible after the bf16 rounding that follows every use except on ~2^-14 of the inputs (1 bf16 ulp there).  The IEEE
// division sequence it replaces was 11 of the ~20 VALU instructions per SwiGLU output of the GEMM epilogue.
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float gelu_tanh_f(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    float inner = k0 * (x + k1 * x * x * x);
    return 0.5f * x * (1.0f + tanhf(inner));
}

// Streaming loads of data that ONE workgroup reads ONCE per launch (decode weights, KV pages): non-temporal cache policy
// (global_load_dwordx4 ... nt).  MI355X_MICROARCH.md, row nt-weights: issue -> landed -18 %, decode layer 5-10 % faster.
// BAGEL_NT_WEIGHTS=0 at build time restores plain loads (tools/ab_build.sh: same-box A/B).  Measured on one box (tools/ab_decode_nt.sh,
// profiles/r02_decode_nt_ab.log): batch-1 decode 280.2 tok/s plain -> 294.0 with nt on the gemv weight stream; nt on the KV pages is
// neutral (kept plain); nt on the skinny MFMA GEMM's weight fragments is SLOWER (batch 8: 1 169 vs 1 284 tok/s: kept plain).
#ifndef BAGEL_NT_WEIGHTS
#define BAGEL_NT_WEIGHTS 1
#endif
template <typename T>
__device__ __forceinline__ T ld_stream(const void* ptr) {
#if BAGEL_NT_WEIGHTS
    return __builtin_nontemporal_load((const T*)ptr);
#else
    return *(const T*)ptr;
#endif
}
#ifndef BAGEL_NT_KV
#define BAGEL_NT_KV 0
#endif
template <typename T>
__device__ __forceinline__ T ld_stream_kv(const void* ptr) {
#if BAGEL_NT_KV
    return __builtin_nontemporal_load((const T*)ptr);
#else
    return *(const T*)ptr;
#endif
}
#ifndef BAGEL_NT_SKINNY
#define BAGEL_NT_SKINNY 0
#endif

// 16 zero bytes a lane can DMA from when its chunk lies outside the operand (K tail, conv padding)
static __device__ __attribute__((aligned(16), used)) unsigned int bagel_zero16[4] = {0u, 0u, 0u, 0u};

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
