// MFMA issue-rate microbenchmark for gfx950: cycles per MFMA as a function of the distance (in instructions) between
// two MFMAs that accumulate into the same register block.  hipcc --offload-arch=gfx950 -O3 mfma_dep.hip -o mfma_dep
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int NACC, int SHAPE>   // SHAPE 0: 32x32x16, 1: 16x16x32
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x % 3); b[i] = (short)(0x3f80 + threadIdx.x % 5); }
    f32x16_t acc32[NACC];
    f32x4_t acc16[NACC];
    for (int n = 0; n < NACC; ++n) { for (int r = 0; r < 16; ++r) acc32[n][r] = 0.f; for (int r = 0; r < 4; ++r) acc16[n][r] = 0.f; }
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 16 / NACC; ++rep)
#pragma unroll
            for (int n = 0; n < NACC; ++n) {
                if (SHAPE == 0) acc32[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc32[n], 0, 0, 0);
                else acc16[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc16[n], 0, 0, 0);
            }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) s += SHAPE == 0 ? acc32[n][0] : acc16[n][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC, int SHAPE>
void run(int threads, const char* name) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, SHAPE><<<256, threads>>>(out, cyc, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC, SHAPE><<<256, threads>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double nm = (double)iters * 16;
    const double flop = (SHAPE == 0 ? 32768.0 : 16384.0) * nm * (threads / 64) * 256;
    printf("%-10s waves/SIMD=%d  dep-distance=%2d : %.1f s_memtime-ticks/MFMA/wave, %.1f TFLOP/s\n", name, threads / 256, NACC, c / nm, flop / ms / 1e9);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int th : {256, 512}) {
        run<1, 0>(th, "32x32x16"); run<2, 0>(th, "32x32x16"); run<4, 0>(th, "32x32x16"); run<8, 0>(th, "32x32x16");
        run<1, 1>(th, "16x16x32"); run<2, 1>(th, "16x16x32"); run<4, 1>(th, "16x16x32"); run<8, 1>(th, "16x16x32");
    }
    return 0;
}
