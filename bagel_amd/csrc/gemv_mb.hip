// Batched-decode projection: C[M <= 32, N] = norm(A)[M,K] W[N,K]^T (+bias)(act)(SwiGLU16)(+R) as a pure WEIGHT STREAM through the MFMA.
//
// Replaces, for 2..32 concurrent requests (and the 17..32 rows of a short text prefill), the reference's F.linear call sites of a decode step (qwen2_navit.py:515-517,591-594;
// modeling_qwen2.py:54-59,200-201; bagel.py:978) and the marker-row side path of a denoise forward (16 und rows per layer).
//
// Why a third small-M kernel: decode.hip's lane-FMA gemv is VALU-bound from two rows, and skinny.hip's MFMA kernel loads the
// ACTIVATION fragment of every 32-deep step from L2 next to the weight fragment -- 1 KB of x per 1 KB of W through the same vector
// memory pipe, which is why 16 requests streamed the weights at 3.3 TB/s where one request gets 6.4.  Here the activations never
// touch the memory pipe inside the loop:
//   * the 8 waves of a workgroup PARTITION K (K = 3584: 14 steps of 32 per wave); a wave loads its x fragments ONCE into registers
//     (mfma_f32_16x16x32_bf16 B operand: lane = (request r = lane % 16, k chunk q = lane / 16), 4 VGPRs per step), applies the fused
//     Qwen2RMSNorm there (row sum of squares: two lane exchanges + one pass through LDS across the waves), and keeps them for every
//     column block the workgroup processes;
//   * per column block (16 weight rows) a wave issues NS fragment loads of 1 KB (row lane % 16, 16 bytes at chunk lane / 16: row-major
//     W IS the A-operand layout) and refills each register quad in place right behind the MFMA that consumed it, so NS KB per wave
//     (112-152 KB per CU) are in flight for the whole life of the workgroup and the loop holds nothing but loads and MFMAs;
//   * the eight K partials of a block meet in LDS (1 KB per wave and block); wave g sums block g in a fixed order and runs the epilogue
//     with gemm.hip's rounding points (lane owns 4 consecutive output columns of one request: 8-byte stores);
//   * SwiGLU16 needs no special loop: with the gate/up rows interleaved in blocks of 16, block b is ALWAYS weight rows [16 b, 16 b + 16);
//     the reducing wave pairs blocks 2 j (gate) and 2 j + 1 (up);
//   * long rows (the down projection, K = 18944 = 592 steps) are cut over KS workgroups as well; those write fp32 slabs [KS][16][N] and a
//     small second kernel sums them in slice order and applies the epilogue -- deterministic (no atomics), one extra graph node per layer.
// Results are independent of the launch geometry only up to the fp32 summation order (8 K partials instead of one chain), like the
// K-split tiles of gemm.hip; tests pin it to the fp32 product within 2 bf16 ulp.
#include "common.h"
#include <stdlib.h>

#define EPI_NONE 0
#define EPI_GELU_TANH 1
#define EPI_SILU 2
#define EPI_SWIGLU16 3

struct MbParams {
    const bf16_t* A; long lda;
    const bf16_t* W; long ldw;
    const bf16_t* bias;
    const bf16_t* R; long ldr;
    bf16_t* C; long ldc;
    const bf16_t* norm_w; float eps;
    float* part;      // KS > 1: fp32 slabs [KS][16][N]
    int M, N, K, epi;

    int per;          // 32-deep k-steps per wave (<= NS)
    int KS;           // K slices over workgroups (gridDim.y)
    int slab_rows;    // rows of one fp32 slab: 16 x MB
};

__device__ __forceinline__ void mb_epilogue(const MbParams& p, f32x4_t a, int m, int n) {
    float o[4] = {a[0], a[1], a[2], a[3]};
    if (p.bias) {
        const u32x2_t bv = *(const u32x2_t*)(p.bias + n);
        o[0] += lo2f(bv[0]); o[1] += hi2f(bv[0]); o[2] += lo2f(bv[1]); o[3] += hi2f(bv[1]);
    }
    if (p.epi == EPI_GELU_TANH) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = gelu_tanh_f(bfround(o[e]));
    } else if (p.epi == EPI_SILU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = silu_f(bfround(o[e]));
    }
    if (p.R) {
        const u32x2_t rv = *(const u32x2_t*)(p.R + (long)m * p.ldr + n);
        o[0] = bfround(o[0]) + lo2f(rv[0]); o[1] = bfround(o[1]) + hi2f(rv[0]);
        o[2] = bfround(o[2]) + lo2f(rv[1]); o[3] = bfround(o[3]) + hi2f(rv[1]);
    }
    const u32x2_t v = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
    *(u32x2_t*)(p.C + (long)m * p.ldc + n) = v;
}

// Weight fragment loads are PLAIN: the non-temporal policy that helps the one-request gemv (common.h ld_stream) costs this kernel 11-13 %
// (round 4, same box, interleaved processes: gate+up 53.2 -> 61.5 us, 28-layer step 3.32 -> 3.68 ms) -- a row-major fragment is half a
// 128-byte line of 16 rows, and the other half (the next step's fragment, already in flight) wants to find the line in the cache.
// MB = request blocks of 16 (1: M <= 16, 2: 17..32 rows -- a short text prefill, a 32-request decode step): every weight fragment feeds MB MFMAs,
// the activation fragments of all MB x 16 rows stay in registers (MB x NS x 4 VGPRs), so the weight stream is read ONCE for up to 32 rows.
// LDS: the K partials [CH][8 waves][MB][64 lanes] f32x4 + the norm's row sums [MB][8][16] -- 48.5 KB static for one request block, 97 KB dynamic for two.
template <int NS, bool NORM, int MB>
__global__ __launch_bounds__(512) void gemv_mb_kernel(const MbParams p) {
    constexpr int CH = 6;                                      // blocks between two reductions (LDS: CH x 8 waves x MB KB)
    // one request block: 48.5 KB static; two: 97 KB, dynamic (bagel_enable_lds)
    extern __shared__ __attribute__((aligned(16))) char mb_smem[];
    __shared__ __attribute__((aligned(16))) f32x4_t part1[MB == 1 ? CH * 8 * 64 : 1];
    __shared__ float red1[MB == 1 ? 8 * 16 : 1];
    f32x4_t* part = MB == 1 ? part1 : (f32x4_t*)mb_smem;       // [CH * 8 * MB * 64]
    float* red = MB == 1 ? red1 : (float*)(mb_smem + CH * 8 * MB * 1024);       // [MB * 8 * 16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const bool swiglu = p.epi == EPI_SWIGLU16;
    const int nsteps = p.K >> 5;
    const int s0 = (blockIdx.y * 8 + wave) * p.per;
    int ns = nsteps - s0;
    ns = ns < 0 ? 0 : (ns > p.per ? p.per : ns);
    // fragment i of this wave = step sb + i, sb = min(s0, nsteps - NS) (the host guarantees nsteps >= NS): every address is ONE lane base +
    // a compile-time offset i * 64 bytes and stays inside the row; fragments outside [s0, s0 + ns) belong to a neighbour and get x = 0
    const int sb = s0 < nsteps - NS ? s0 : nsteps - NS;
    const int lo = s0 - sb, hi = lo + ns;                      // live fragments: lo <= i < hi
    // this workgroup's blocks: units (a block; SwiGLU16: a gate/up pair of blocks) [u0, u1) of the launch's even split over gridDim.x
    const int unit = swiglu ? 2 : 1;
    const int nunits = (p.N >> 4) / unit;
    const int bA = (int)((long)nunits * blockIdx.x / gridDim.x) * unit;
    const int bB = (int)((long)nunits * (blockIdx.x + 1) / gridDim.x) * unit;
    if (bA >= bB) return;                                      // (whole workgroup: no barrier is skipped by a part of it)

    // ---- (1) this wave's activation fragments, the norm weights at the same k positions, the first block's weight fragments ----
    bf16x8_t xf[MB][NS];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int m = 16 * mb + r < p.M ? 16 * mb + r : p.M - 1;      // requests >= M compute a duplicate that is never stored
        const bf16_t* xa = p.A + (long)m * p.lda + q * 8 + (long)sb * 32;
#pragma unroll
        for (int i = 0; i < NS; ++i) xf[mb][i] = *(const bf16x8_t*)(xa + i * 32);
    }
    // W row-major [N][ldw]: fragment (block, step) = 16 rows x 64 bytes.  A fragment-major copy of the weights ([N / 16][K / 32][16][32]: 1 KB
    // contiguous per fragment) was timed in round 4 and is NOT built: gate+up 53.3 -> 52.3 us at 16 requests, 52.0 -> 50.1 at 2 -- not worth a
    // second 14 GB image of the weights (profiles/r04_gemv_mb_bench.log).
    const bf16_t* wrow = p.W + (long)r * p.ldw + q * 8 + (long)sb * 32;       // + block * 16 rows
    const long blk_stride = 16 * p.ldw;
    bf16x8_t wf[NS];
    if constexpr (NORM) {
        // Qwen2RMSNorm of the request rows (modeling_qwen2.py:54-59: fp32 statistics, bf16(x * inv) * w rounded to bf16).  The weight
        // stream starts underneath it: loads return in order, so the activations and norm weights (L2) come back first.
        const bf16_t* ga = p.norm_w + q * 8 + (long)sb * 32;
        u32x4_t gw[MB == 1 ? NS : 1];
        if constexpr (MB == 1) {
#pragma unroll
            for (int i = 0; i < NS; ++i) gw[i] = *(const u32x4_t*)(ga + i * 32);
            const bf16_t* wb = wrow + (long)bA * blk_stride;
#pragma unroll
            for (int i = 0; i < NS; ++i) wf[i] = *(const bf16x8_t*)(wb + i * 32);
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const u32x4_t v = __builtin_bit_cast(u32x4_t, xf[mb][i]);
                float t = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = lo2f(v[e]), b = hi2f(v[e]);
                    t += a * a + b * b;
                }
                ss += (i >= lo && i < hi) ? t : 0.f;
            }
            ss += __shfl_xor(ss, 16, 64);
            ss += __shfl_xor(ss, 32, 64);
            if (lane < 16) red[(mb * 8 + wave) * 16 + lane] = ss;
        }
        __syncthreads();
        float invs[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) tot += red[(mb * 8 + w) * 16 + r];
            invs[mb] = rsqrtf(tot / (float)p.K + p.eps);
        }
        if constexpr (MB == 1) {
            float invo = invs[0];
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                u32x4_t v = __builtin_bit_cast(u32x4_t, xf[0][i]);
                if (i % 4 == 0) asm volatile("" : "+v"(invo));     // four fragments at a time: hipcc otherwise unpacks all NS at once and spills
                const float inv = invo;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v[e] = pack2bf(bfround(lo2f(v[e]) * inv) * lo2f(gw[i][e]), bfround(hi2f(v[e]) * inv) * hi2f(gw[i][e]));
                xf[0][i] = __builtin_bit_cast(bf16x8_t, v);
            }
        } else {
            // two request blocks: the register file does not hold all norm weights beside 2 x NS activation fragments and the first weight fragments
            // while those are being unpacked -- the norm weights come four fragments at a time (L2 hits); the weight stream starts HERE, behind the
            // row statistics, and flies under the scaling pass
            constexpr int NE = NS / 2;                         // fragments of the first block requested ahead of the scaling pass
            {
                const bf16_t* wb = wrow + (long)bA * blk_stride;
#pragma unroll
                for (int i = 0; i < NE; ++i) wf[i] = *(const bf16x8_t*)(wb + i * 32);
            }
#pragma unroll
            for (int i0 = 0; i0 < NS; i0 += 4) {
                u32x4_t g4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (i0 + j < NS) g4[j] = *(const u32x4_t*)(ga + (i0 + j) * 32);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    float inv = invs[mb];
                    asm volatile("" : "+v"(inv));
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (i0 + j >= NS) continue;
                        u32x4_t v = __builtin_bit_cast(u32x4_t, xf[mb][i0 + j]);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            v[e] = pack2bf(bfround(lo2f(v[e]) * inv) * lo2f(g4[j][e]), bfround(hi2f(v[e]) * inv) * hi2f(g4[j][e]));
                        xf[mb][i0 + j] = __builtin_bit_cast(bf16x8_t, v);
                    }
                }
            }
            {
                const bf16_t* wb = wrow + (long)bA * blk_stride;
#pragma unroll
                for (int i = NE; i < NS; ++i) wf[i] = *(const bf16x8_t*)(wb + i * 32);
            }
        }
    } else {
        const bf16_t* wb = wrow + (long)bA * blk_stride;
#pragma unroll
        for (int i = 0; i < NS; ++i) wf[i] = *(const bf16x8_t*)(wb + i * 32);
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int i = 0; i < NS; ++i)
            if (i < lo || i >= hi) xf[mb][i] = (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0};

    // ---- (2) the stream.  Per block NS x MB MFMAs; every fragment register is refilled in place, right behind the MFMAs that consumed it, with
    //      the same fragment of the NEXT block (across chunk seams too: the stream never stops for a reduction).  Every CH blocks the
    //      eight K partials meet in LDS: wave j sums output j (a block; SwiGLU16: gate block 2 j + up block 2 j + 1) and finishes it. ----
    for (int c0 = bA; c0 < bB; c0 += CH) {
        const int cb = bB - c0 < CH ? bB - c0 : CH;             // blocks of this chunk
        for (int g = 0; g < cb; ++g) {
            const int bn = c0 + g + 1;                         // the block to prefetch (the last one re-reads itself: L2 hits, never used)
            const bf16_t* wn = wrow + (long)(bn < bB ? bn : bB - 1) * blk_stride;
            f32x4_t acc[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[mb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            if (bn < bB) {
#pragma unroll
                for (int i = 0; i < NS; ++i) {
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[mb][i], acc[mb], 0, 0, 0);
                    wf[i] = *(const bf16x8_t*)(wn + i * 32);
                }
            } else {
#pragma unroll
                for (int i = 0; i < NS; ++i)
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[mb][i], acc[mb], 0, 0, 0);
            }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) part[((g * 8 + wave) * MB + mb) * 64 + lane] = acc[mb];
        }
        __syncthreads();
        const int nout = swiglu ? cb >> 1 : cb;
        // (output j, request block mb) pairs dealt over the 8 waves: with two request blocks a chunk of up to four outputs still finishes in ONE pass (the
        // residual / store latency of an epilogue is paid once, not once per request block)
        for (int t = wave; t < nout * MB; t += 8) {
            const int j = t % nout, mb = t / nout;
            {
                const int m = 16 * mb + r;
                if (swiglu) {
                    f32x4_t ag = part[(((2 * j) * 8) * MB + mb) * 64 + lane], au = part[(((2 * j + 1) * 8) * MB + mb) * 64 + lane];
#pragma unroll
                    for (int w = 1; w < 8; ++w) {
                        ag = ag + part[(((2 * j) * 8 + w) * MB + mb) * 64 + lane];
                        au = au + part[(((2 * j + 1) * 8 + w) * MB + mb) * 64 + lane];
                    }
                    if (m < p.M) {
                        float o[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = bfround(silu_f(bfround(ag[e]))) * bfround(au[e]);
                        const u32x2_t v = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
                        *(u32x2_t*)(p.C + (long)m * p.ldc + ((c0 >> 1) + j) * 16 + q * 4) = v;
                    }
                } else {
                    f32x4_t a = part[((j * 8) * MB + mb) * 64 + lane];
#pragma unroll
                    for (int w = 1; w < 8; ++w) a = a + part[((j * 8 + w) * MB + mb) * 64 + lane];
                    const int n = (c0 + j) * 16 + q * 4;
                    if (p.KS > 1) *(f32x4_t*)(p.part + ((long)blockIdx.y * (16 * MB) + m) * p.N + n) = a;      // all 16 x MB rows: the slab has them
                    else if (m < p.M) mb_epilogue(p, a, m, n);
                }
            }
        }
        if (c0 + CH < bB) __syncthreads();                     // the partials are consumed before the next chunk overwrites them
    }
}

// KS > 1: C = epilogue(sum over the K slices, in slice order)
__global__ __launch_bounds__(256) void gemv_mb_reduce_kernel(const MbParams p) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int nq = p.N >> 2;
    if (idx >= p.M * nq) return;
    const int m = idx / nq, n = (idx - m * nq) * 4;
    f32x4_t a = *(const f32x4_t*)(p.part + (long)m * p.N + n);
    for (int s = 1; s < p.KS; ++s) a = a + *(const f32x4_t*)(p.part + ((long)s * p.slab_rows + m) * p.N + n);
    mb_epilogue(p, a, m, n);
}

// Geometry.  Instantiations: NS = 4 (short rows, K <= 1024: the tiny test models), 10, 14 (K = 3584 in one slice: 14 steps per wave at the 7B
// hidden size), 19.  The launch is PERSISTENT: one workgroup per CU (its ~170 registers x 8 waves leave no room for a second one), each
// taking an even share of the column blocks of its K slice, so the activation prologue is paid once per workgroup and launch.
// Rows too long for one slice (nsteps > 8 x 19) are cut over KS workgroups; KS and the workgroups per slice are chosen together so that
// the blocks divide evenly and the whole chip is used: the 7B down projection (592 steps, 224 blocks, 256 CUs) runs as 8 slices x 32
// workgroups x 7 blocks x 10 steps per wave -- 4 slices x 64 workgroups would leave 3 or 4 blocks per workgroup (makespan 4 of 3.5).
// BAGEL_MB_WGS overrides the workgroup count (tuning / tests).
static const int MB_NS[] = {4, 10, 14, 19};

// Two request blocks (17..32 rows) keep 2 x NS activation fragments per wave: NS = 19 does not fit the register file beside them, so the longest
// one-slice row is 8 x 14 steps (K = 3584, the 7B hidden size) and longer rows take more, shorter slices.
static int mb_max_ns(int MB) { return MB > 1 ? 14 : 19; }

static int mb_ns_for(int per, int MB = 1) {
    for (int ns : MB_NS)
        if (per <= ns && ns <= mb_max_ns(MB)) return ns;
    return 0;
}

static void mb_geometry(int N, int K, int wgs, int MB, int* per, int* KS, int* gx_out) {
    const int nsteps = K / 32, nblk = N / 16;
    const int NSM = mb_max_ns(MB);
    if (nsteps <= 8 * NSM) {                           // one slice: the fused norm / SwiGLU need the whole row in one workgroup
        *KS = 1;
        *per = (nsteps + 7) / 8;
        *gx_out = wgs < nblk ? wgs : nblk;
        return;
    }
    const int ks_min = (nsteps + 8 * NSM - 1) / (8 * NSM);
    int best_ks = ks_min, best_gx = 1;
    double best_cost = 1e30;
    for (int ks = ks_min; ks <= 4 * ks_min && ks <= wgs; ++ks) {
        const int pw = (nsteps + 8 * ks - 1) / (8 * ks);
        const int ns = mb_ns_for(pw, MB);
        if (ns == 0 || nsteps < ns) continue;
        int gx = wgs / ks;
        if (gx < 1) gx = 1;
        if (gx > nblk) gx = nblk;
        const int blocks = (nblk + gx - 1) / gx;       // of the busiest workgroup
        // time ~ the busiest workgroup's stream (blocks x fragments it LOADS, live or not) + a fixed cost per slice of slab traffic
        const double cost = (double)blocks * ns * (1.0 + 0.02 * ks);
        if (cost < best_cost) { best_cost = cost; best_ks = ks; best_gx = gx; }
    }
    *KS = best_ks;
    *per = (nsteps + 8 * best_ks - 1) / (8 * best_ks);
    *gx_out = best_gx;
}

static int mb_wgs() {
    static int wgs_env = -1, cus_of_dev[16] = {0};
    if (wgs_env < 0) {
        const char* e = getenv("BAGEL_MB_WGS");
        wgs_env = (e && atoi(e) > 0) ? atoi(e) : 0;
    }
    if (wgs_env) return wgs_env;
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 15;
    if (cus_of_dev[dev] == 0) {
        int cus = 0;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        cus_of_dev[dev] = cus > 0 ? cus : 256;
    }
    return cus_of_dev[dev];
}

extern "C" int bagel_gemv_mb_workspace_bytes(int32_t N, int32_t K, int64_t* bytes) {
    BAGEL_REQUIRE(bytes && N > 0 && K > 0 && (K % 32) == 0 && (N % 16) == 0, "gemv_mb_workspace_bytes: bad argument");
    // the slice count depends on the device's CU count and on the row count (17..32 rows: slices of at most 8 x 14 steps, slabs of 32 rows): an upper
    // bound that holds for every geometry the launcher may pick (4 x the minimum slice count of the 32-row form, 32-row slabs)
    const int nsteps = K / 32;
    const int ks_min = (nsteps + 8 * 14 - 1) / (8 * 14);
    *bytes = nsteps <= 8 * 14 ? 0 : (int64_t)4 * ks_min * 32 * N * 4;
    return BAGEL_OK;
}

template <int NS, bool NORM, int MB>
static int mb_launch3(const MbParams& p, dim3 grid, hipStream_t stream) {
    constexpr int smem = MB == 1 ? 0 : 6 * 8 * MB * 1024 + MB * 8 * 16 * 4;       // two request blocks: K partials of a chunk + the norm's row sums (dynamic: > 64 KB)
    if constexpr (MB > 1) {
        if (int rc = bagel_enable_lds((const void*)gemv_mb_kernel<NS, NORM, MB>, smem, "gemv_mb_kernel")) return rc;
    }
    hipLaunchKernelGGL((gemv_mb_kernel<NS, NORM, MB>), grid, dim3(512), smem, stream, p);
    return bagel_check_launch("gemv_mb_kernel");
}
template <int NS>
static int mb_launch(const MbParams& p, int MB, dim3 grid, hipStream_t stream) {
    if constexpr (NS <= 14) {
        if (MB > 1) return p.norm_w ? mb_launch3<NS, true, 2>(p, grid, stream) : mb_launch3<NS, false, 2>(p, grid, stream);
    }
    return p.norm_w ? mb_launch3<NS, true, 1>(p, grid, stream) : mb_launch3<NS, false, 1>(p, grid, stream);
}

extern "C" int bagel_gemv_mb_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias, const void* R, int64_t ldr,
                                  void* C, int64_t ldc, const void* norm_w, float eps, int32_t M, int32_t N, int32_t K, int32_t epilogue,
                                  void* workspace, int64_t workspace_bytes, hipStream_t stream) {
    BAGEL_REQUIRE(A && W && C, "gemv_mb: null pointer");
    BAGEL_REQUIRE(M >= 1 && M <= 32, "gemv_mb: M=%d not in [1,32]", M);
    BAGEL_REQUIRE(K > 0 && (K % 32) == 0 && (lda % 8) == 0 && (ldw % 8) == 0, "gemv_mb: K %% 32 == 0 and 16-byte rows required");
    BAGEL_REQUIRE(epilogue >= 0 && epilogue <= 3, "gemv_mb: unknown epilogue %d", epilogue);
    BAGEL_REQUIRE(epilogue == EPI_SWIGLU16 ? ((N % 32) == 0 && !bias && !R) : (N % 16) == 0, "gemv_mb: N %% 16 (SwiGLU: N %% 32, no bias/residual)");
    BAGEL_REQUIRE((ldc % 4) == 0 && (ldr % 4) == 0, "gemv_mb: ldc/ldr must be multiples of 4");
    BAGEL_REQUIRE((((uintptr_t)A | (uintptr_t)W | (uintptr_t)norm_w) & 15) == 0 && (((uintptr_t)C | (uintptr_t)R | (uintptr_t)bias) & 7) == 0, "gemv_mb: alignment");
    MbParams p;
    p.A = (const bf16_t*)A; p.lda = lda; p.W = (const bf16_t*)W; p.ldw = ldw; p.bias = (const bf16_t*)bias;
    p.R = (const bf16_t*)R; p.ldr = ldr; p.C = (bf16_t*)C; p.ldc = ldc; p.norm_w = (const bf16_t*)norm_w; p.eps = eps;
    p.M = M; p.N = N; p.K = K; p.epi = epilogue;
    const bool sw = epilogue == EPI_SWIGLU16;
    const int MB = M > 16 ? 2 : 1;                                  // request blocks of 16 rows that share every weight fragment
    p.slab_rows = 16 * MB;
    int gx = 1;
    mb_geometry(N, K, mb_wgs(), MB, &p.per, &p.KS, &gx);
    const int NS = mb_ns_for(p.per, MB);                            // the instantiation that runs; its fragments must fit the row
    if (NS == 0 || K / 32 < NS) return bagel_set_error(BAGEL_ERR_UNSUPPORTED, "gemv_mb: K=%d (%d steps per wave) has no instantiation", K, p.per);
    if (p.KS > 1) {
        if (norm_w || sw) return bagel_set_error(BAGEL_ERR_UNSUPPORTED, "gemv_mb: fused RMSNorm / SwiGLU need the whole row in one slice (K=%d)", K);
        BAGEL_REQUIRE(workspace && ((uintptr_t)workspace & 15) == 0 && workspace_bytes >= (int64_t)p.KS * p.slab_rows * N * 4,
                      "gemv_mb: K=%d runs as %d slices and needs a 16-byte aligned fp32 workspace of %lld bytes (bagel_gemv_mb_workspace_bytes)", K, p.KS,
                      (long long)p.KS * p.slab_rows * N * 4);
    }
    p.part = (float*)workspace;
    if (sw) {                                                       // units are gate/up PAIRS of blocks
        const int nunits = N / 32;
        if (gx > nunits) gx = nunits;
    }
    const dim3 grid(gx, p.KS);
    if (int rc = NS == 4 ? mb_launch<4>(p, MB, grid, stream) : NS == 10 ? mb_launch<10>(p, MB, grid, stream) : NS == 14 ? mb_launch<14>(p, MB, grid, stream)
                                                                                                             : mb_launch<19>(p, MB, grid, stream)) return rc;
    if (p.KS > 1) {
        hipLaunchKernelGGL(gemv_mb_reduce_kernel, dim3(ceil_div((long)M * (N / 4), 256)), dim3(256), 0, stream, p);
        return bagel_check_launch("gemv_mb_reduce_kernel");
    }
    return BAGEL_OK;
}
