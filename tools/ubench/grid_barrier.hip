// Grid-wide barrier microbenchmark for MI355X (256 CUs, 8 XCDs): what does one all-to-all synchronisation cost inside a persistent
// launch, against the ~1.1 us graph edge + ~3.6 us ramp of a launch boundary?  Decides whether a one-launch-per-token decode pays.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/grid_barrier.hip -o /tmp/grid_barrier && /tmp/grid_barrier
// Forms:  0 = one monotonic device-scope counter, every workgroup polls it
//         1 = sharded: 8 group counters (group = blockIdx & 7, the XCD under round-robin dispatch; correctness does not depend on it),
//             last arriver of a group -> top counter, last of those -> 8 per-group generation words, workgroups poll their group's word
// Every form: all waves drain their stores, one lane does release fence -> arrive -> relaxed poll with s_sleep -> ONE acquire fence.
// The check phase proves visibility: before each barrier every workgroup publishes a value, after it reads the values of 16 others.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define LINE 32   // unsigned per 128-byte line
struct Bar { unsigned cnt[8][LINE]; unsigned top[LINE]; unsigned gen[8][LINE]; unsigned one[LINE]; unsigned err[LINE]; };

#define RLX __ATOMIC_RELAXED
#define AG __HIP_MEMORY_SCOPE_AGENT

template <int FORM>
__device__ __forceinline__ void grid_barrier(Bar* b, unsigned epoch, unsigned nwg) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned spins = 0;
        if (FORM == 0) {
            __hip_atomic_fetch_add(&b->one[0], 1u, RLX, AG);
            while (__hip_atomic_load(&b->one[0], RLX, AG) < epoch * nwg) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 17)) { b->err[0] = epoch; break; }
            }
        } else {
            const unsigned grp = blockIdx.x & 7u, per = nwg >> 3;
            const unsigned old = __hip_atomic_fetch_add(&b->cnt[grp][0], 1u, RLX, AG);
            if (old == epoch * per - 1u) {
                const unsigned o2 = __hip_atomic_fetch_add(&b->top[0], 1u, RLX, AG);
                if (o2 == epoch * 8u - 1u)
                    for (int g = 0; g < 8; ++g) __hip_atomic_store(&b->gen[g][0], epoch, RLX, AG);
            }
            while (__hip_atomic_load(&b->gen[grp][0], RLX, AG) < epoch) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 17)) { b->err[0] = epoch; break; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <int FORM, bool CHECK>
__global__ __launch_bounds__(256) void k(Bar* b, unsigned* data, unsigned* bad, int iters) {
    const unsigned nwg = gridDim.x;
    unsigned wrong = 0;
    for (int it = 1; it <= iters; ++it) {
        if (CHECK) {
            // publish 256 words per workgroup (plain stores), then read 16 other workgroups' words after the barrier
            data[(size_t)blockIdx.x * 256 + threadIdx.x] = (unsigned)it * 1000003u + blockIdx.x * 256u + threadIdx.x;
        }
        grid_barrier<FORM>(b, CHECK ? 2u * it - 1u : (unsigned)it, nwg);
        if (CHECK) {
            const unsigned src = (blockIdx.x * 17u + (threadIdx.x >> 4) * 31u + 5u) % nwg, w = (threadIdx.x * 7u) & 255u;
            const unsigned v = data[(size_t)src * 256 + w];
            wrong += v != (unsigned)it * 1000003u + src * 256u + w;
            grid_barrier<FORM>(b, 2u * it, nwg);      // nobody overwrites before everyone has read
        }
    }
    if (CHECK && wrong) atomicAdd(bad, wrong);
}

template <int FORM, bool CHECK>
static void run(const char* name, int nwg, int iters) {
    Bar* b; unsigned *data, *bad;
    hipMalloc(&b, sizeof(Bar)); hipMalloc(&data, (size_t)nwg * 256 * 4); hipMalloc(&bad, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    unsigned hbad = 0, herr = 0;
    for (int rep = 0; rep < 5; ++rep) {
        hipMemset(b, 0, sizeof(Bar)); hipMemset(bad, 0, 4); hipMemset(data, 0, (size_t)nwg * 256 * 4);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<FORM, CHECK>), dim3(nwg), dim3(256), 0, 0, b, data, bad, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        unsigned t; hipMemcpy(&t, bad, 4, hipMemcpyDeviceToHost); hbad += t;
        hipMemcpy(&t, &b->err[0], 4, hipMemcpyDeviceToHost); herr |= t;
    }
    const int nbar = CHECK ? 2 * iters : iters;
    printf("%-28s %4d workgroups: %6.2f us per barrier%s   stale reads %u, timeouts %u\n", name, nwg, best * 1e3f / nbar,
           CHECK ? " (incl. publish + read-back)" : "", hbad, herr);
    fflush(stdout);
    hipFree(b); hipFree(data); hipFree(bad);
}

int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    printf("%s, %d CUs\n", pr.gcnArchName, cus);
    const int iters = 100;
    run<0, false>("one counter", cus, iters);
    run<1, false>("sharded 8 + top", cus, iters);
    run<0, true>("one counter, data check", cus, iters);
    run<1, true>("sharded 8 + top, data check", cus, iters);
    run<1, false>("sharded 8 + top", cus / 2, iters);
    return 0;
}
