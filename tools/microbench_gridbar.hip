// Cost of a grid barrier among N resident workgroups of 256 threads (one monotonic counter, bounded spin), in two
// flavours: (a) release fence before the arrival + acquire fence after the wait (plain loads/stores may carry the
// data), (b) no fences: the data must travel through sc1 (agent-scope relaxed atomic) stores and loads.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench_gridbar.hip -o tools/bin/mb_gridbar && tools/bin/mb_gridbar
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <bool FENCE>
__device__ __forceinline__ bool grid_barrier(unsigned *bar, unsigned nwg, unsigned &gen)
{
    __shared__ int ok_s;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        ++gen;
        if (FENCE) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = gen * nwg;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        int ok = 1;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (__builtin_amdgcn_s_memrealtime() - t0 > 20000000ull) { ok = 0; break; }  // 0.2 s at 100 MHz
        }
        if (FENCE) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        ok_s = ok;
    }
    __syncthreads();
    return ok_s != 0;
}

// every workgroup publishes a value per round and reads every other workgroup's value of that round: checks visibility too
template <bool FENCE>
__global__ __launch_bounds__(256) void k(unsigned *bar, double *slots, int rounds, unsigned long long *out, int *bad)
{
    unsigned gen = 0;
    const unsigned nwg = gridDim.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    int wrong = 0;
    for (int r = 0; r < rounds; ++r) {
        if (threadIdx.x == 0) {
            const double v = r * 1000.0 + blockIdx.x;
            if (FENCE) slots[(r & 1) * 1024 + blockIdx.x] = v;
            else __hip_atomic_store(&slots[(r & 1) * 1024 + blockIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!grid_barrier<FENCE>(bar, nwg, gen)) { if (threadIdx.x == 0) atomicAdd(bad, 1000000); return; }
        for (unsigned w = threadIdx.x; w < nwg; w += 256) {
            const double v = FENCE ? slots[(r & 1) * 1024 + w] : __hip_atomic_load(&slots[(r & 1) * 1024 + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v != r * 1000.0 + w) ++wrong;
        }
    }
    if (wrong) atomicAdd(bad, wrong);
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_memrealtime() - t0;
}

template <bool FENCE>
int run(const char *name, int nwg, int rounds)
{
    unsigned *bar; double *slots; unsigned long long *out; int *bad;
    CHECK(hipMalloc(&bar, 4)); CHECK(hipMalloc(&slots, 2048 * 8)); CHECK(hipMalloc(&out, 1024 * 8)); CHECK(hipMalloc(&bad, 4));
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(bar, 0, 4)); CHECK(hipMemset(bad, 0, 4)); CHECK(hipMemset(slots, 0, 2048 * 8));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        k<FENCE><<<nwg, 256>>>(bar, slots, rounds, out, bad);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        int hb = 0; CHECK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        if (rep == 2) printf("%-28s %4d workgroups: %.2f us per barrier round (incl. publish + read), stale or timed out: %d\n", name, nwg, ms * 1e3 / rounds, hb);
    }
    (void)hipFree(bar); (void)hipFree(slots); (void)hipFree(out); (void)hipFree(bad);
    return 0;
}

int main()
{
    for (int nwg : {20, 80, 128, 256}) {
        if (run<true>("fences, plain data", nwg, 2000)) return 1;
        if (run<false>("no fences, sc1 data", nwg, 2000)) return 1;
    }
    return 0;
}
