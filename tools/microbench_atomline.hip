// Is a line that the previous kernel updated with atomics slower to load than a line it wrote with plain stores?
// Kernel A: every wave adds to counters c[0..3] (atomics) and workgroup 0 stores p[0..3]; kernel B (dependent): thread 0
// of workgroup 0 times single loads with s_memrealtime (10 ns units), order: plain, atomic, atomic, plain.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/bin/microbench_atomline tools/microbench_atomline.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(256) void produce(int *c, int *p, int it)
{
    if ((threadIdx.x & 63) == 0) for (int k = 0; k < 4; ++k) atomicAdd(&c[k * 64], 1);
    if (blockIdx.x == 0 && threadIdx.x == 0) for (int k = 0; k < 4; ++k) p[k * 64] = it;
}
__device__ __forceinline__ int timed_load(const int *q, unsigned long long &dt)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    int v;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(q) : "memory");
    dt = __builtin_amdgcn_s_memrealtime() - t0;
    return v;
}
__global__ __launch_bounds__(256) void consume(const int *c, const int *p, unsigned long long *out, int *sink)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned long long d[6];
        int s = timed_load(p, d[0]);          // plain, first touch of the kernel
        s += timed_load(c, d[1]);             // atomic line
        s += timed_load(c + 64, d[2]);        // another atomic line
        s += timed_load(p + 64, d[3]);        // another plain line
        s += timed_load(c + 128, d[4]);
        s += timed_load(p + 128, d[5]);
        for (int k = 0; k < 6; ++k) out[k] = d[k];
        *sink = s;
    }
}
int main()
{
    int *c, *p, *sink; unsigned long long *out;
    CK(hipMalloc(&c, 4096)); CK(hipMalloc(&p, 4096)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&out, 64));
    CK(hipMemset(c, 0, 4096)); CK(hipMemset(p, 0, 4096));
    for (int it = 0; it < 6; ++it) {
        produce<<<80, 256>>>(c, p, it);
        consume<<<80, 256>>>(c, p, out, sink);
        unsigned long long h[6];
        CK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
        printf("loads (x10 ns): plain %llu | atomic %llu | atomic %llu | plain %llu | atomic %llu | plain %llu\n", h[0], h[1], h[2], h[3], h[4], h[5]);
    }
    return 0;
}
