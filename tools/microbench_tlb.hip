// Cost of the first touches of a small dependent kernel: K separately allocated buffers against K regions of one
// allocation (address-translation misses at kernel start).
// build: hipcc -O3 --offload-arch=gfx950 -o tools/bin/microbench_tlb tools/microbench_tlb.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Ptrs { const int *p[16]; };
template <int K>
__global__ __launch_bounds__(256) void step(Ptrs in, int *__restrict__ out, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    int v = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) v += in.p[k][i];
    out[i] = v;
}
template <int K>
int run(const char *what, Ptrs a, Ptrs b, int *oa, int *ob, int n, hipStream_t s)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    const int chain = 64, reps = 20;
    for (int w = 0; w < 2; ++w) {
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; ++r)
            for (int k = 0; k < chain; ++k) { if (k & 1) step<K><<<n / 256, 256, 0, s>>>(b, oa, n); else step<K><<<n / 256, 256, 0, s>>>(a, ob, n); }
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("%-28s K=%2d: %.2f us per kernel\n", what, K, ms * 1e3 / (reps * chain));
    return 0;
}
int main()
{
    const int n = 20480;
    hipStream_t s; CK(hipStreamCreate(&s));
    Ptrs sa, sb, ja, jb;
    int *oa, *ob;
    CK(hipMalloc(&oa, n * 4)); CK(hipMalloc(&ob, n * 4));
    for (int k = 0; k < 16; ++k) {
        int *p, *q;
        CK(hipMalloc(&p, n * 4)); CK(hipMalloc(&q, n * 4));
        CK(hipMemset(p, 0, n * 4)); CK(hipMemset(q, 0, n * 4));
        sa.p[k] = p; sb.p[k] = q;
    }
    int *big;
    CK(hipMalloc(&big, 34 * n * 4));
    CK(hipMemset(big, 0, 34 * n * 4));
    for (int k = 0; k < 16; ++k) { ja.p[k] = big + (2 * k) * n; jb.p[k] = big + (2 * k + 1) * n; }
    run<1>("separate allocations", sa, sb, oa, ob, n, s);
    run<4>("separate allocations", sa, sb, oa, ob, n, s);
    run<8>("separate allocations", sa, sb, oa, ob, n, s);
    run<16>("separate allocations", sa, sb, oa, ob, n, s);
    run<1>("one allocation", ja, jb, big + 32 * n, big + 33 * n, n, s);
    run<4>("one allocation", ja, jb, big + 32 * n, big + 33 * n, n, s);
    run<8>("one allocation", ja, jb, big + 32 * n, big + 33 * n, n, s);
    run<16>("one allocation", ja, jb, big + 32 * n, big + 33 * n, n, s);
    return 0;
}
