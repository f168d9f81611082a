// Does straight-line code cost more than a loop in a short kernel?  N dependent v_fma per thread, once fully unrolled
// (8 N bytes of code, every line fetched once) and once as a loop of 16; 80 workgroups x 256 threads, kernels chained.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/bin/microbench_icache tools/microbench_icache.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int N, bool UNROLL>
__global__ __launch_bounds__(256) void step(const float *__restrict__ in, float *__restrict__ out, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    float v = in[i], c = in[(i + 1) % n];
    if (UNROLL) {
#pragma unroll
        for (int k = 0; k < N; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(c));
    } else {
#pragma unroll 1
        for (int k = 0; k < N / 16; ++k) {
#pragma unroll
            for (int u = 0; u < 16; ++u) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(c));
        }
    }
    out[i] = v;
}
template <int N, bool UNROLL>
int run(float *a, float *b, int n, hipStream_t s)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    const int chain = 64, reps = 10;
    for (int w = 0; w < 2; ++w) {
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; ++r)
            for (int k = 0; k < chain; ++k) step<N, UNROLL><<<n / 256, 256, 0, s>>>(k & 1 ? b : a, k & 1 ? a : b, n);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("N = %5d fma (%3d KB if unrolled) %-8s: %.2f us per kernel\n", N, N * 8 / 1024, UNROLL ? "unrolled" : "loop", ms * 1e3 / (reps * chain));
    return 0;
}
int main()
{
    const int n = 20480;
    float *a, *b;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    run<256, false>(a, b, n, s);  run<256, true>(a, b, n, s);
    run<1024, false>(a, b, n, s); run<1024, true>(a, b, n, s);
    run<4096, false>(a, b, n, s); run<4096, true>(a, b, n, s);
    run<8192, false>(a, b, n, s); run<8192, true>(a, b, n, s);
    run<16384, false>(a, b, n, s); run<16384, true>(a, b, n, s);
    return 0;
}
