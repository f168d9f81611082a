// Short chained kernels in which EVERY workgroup reads the same R KB that the kernel before wrote (the pattern of the
// two-launch light passes: all BH ranks read by all workgroups).  Time per kernel against R.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/bin/microbench_bcast tools/microbench_bcast.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int ROWS>  // rows of 1 KB per wave; the workgroup reads 4 ROWS KB
__global__ __launch_bounds__(256) void step(const int4 *__restrict__ in, int4 *__restrict__ out, int own_rows)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int4 v[ROWS];
#pragma unroll
    for (int q = 0; q < ROWS; ++q) v[q] = in[(wave + 4 * q) * 64 + lane];
    int s = 0;
#pragma unroll
    for (int q = 0; q < ROWS; ++q) s += v[q].x + v[q].y + v[q].z + v[q].w;
    // every workgroup writes its own row of the output (so that the next kernel's input is freshly written)
    for (int r = blockIdx.x; r < own_rows; r += gridDim.x) out[r * 64 + lane] = make_int4(s, wave, r, lane);
}
template <int ROWS>
int run(int4 *a, int4 *b, int nblk, hipStream_t s)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    const int chain = 64, reps = 10;
    for (int w = 0; w < 2; ++w) {
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; ++r)
            for (int k = 0; k < chain; ++k) step<ROWS><<<nblk, 256, 0, s>>>(k & 1 ? b : a, k & 1 ? a : b, 4 * ROWS);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("%3d workgroups, each reads the same %3d KB: %.2f us per kernel\n", nblk, 4 * ROWS, ms * 1e3 / (reps * chain));
    return 0;
}
int main()
{
    int4 *a, *b;
    CK(hipMalloc(&a, 1 << 20)); CK(hipMalloc(&b, 1 << 20));
    CK(hipMemset(a, 0, 1 << 20)); CK(hipMemset(b, 0, 1 << 20));
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int nblk : {80, 20, 8}) {
        run<1>(a, b, nblk, s); run<4>(a, b, nblk, s); run<8>(a, b, nblk, s); run<20>(a, b, nblk, s); run<32>(a, b, nblk, s);
    }
    return 0;
}
