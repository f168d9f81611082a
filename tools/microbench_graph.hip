// Per-kernel cost of a chain of small dependent kernels: plain stream launches against one hipGraph launch.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/bin/microbench_graph tools/microbench_graph.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(256) void step(const int *__restrict__ in, int *__restrict__ out, int n, int work)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int v = in[(i * 7 + 3) % n];
    for (int k = 0; k < work; ++k) v = v * 1664525 + 1013904223;
    out[i] = v;
}
int main()
{
    const int n = 20480, chain = 64, reps = 20;
    int *a, *b;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
    CK(hipMemset(a, 1, n * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int work : {1, 2000}) {
        float ms;
        for (int w = 0; w < 2; ++w) {
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; ++r)
                for (int k = 0; k < chain; ++k) step<<<n / 256, 256, 0, s>>>(k & 1 ? b : a, k & 1 ? a : b, n, work);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("work %d: stream launches %.2f us per kernel\n", work, ms * 1e3 / (reps * chain));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int k = 0; k < chain; ++k) step<<<n / 256, 256, 0, s>>>(k & 1 ? b : a, k & 1 ? a : b, n, work);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 2; ++w) {
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("work %d: graph launches  %.2f us per kernel\n", work, ms * 1e3 / (reps * chain));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
