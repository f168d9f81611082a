// How fast is a one-workgroup-per-sample LDS radix sort (rocprim::block_radix_sort, 1024 threads x IPT items)
// against the segmented device sort the transform uses?  20 000 keys of 15 varying bits per sample, 1000 samples.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <vector>
template <int IPT>
__global__ __launch_bounds__(1024) void k(const uint32_t *keys, uint32_t *okeys, uint16_t *ovals, int n, unsigned bits)
{
    using brs = rocprim::block_radix_sort<uint32_t, 1024, IPT, uint16_t>;
    extern __shared__ unsigned char smem[];
    typename brs::storage_type &st = *reinterpret_cast<typename brs::storage_type *>(smem);
    const uint32_t *in = keys + (size_t)blockIdx.x * n;
    uint32_t k[IPT]; uint16_t v[IPT];
#pragma unroll
    for (int e = 0; e < IPT; ++e) { int i = threadIdx.x * IPT + e; k[e] = i < n ? in[i] : 0xFFFFFFFFu; v[e] = (uint16_t)i; }
    brs().sort(k, v, st, 0, bits);
#pragma unroll
    for (int e = 0; e < IPT; ++e) { int i = threadIdx.x * IPT + e; if (i < n) { okeys[(size_t)blockIdx.x * n + i] = k[e]; ovals[(size_t)blockIdx.x * n + i] = v[e]; } }
}
int main() {
    const int n = 20000, S = 1000;
    std::vector<uint32_t> h((size_t)n * S);
    for (int s = 0; s < S; ++s) for (int i = 0; i < n; ++i) h[(size_t)s * n + i] = (uint32_t)(((uint64_t)i * 7919u + s * 104729u) % 20000u);
    uint32_t *d, *o; uint16_t *v;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o, h.size() * 4); hipMalloc(&v, h.size() * 2);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    using brs = rocprim::block_radix_sort<uint32_t, 1024, 20, uint16_t>;
    const size_t lds = sizeof(brs::storage_type);
    hipFuncSetAttribute((const void *)k<20>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (unsigned bits : {15u, 16u, 32u}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(k<20>, S, 1024, lds, 0, d, o, v, n, bits);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("block sort: %d samples x %d keys, %u bits, lds %zu B: %.3f ms  (%s)\n", S, n, bits, lds, ms, hipGetErrorString(hipGetLastError()));
        }
    }
    std::vector<uint32_t> r(n); hipMemcpy(r.data(), o + (size_t)3 * n, n * 4, hipMemcpyDeviceToHost);
    bool ok = true; for (int i = 1; i < n; ++i) ok &= r[i - 1] <= r[i];
    printf("sorted: %d\n", (int)ok);
    return 0;
}
