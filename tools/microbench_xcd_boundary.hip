// Does data written by workgroups of ONE XCD stay in that XCD's L2 across a kernel boundary?  Kernel A: the workgroups that
// run on XCD `wx` write a buffer.  Kernel B (next launch, same stream): the workgroups on XCD `rx` time a dependent chain of
// loads from it (s_memrealtime, 100 MHz).  Same XCD against another XCD; 16 KB and 128 KB; plain loads.
// hipcc --offload-arch=gfx950 -O3 -o tools/bin/microbench_xcd_boundary tools/microbench_xcd_boundary.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ unsigned xcc_id() { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x)); return x & 7; }
__global__ void k_write(uint4 *buf, int n16, int wx, unsigned seed)
{
    if ((int)xcc_id() != wx) return;
    for (int i = (blockIdx.x >> 3) * 256 + threadIdx.x; i < n16; i += (gridDim.x >> 3) * 256) buf[i] = uint4{seed + (unsigned)i, 1, 2, 3};
}
__global__ void k_read(const uint4 *buf, int n16, int rx, unsigned long long *out, unsigned *sink)
{
    if ((int)xcc_id() != rx) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned acc = 0;
    // every thread: its share of the buffer, all loads independent (what a kernel that asks for its inputs first does)
    for (int i = (blockIdx.x >> 3) * 256 + threadIdx.x; i < n16; i += (gridDim.x >> 3) * 256) { const uint4 v = buf[i]; acc += v.x + v.y; }
    if (acc == 0xDEADBEEF) *sink = acc;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) out[blockIdx.x >> 3] = t1 - t0;
}
int main()
{
    const int WG = 8 * 16;   // 16 active workgroups on the chosen XCD
    uint4 *buf; unsigned long long *out; unsigned *sink;
    hipMalloc(&buf, 1 << 20); hipMalloc(&out, 64 * 8); hipMalloc(&sink, 4);
    for (int kb : {16, 128}) {
        const int n16 = kb * 1024 / 16;
        for (int rx : {0, 3}) {
            double sum = 0; int cnt = 0;
            for (int rep = 0; rep < 40; ++rep) {
                hipMemset(out, 0, 64 * 8);
                k_write<<<WG, 256>>>(buf, n16, 0, rep * 7919u);
                k_read<<<WG, 256>>>(buf, n16, rx, out, sink);
                hipDeviceSynchronize();
                std::vector<unsigned long long> h(16);
                hipMemcpy(h.data(), out, 16 * 8, hipMemcpyDeviceToHost);
                if (rep >= 8) for (auto v : h) if (v) { sum += (double)v; ++cnt; }
            }
            printf("%4d KB written on XCD 0, read on XCD %d in the next launch: %.2f us per workgroup (mean of %d)\n", kb, rx, cnt ? sum / cnt / 100.0 : -1.0, cnt);
        }
    }
    return 0;
}
