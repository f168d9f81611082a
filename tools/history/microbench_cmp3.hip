// Third issue-rate microbenchmark: how to feed the wave-uniform operand.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench_cmp3.hip -o /tmp/mb3 && /tmp/mb3
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int kIters = 4096;
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));

// A: v_sub_f32 clamp with FOUR DISTINCT SGPR operands + v_add_f32
__global__ __launch_bounds__(256) void k_sub_sgpr4(float *out, float s0, float s1, float s2, float s3, float s4, float s5, float s6, float s7)
{
    float b = threadIdx.x * 0.5f, c0 = 0, c1 = 0, c2 = 0, c3 = 0, t0, t1, t2, t3;
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            asm volatile(
                "v_sub_f32_e64 %4, %8, %12 clamp\n\tv_sub_f32_e64 %5, %9, %12 clamp\n\tv_sub_f32_e64 %6, %10, %12 clamp\n\tv_sub_f32_e64 %7, %11, %12 clamp\n\t"
                "v_add_f32_e32 %0, %4, %0\n\tv_add_f32_e32 %1, %5, %1\n\tv_add_f32_e32 %2, %6, %2\n\tv_add_f32_e32 %3, %7, %3"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                : "s"(s0), "s"(s1), "s"(s2), "s"(s3), "v"(b));
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0 + c1 + c2 + c3;
}

// B: v_pk_add_f32 clamp with DISTINCT SGPR PAIRS (two uniform a's per instruction), b broadcast by op_sel; pk accumulate
__global__ __launch_bounds__(256) void k_pk_sgprpair(float *out, float s0, float s1, float s2, float s3, float s4, float s5, float s6, float s7)
{
    float2v b = {threadIdx.x * 0.5f, 77.f};
    float2v c0 = {0, 0}, c1 = c0, c2 = c0, c3 = c0, t0, t1, t2, t3;
    float2v a0 = {s0, s1}, a1 = {s2, s3}, a2 = {s4, s5}, a3 = {s6, s7};
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            asm volatile(
                "v_pk_add_f32 %4, %8, %12 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1] clamp\n\t"
                "v_pk_add_f32 %5, %9, %12 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1] clamp\n\t"
                "v_pk_add_f32 %6, %10, %12 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1] clamp\n\t"
                "v_pk_add_f32 %7, %11, %12 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1] clamp\n\t"
                "v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %5\n\tv_pk_add_f32 %2, %2, %6\n\tv_pk_add_f32 %3, %3, %7"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                : "s"(a0), "s"(a1), "s"(a2), "s"(a3), "v"(b));
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0.x + c1.x + c2.x + c3.x + c0.y + c1.y + c2.y + c3.y;
}

// C: uniform operands from LDS broadcast reads (ds_read_b128, every lane the same address) into VGPRs,
//    then all-VGPR v_sub clamp + v_add, RJ = 2 lane genes per uniform value (like the pair kernel would do)
__global__ __launch_bounds__(256) void k_lds_feed(float *out, float s0, float s1, float s2, float s3, float s4, float s5, float s6, float s7)
{
    __shared__ float4v tile[512];
    for (int t = threadIdx.x; t < 512; t += 256) tile[t] = float4v{s0 + t, s1 + t, s2 + t, s3 + t};
    __syncthreads();
    float b0 = threadIdx.x * 0.5f, b1 = b0 + 300.f;
    float c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4v a = tile[(it * 8 + u) & 511];  // uniform address -> broadcast
            c[0] += __builtin_amdgcn_fmed3f(a.x - b0, 0.f, 1.f);
            c[1] += __builtin_amdgcn_fmed3f(a.y - b0, 0.f, 1.f);
            c[2] += __builtin_amdgcn_fmed3f(a.z - b0, 0.f, 1.f);
            c[3] += __builtin_amdgcn_fmed3f(a.w - b0, 0.f, 1.f);
            c[4] += __builtin_amdgcn_fmed3f(a.x - b1, 0.f, 1.f);
            c[5] += __builtin_amdgcn_fmed3f(a.y - b1, 0.f, 1.f);
            c[6] += __builtin_amdgcn_fmed3f(a.z - b1, 0.f, 1.f);
            c[7] += __builtin_amdgcn_fmed3f(a.w - b1, 0.f, 1.f);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = c[0] + c[1] + c[2] + c[3] + c[4] + c[5] + c[6] + c[7];
}

// D: v_mov_b32 from SGPR once, then RJ = 4 lane genes (all-VGPR compute)
__global__ __launch_bounds__(256) void k_mov_rj4(float *out, float s0, float s1, float s2, float s3, float s4, float s5, float s6, float s7)
{
    float b0 = threadIdx.x * 0.5f, b1 = b0 + 1, b2 = b0 + 2, b3 = b0 + 3, c0 = 0, c1 = 0, c2 = 0, c3 = 0, t0, t1, t2, t3, va;
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            asm volatile(
                "v_mov_b32_e32 %8, %9\n\t"
                "v_sub_f32_e64 %4, %8, %10 clamp\n\tv_sub_f32_e64 %5, %8, %11 clamp\n\tv_sub_f32_e64 %6, %8, %12 clamp\n\tv_sub_f32_e64 %7, %8, %13 clamp\n\t"
                "v_add_f32_e32 %0, %4, %0\n\tv_add_f32_e32 %1, %5, %1\n\tv_add_f32_e32 %2, %6, %2\n\tv_add_f32_e32 %3, %7, %3"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(va)
                : "s"((u & 1) ? s0 : s1), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0 + c1 + c2 + c3;
}

struct V { const char *name; void (*fn)(float *, float, float, float, float, float, float, float, float); double cmps; };

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *out;
    CHECK(hipMalloc(&out, sizeof(float) * 256 * cus * 8));
    V vs[] = {
        {"A v_sub clamp (4 distinct SGPR) + v_add", k_sub_sgpr4, 4},
        {"B v_pk_add clamp (SGPR pairs) + v_pk_add", k_pk_sgprpair, 8},
        {"C LDS broadcast b128 -> VGPR, RJ=2", k_lds_feed, 8},
        {"D v_mov from SGPR, RJ=4", k_mov_rj4, 4},
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (const V &v : vs)
        for (int bpc : {2, 4, 8}) {
            const int grid = cus * bpc;
            v.fn<<<grid, 256>>>(out, 100.f, 101.f, 102.f, 103.f, 104.f, 105.f, 106.f, 107.f);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            v.fn<<<grid, 256>>>(out, 100.f, 101.f, 102.f, 103.f, 104.f, 105.f, 106.f, 107.f);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double ns_blk = ms * 1e6 / (kIters * 8.0);
            printf("%-44s waves/SIMD %d: %7.3f ms  %.3f ns per 64-lane cmp (%.1f Tcmp/s)\n", v.name, bpc, ms,
                   ns_blk / (v.cmps * bpc), 64.0 * v.cmps * bpc / ns_blk * 1e-3 * cus * 4);
        }
    return 0;
}
