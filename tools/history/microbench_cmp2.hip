// Second issue-rate microbenchmark: all-VGPR candidates for "count += (b < a)".
//   hipcc -O3 --offload-arch=gfx950 tools/microbench_cmp2.hip -o /tmp/mb2 && /tmp/mb2
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int kIters = 4096;

#define KERNEL(NAME, BODY)                                                                                \
    __global__ __launch_bounds__(256) void NAME(float *out, float fa, uint32_t ua)                        \
    {                                                                                                     \
        float a0 = fa + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = threadIdx.x * 0.5f;       \
        float c0 = 0, c1 = 0, c2 = 0, c3 = 0, t0 = 0, t1 = 0, t2 = 0, t3 = 0;                             \
        float one = 1.0f;                                                                                 \
        for (int it = 0; it < kIters; ++it) {                                                             \
            _Pragma("unroll") for (int u = 0; u < 8; ++u) asm volatile(                                   \
                BODY                                                                                      \
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3)           \
                : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b), "s"(fa), "s"(ua), "v"(one)                  \
                : "vcc", "s90", "s91", "s92", "s93", "s94", "s95");                                       \
        }                                                                                                 \
        out[blockIdx.x * 256 + threadIdx.x] = c0 + c1 + c2 + c3 + t0 + t1 + t2 + t3;                      \
    }

// operands: %0-3 acc, %4-7 tmp, %8-11 a (VGPR), %12 b (VGPR), %13 s float, %14 s uint, %15 one
KERNEL(k_add_u32_vvv, "v_add_u32_e32 %0, %8, %0\n\tv_add_u32_e32 %1, %9, %1\n\tv_add_u32_e32 %2, %10, %2\n\tv_add_u32_e32 %3, %11, %3\n\t"
                      "v_add_u32_e32 %0, %9, %0\n\tv_add_u32_e32 %1, %10, %1\n\tv_add_u32_e32 %2, %11, %2\n\tv_add_u32_e32 %3, %8, %3")
KERNEL(k_add_f32_vvv, "v_add_f32_e32 %0, %8, %0\n\tv_add_f32_e32 %1, %9, %1\n\tv_add_f32_e32 %2, %10, %2\n\tv_add_f32_e32 %3, %11, %3\n\t"
                      "v_add_f32_e32 %0, %9, %0\n\tv_add_f32_e32 %1, %10, %1\n\tv_add_f32_e32 %2, %11, %2\n\tv_add_f32_e32 %3, %8, %3")
KERNEL(k_sub_clamp_vvv, "v_sub_f32_e64 %4, %8, %12 clamp\n\tv_sub_f32_e64 %5, %9, %12 clamp\n\tv_sub_f32_e64 %6, %10, %12 clamp\n\tv_sub_f32_e64 %7, %11, %12 clamp\n\t"
                        "v_sub_f32_e64 %4, %9, %12 clamp\n\tv_sub_f32_e64 %5, %10, %12 clamp\n\tv_sub_f32_e64 %6, %11, %12 clamp\n\tv_sub_f32_e64 %7, %8, %12 clamp")
KERNEL(k_sub_clamp_add, "v_sub_f32_e64 %4, %8, %12 clamp\n\tv_sub_f32_e64 %5, %9, %12 clamp\n\tv_sub_f32_e64 %6, %10, %12 clamp\n\tv_sub_f32_e64 %7, %11, %12 clamp\n\t"
                        "v_add_f32_e32 %0, %4, %0\n\tv_add_f32_e32 %1, %5, %1\n\tv_add_f32_e32 %2, %6, %2\n\tv_add_f32_e32 %3, %7, %3")
KERNEL(k_sub_clamp_sgpr_add, "v_sub_f32_e64 %4, %13, %12 clamp\n\tv_sub_f32_e64 %5, %13, %12 clamp\n\tv_sub_f32_e64 %6, %13, %12 clamp\n\tv_sub_f32_e64 %7, %13, %12 clamp\n\t"
                             "v_add_f32_e32 %0, %4, %0\n\tv_add_f32_e32 %1, %5, %1\n\tv_add_f32_e32 %2, %6, %2\n\tv_add_f32_e32 %3, %7, %3")
KERNEL(k_int_sub_shr_add, "v_sub_u32_e32 %4, %12, %8\n\tv_sub_u32_e32 %5, %12, %9\n\tv_sub_u32_e32 %6, %12, %10\n\tv_sub_u32_e32 %7, %12, %11\n\t"
                          "v_lshrrev_b32_e32 %4, 31, %4\n\tv_lshrrev_b32_e32 %5, 31, %5\n\tv_lshrrev_b32_e32 %6, 31, %6\n\tv_lshrrev_b32_e32 %7, 31, %7\n\t"
                          "v_add_u32_e32 %0, %0, %4\n\tv_add_u32_e32 %1, %1, %5\n\tv_add_u32_e32 %2, %2, %6\n\tv_add_u32_e32 %3, %3, %7")
KERNEL(k_cmp_vv_addc, "v_cmp_gt_u32_e32 vcc, %8, %12\n\tv_cmp_gt_u32_e64 s[90:91], %9, %12\n\tv_cmp_gt_u32_e64 s[92:93], %10, %12\n\tv_cmp_gt_u32_e64 s[94:95], %11, %12\n\t"
                      "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n\tv_addc_co_u32_e64 %1, vcc, 0, %1, s[90:91]\n\tv_addc_co_u32_e64 %2, vcc, 0, %2, s[92:93]\n\tv_addc_co_u32_e64 %3, vcc, 0, %3, s[94:95]")
KERNEL(k_cmp_vv_only, "v_cmp_gt_u32_e32 vcc, %8, %12\n\tv_cmp_gt_u32_e64 s[90:91], %9, %12\n\tv_cmp_gt_u32_e64 s[92:93], %10, %12\n\tv_cmp_gt_u32_e64 s[94:95], %11, %12\n\t"
                      "v_cmp_gt_u32_e32 vcc, %9, %12\n\tv_cmp_gt_u32_e64 s[90:91], %10, %12\n\tv_cmp_gt_u32_e64 s[92:93], %11, %12\n\tv_cmp_gt_u32_e64 s[94:95], %8, %12")
KERNEL(k_mov_from_sgpr, "v_mov_b32_e32 %4, %13\n\tv_mov_b32_e32 %5, %14\n\tv_mov_b32_e32 %6, %13\n\tv_mov_b32_e32 %7, %14\n\t"
                        "v_mov_b32_e32 %4, %14\n\tv_mov_b32_e32 %5, %13\n\tv_mov_b32_e32 %6, %14\n\tv_mov_b32_e32 %7, %13")
// packed f32: two comparisons per instruction (VOP3P): d = clamp(a - b), acc += d

struct V { const char *name; void (*fn)(float *, float, uint32_t); double instr; double cmps; };

// packed variant written by hand (needs 64-bit register pairs)
__global__ __launch_bounds__(256) void k_pk(float *out, float fa, uint32_t ua)
{
    typedef float float2v __attribute__((ext_vector_type(2)));
    float2v a0 = {fa + threadIdx.x, fa + 1}, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f, b = {threadIdx.x * 0.5f, 3.0f};
    float2v c0 = {0, 0}, c1 = c0, c2 = c0, c3 = c0, t0, t1, t2, t3;
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            asm volatile(
                "v_pk_add_f32 %4, %8, %12 neg_lo:[0,1] neg_hi:[0,1] clamp\n\t"
                "v_pk_add_f32 %5, %9, %12 neg_lo:[0,1] neg_hi:[0,1] clamp\n\t"
                "v_pk_add_f32 %6, %10, %12 neg_lo:[0,1] neg_hi:[0,1] clamp\n\t"
                "v_pk_add_f32 %7, %11, %12 neg_lo:[0,1] neg_hi:[0,1] clamp\n\t"
                "v_pk_add_f32 %0, %0, %4\n\t"
                "v_pk_add_f32 %1, %1, %5\n\t"
                "v_pk_add_f32 %2, %2, %6\n\t"
                "v_pk_add_f32 %3, %3, %7"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b));
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0.x + c1.x + c2.x + c3.x + c0.y + c1.y + c2.y + c3.y;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *out;
    CHECK(hipMalloc(&out, sizeof(float) * 256 * cus * 8));
    V vs[] = {
        {"v_add_u32 v,v,v x8", k_add_u32_vvv, 8, 0},
        {"v_add_f32 v,v,v x8", k_add_f32_vvv, 8, 0},
        {"v_sub_f32 clamp v,v,v x8", k_sub_clamp_vvv, 8, 8},
        {"v_sub_f32 clamp + v_add_f32 (all VGPR)", k_sub_clamp_add, 8, 4},
        {"v_sub_f32 clamp (SGPR a) + v_add_f32", k_sub_clamp_sgpr_add, 8, 4},
        {"v_sub_u32 + v_lshr + v_add (all VGPR)", k_int_sub_shr_add, 12, 4},
        {"v_cmp v,v + v_addc", k_cmp_vv_addc, 8, 4},
        {"v_cmp v,v only x8", k_cmp_vv_only, 8, 8},
        {"v_mov_b32 v, s x8", k_mov_from_sgpr, 8, 0},
        {"v_pk_add_f32 neg clamp + v_pk_add_f32", k_pk, 8, 8},
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (const V &v : vs)
        for (int bpc : {2, 4, 8}) {
            const int grid = cus * bpc;
            v.fn<<<grid, 256>>>(out, 100.f, 7);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            v.fn<<<grid, 256>>>(out, 100.f, 7);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double ns_blk = ms * 1e6 / (kIters * 8.0);
            printf("%-42s waves/SIMD %d: %7.3f ms  %.3f ns/instr/SIMD", v.name, bpc, ms, ns_blk / (v.instr * bpc));
            if (v.cmps > 0) printf("  %.3f ns per 64-lane cmp (%.1f Tcmp/s)", ns_blk / (v.cmps * bpc), 64.0 * v.cmps * bpc / ns_blk * 1e-3 * cus * 4);
            printf("\n");
        }
    return 0;
}
