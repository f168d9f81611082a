// Issue-rate microbenchmark for the candidate inner loops of the pair kernel (K1).
// Every variant performs "count += (b < a)" for wave-uniform a and per-lane b; the
// question is how many SIMD cycles one comparison costs.  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/microbench_cmp.hip -o /tmp/mb && /tmp/mb
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kIters = 4096;

// V0: v_cmp -> vcc, v_addc (4 interleaved chains, no nop)
__global__ __launch_bounds__(256) void v_cmp_addc(uint32_t *out, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3)
{
    uint32_t b = threadIdx.x * 2654435761u, c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            asm volatile(
                "v_cmp_gt_u32_e32 vcc, %4, %8\n\t"
                "v_cmp_gt_u32_e64 s[90:91], %5, %8\n\t"
                "v_cmp_gt_u32_e64 s[92:93], %6, %8\n\t"
                "v_cmp_gt_u32_e64 s[94:95], %7, %8\n\t"
                "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n\t"
                "v_addc_co_u32_e64 %1, vcc, 0, %1, s[90:91]\n\t"
                "v_addc_co_u32_e64 %2, vcc, 0, %2, s[92:93]\n\t"
                "v_addc_co_u32_e64 %3, vcc, 0, %3, s[94:95]"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
                : "s"(a0), "s"(a1), "s"(a2), "s"(a3), "v"(b)
                : "vcc", "s90", "s91", "s92", "s93", "s94", "s95");
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0 + c1 + c2 + c3;
}

// V1: v_cmp only (8 per block) -- how expensive is a VALU op that writes an SGPR pair?
__global__ __launch_bounds__(256) void v_cmp_only(uint32_t *out, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3)
{
    uint32_t b = threadIdx.x * 2654435761u, c0 = 0;
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            asm volatile(
                "v_cmp_gt_u32_e32 vcc, %1, %5\n\t"
                "v_cmp_gt_u32_e64 s[90:91], %2, %5\n\t"
                "v_cmp_gt_u32_e64 s[92:93], %3, %5\n\t"
                "v_cmp_gt_u32_e64 s[94:95], %4, %5\n\t"
                "v_cmp_gt_u32_e32 vcc, %2, %5\n\t"
                "v_cmp_gt_u32_e64 s[90:91], %3, %5\n\t"
                "v_cmp_gt_u32_e64 s[92:93], %4, %5\n\t"
                "v_cmp_gt_u32_e64 s[94:95], %1, %5"
                : "+v"(c0)
                : "s"(a0), "s"(a1), "s"(a2), "s"(a3), "v"(b)
                : "vcc", "s90", "s91", "s92", "s93", "s94", "s95");
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0;
}

// V2: plain 32-bit: sub, shift right 31, add (3 VALU, no SGPR traffic), a from SGPR
__global__ __launch_bounds__(256) void v_sub_shr_add(uint32_t *out, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3)
{
    uint32_t b = threadIdx.x & 1023, c0 = 0, c1 = 0, c2 = 0, c3 = 0, t0, t1, t2, t3;
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            asm volatile(
                "v_subrev_u32_e32 %4, %8, %12\n\t"
                "v_subrev_u32_e32 %5, %9, %12\n\t"
                "v_subrev_u32_e32 %6, %10, %12\n\t"
                "v_subrev_u32_e32 %7, %11, %12\n\t"
                "v_lshrrev_b32_e32 %4, 31, %4\n\t"
                "v_lshrrev_b32_e32 %5, 31, %5\n\t"
                "v_lshrrev_b32_e32 %6, 31, %6\n\t"
                "v_lshrrev_b32_e32 %7, 31, %7\n\t"
                "v_add_u32_e32 %0, %0, %4\n\t"
                "v_add_u32_e32 %1, %1, %5\n\t"
                "v_add_u32_e32 %2, %2, %6\n\t"
                "v_add_u32_e32 %3, %3, %7"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                : "s"(a0), "s"(a1), "s"(a2), "s"(a3), "v"(b));
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0 + c1 + c2 + c3;
}

// V3: packed 16-bit, two samples per op: saturating sub, min with 1, packed add (3 VALU per 2 comparisons)
__global__ __launch_bounds__(256) void v_pk_sat(uint32_t *out, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3)
{
    uint32_t b = (threadIdx.x & 1023) * 0x00010001u, c0 = 0, c1 = 0, c2 = 0, c3 = 0, t0, t1, t2, t3;
    const uint32_t one = 0x00010001u;
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            asm volatile(
                "v_pk_sub_u16 %4, %8, %12 clamp\n\t"
                "v_pk_sub_u16 %5, %9, %12 clamp\n\t"
                "v_pk_sub_u16 %6, %10, %12 clamp\n\t"
                "v_pk_sub_u16 %7, %11, %12 clamp\n\t"
                "v_pk_min_u16 %4, %4, %13\n\t"
                "v_pk_min_u16 %5, %5, %13\n\t"
                "v_pk_min_u16 %6, %6, %13\n\t"
                "v_pk_min_u16 %7, %7, %13\n\t"
                "v_pk_add_u16 %0, %0, %4\n\t"
                "v_pk_add_u16 %1, %1, %5\n\t"
                "v_pk_add_u16 %2, %2, %6\n\t"
                "v_pk_add_u16 %3, %3, %7"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                : "s"(a0), "s"(a1), "s"(a2), "s"(a3), "v"(b), "v"(one));
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0 + c1 + c2 + c3;
}

// V4: packed signed: sub_i16, then fused "acc - (d >> 15)" is not available; use pk_ashrrev + pk_sub
__global__ __launch_bounds__(256) void v_pk_sign(uint32_t *out, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3)
{
    uint32_t b = (threadIdx.x & 1023) * 0x00010001u, c0 = 0, c1 = 0, c2 = 0, c3 = 0, t0, t1, t2, t3;
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            asm volatile(
                "v_pk_sub_i16 %4, %12, %8\n\t"
                "v_pk_sub_i16 %5, %12, %9\n\t"
                "v_pk_sub_i16 %6, %12, %10\n\t"
                "v_pk_sub_i16 %7, %12, %11\n\t"
                "v_pk_lshrrev_b16 %4, 15, %4 op_sel_hi:[0,1]\n\t"
                "v_pk_lshrrev_b16 %5, 15, %5 op_sel_hi:[0,1]\n\t"
                "v_pk_lshrrev_b16 %6, 15, %6 op_sel_hi:[0,1]\n\t"
                "v_pk_lshrrev_b16 %7, 15, %7 op_sel_hi:[0,1]\n\t"
                "v_pk_add_u16 %0, %0, %4\n\t"
                "v_pk_add_u16 %1, %1, %5\n\t"
                "v_pk_add_u16 %2, %2, %6\n\t"
                "v_pk_add_u16 %3, %3, %7"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                : "s"(a0), "s"(a1), "s"(a2), "s"(a3), "v"(b));
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0 + c1 + c2 + c3;
}

// V5: v_cmp -> SGPR pair, popcount and add on the scalar unit (the ballot/popcount form; lane = sample)
__global__ __launch_bounds__(256) void v_cmp_bcnt(uint32_t *out, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3)
{
    uint32_t b = threadIdx.x * 2654435761u;
    uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            asm volatile(
                "v_cmp_gt_u32_e64 s[88:89], %4, %8\n\t"
                "v_cmp_gt_u32_e64 s[90:91], %5, %8\n\t"
                "v_cmp_gt_u32_e64 s[92:93], %6, %8\n\t"
                "v_cmp_gt_u32_e64 s[94:95], %7, %8\n\t"
                "s_nop 1\n\t"
                "s_bcnt1_i32_b64 s88, s[88:89]\n\t"
                "s_bcnt1_i32_b64 s90, s[90:91]\n\t"
                "s_bcnt1_i32_b64 s92, s[92:93]\n\t"
                "s_bcnt1_i32_b64 s94, s[94:95]\n\t"
                "s_add_u32 %0, %0, s88\n\t"
                "s_add_u32 %1, %1, s90\n\t"
                "s_add_u32 %2, %2, s92\n\t"
                "s_add_u32 %3, %3, s94"
                : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)
                : "s"(a0), "s"(a1), "s"(a2), "s"(a3), "v"(b)
                : "scc", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95");
    }
    out[blockIdx.x * 256 + threadIdx.x] = s0 + s1 + s2 + s3;
}

// V6: plain VALU baseline: 8 independent v_add_u32 (what "2 cycles per wave64 op" looks like)
__global__ __launch_bounds__(256) void v_add_only(uint32_t *out, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3)
{
    uint32_t c0 = threadIdx.x, c1 = 1, c2 = 2, c3 = 3;
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            asm volatile(
                "v_add_u32_e32 %0, %4, %0\n\t"
                "v_add_u32_e32 %1, %5, %1\n\t"
                "v_add_u32_e32 %2, %6, %2\n\t"
                "v_add_u32_e32 %3, %7, %3\n\t"
                "v_add_u32_e32 %0, %5, %0\n\t"
                "v_add_u32_e32 %1, %6, %1\n\t"
                "v_add_u32_e32 %2, %7, %2\n\t"
                "v_add_u32_e32 %3, %4, %3"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
                : "s"(a0), "s"(a1), "s"(a2), "s"(a3));
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0 + c1 + c2 + c3;
}

// V7: v_addc only, carry-in from a fixed SGPR pair (cost of the carry-consuming add)
__global__ __launch_bounds__(256) void v_addc_only(uint32_t *out, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3)
{
    uint32_t c0 = threadIdx.x, c1 = 1, c2 = 2, c3 = 3;
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            asm volatile(
                "v_addc_co_u32_e64 %0, s[92:93], 0, %0, s[90:91]\n\t"
                "v_addc_co_u32_e64 %1, s[92:93], 0, %1, s[90:91]\n\t"
                "v_addc_co_u32_e64 %2, s[92:93], 0, %2, s[90:91]\n\t"
                "v_addc_co_u32_e64 %3, s[92:93], 0, %3, s[90:91]\n\t"
                "v_addc_co_u32_e64 %0, s[92:93], 0, %0, s[90:91]\n\t"
                "v_addc_co_u32_e64 %1, s[92:93], 0, %1, s[90:91]\n\t"
                "v_addc_co_u32_e64 %2, s[92:93], 0, %2, s[90:91]\n\t"
                "v_addc_co_u32_e64 %3, s[92:93], 0, %3, s[90:91]"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
                : "s"(a0), "s"(a1), "s"(a2), "s"(a3)
                : "s90", "s91", "s92", "s93");
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0 + c1 + c2 + c3;
}

struct Variant {
    const char *name;
    void (*fn)(uint32_t *, uint32_t, uint32_t, uint32_t, uint32_t);
    double instr_per_block;  // VALU (+SALU) instructions per asm block
    double cmps_per_block;   // wave-wide comparisons-instructions' worth: comparisons per lane per block
};

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    uint32_t *out;
    const int blocks_per_cu[] = {1, 2, 4, 8};
    CHECK(hipMalloc(&out, sizeof(uint32_t) * 256 * cus * 8));
    Variant vs[] = {
        {"v_add_u32 x8 (baseline)", v_add_only, 8, 0},
        {"v_cmp->sgpr x8", v_cmp_only, 8, 8},
        {"v_addc(carry-in sgpr) x8", v_addc_only, 8, 0},
        {"v_cmp + v_addc (4 chains)", v_cmp_addc, 8, 4},
        {"v_sub + v_lshr + v_add (32-bit)", v_sub_shr_add, 12, 4},
        {"pk_sub_u16 clamp + pk_min + pk_add", v_pk_sat, 12, 8},
        {"pk_sub_i16 + pk_lshr + pk_add", v_pk_sign, 12, 8},
        {"v_cmp + s_bcnt1 + s_add (ballot form)", v_cmp_bcnt, 12, 4},
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (const Variant &v : vs) {
        for (int bpc : blocks_per_cu) {
            const int grid = cus * bpc;
            v.fn<<<grid, 256>>>(out, 100, 200, 300, 400);  // warm-up
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            v.fn<<<grid, 256>>>(out, 100, 200, 300, 400);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            // each SIMD hosts bpc waves (a block = 4 waves, one per SIMD); per wave kIters*8 asm blocks
            const double blocks = static_cast<double>(kIters) * 8;
            const double ns_per_wave_block = ms * 1e6 / blocks;  // wall ns per asm block with bpc waves sharing a SIMD
            const double ns_per_instr_simd = ns_per_wave_block / (v.instr_per_block * bpc);
            printf("%-40s waves/SIMD %d: %8.3f ms  %.3f ns per instr per SIMD", v.name, bpc, ms, ns_per_instr_simd);
            if (v.cmps_per_block > 0)
                printf("  -> %.3f ns per 64-lane comparison (%.2f Tcmp/s chip)", ns_per_wave_block / (v.cmps_per_block * bpc),
                       64.0 * v.cmps_per_block * bpc / ns_per_wave_block * 1e-3 * cus * 4);
            printf("\n");
        }
    }
    return 0;
}
