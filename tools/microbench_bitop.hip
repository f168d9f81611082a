// Issue-rate microbenchmark for the bit-sliced pair loop: which 32-bit logic ops run at full rate on gfx950?
//   hipcc -O3 --offload-arch=gfx950 tools/microbench_bitop.hip -o tools/bin/mb_bitop && tools/bin/mb_bitop
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int kIters = 1024;

#define BODY8(INSN)                                                                                                     \
    asm volatile(INSN(0) INSN(1) INSN(2) INSN(3) INSN(4) INSN(5) INSN(6) INSN(7)                                        \
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)                       \
                 : "v"(p), "v"(q), "s"(s0));

#define K(NAME, INSN)                                                                                                   \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t s0)                                             \
    {                                                                                                                   \
        uint32_t p = threadIdx.x * 2654435761u, q = p ^ 0x55aa55aau;                                                     \
        uint32_t c0 = 1, c1 = 2, c2 = 3, c3 = 4, c4 = 5, c5 = 6, c6 = 7, c7 = 8;                                        \
        for (int it = 0; it < kIters; ++it) {                                                                           \
            BODY8(INSN) BODY8(INSN) BODY8(INSN) BODY8(INSN)                                                             \
        }                                                                                                               \
        out[blockIdx.x * 256 + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;                                   \
    }

#define I_BITOP3_VVV(n) "v_bitop3_b32 %" #n ", %8, %9, %" #n " bitop3:0x8e\n\t"
#define I_BITOP3_VSV(n) "v_bitop3_b32 %" #n ", %8, %10, %" #n " bitop3:0x8e\n\t"
#define I_BFI(n) "v_bfi_b32 %" #n ", %8, %9, %" #n "\n\t"
#define I_XOR(n) "v_xor_b32 %" #n ", %8, %" #n "\n\t"
#define I_ANDOR(n) "v_and_or_b32 %" #n ", %8, %9, %" #n "\n\t"
#define I_BCNT(n) "v_bcnt_u32_b32 %" #n ", %8, %" #n "\n\t"
#define I_ADD3(n) "v_add3_u32 %" #n ", %8, %9, %" #n "\n\t"
#define I_LSHLADD(n) "v_lshl_add_u32 %" #n ", %8, 16, %" #n "\n\t"
#define I_FMA(n) "v_fma_f32 %" #n ", %8, %9, %" #n "\n\t"
#define I_SUBU(n) "v_sub_u32 %" #n ", %8, %" #n "\n\t"
#define I_PKADDF(n) "v_pk_add_f32 %" #n ", %8, %" #n "\n\t"

K(k_bitop3_vvv, I_BITOP3_VVV)
K(k_bitop3_vsv, I_BITOP3_VSV)
K(k_bfi, I_BFI)
K(k_xor, I_XOR)
K(k_andor, I_ANDOR)
K(k_bcnt, I_BCNT)
K(k_add3, I_ADD3)
K(k_lshladd, I_LSHLADD)
K(k_fma, I_FMA)
K(k_subu, I_SUBU)

template <class F>
int run(const char *name, F kern, uint32_t *out, int wg_per_cu)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = 256 * 24;
    const size_t lds = (160 * 1024 / wg_per_cu) & ~size_t(1023);
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        kern<<<grid, 256, lds>>>(out, 0x12345678u);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    // wave-instructions per SIMD: grid*4 waves / 1024 SIMDs, each kIters*32 instructions
    const double insn_per_simd = static_cast<double>(grid) * 4 / 1024 * kIters * 32;
    printf("%-16s wg/cu=%d  %7.3f ms  %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", name, wg_per_cu, best,
           best * 1e6 / insn_per_simd, best * 1e6 / insn_per_simd * 2.4);
    return 0;
}

int main()
{
    uint32_t *out;
    CHECK(hipMalloc(&out, 256 * 24 * 256 * 4));
    for (int w : {1, 2, 3, 4, 5, 6, 8}) {
        if (run("v_bitop3 vvv", k_bitop3_vvv, out, w)) return 1;
        if (run("v_bcnt", k_bcnt, out, w)) return 1;
        if (run("v_fma_f32", k_fma, out, w)) return 1;
    }
    return 0;
}
