// Issue rate of the candidates for "add a popcount into the HIGH 16-bit half of a packed count register" (the odd rows
// of the pair loop, csrc/gen_k1_loop.py) on gfx950, measured like tools/microbench_bitop.hip: 8 independent
// accumulators per lane, 3 workgroups of 256 per CU.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench_accum.hip -o tools/bin/mb_accum && tools/bin/mb_accum
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
        uint32_t p = threadIdx.x & 31, q = 65536;                                                                       \
        uint32_t c0 = 1, c1 = 2, c2 = 3, c3 = 4, c4 = 5, c5 = 6, c6 = 7, c7 = 8;                                        \
        for (int it = 0; it < kIters; ++it) {                                                                           \
            BODY8(INSN) BODY8(INSN) BODY8(INSN) BODY8(INSN)                                                             \
        }                                                                                                               \
        out[blockIdx.x * 256 + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;                                   \
    }

#define I_BITOP3(n) "v_bitop3_b32 %" #n ", %8, %9, %" #n " bitop3:0x8e\n\t"
#define I_BCNT(n) "v_bcnt_u32_b32 %" #n ", %8, %" #n "\n\t"
#define I_SDWA(n) "v_add_u32_sdwa %" #n ", %8, %" #n " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n\t"
#define I_LSHLADD(n) "v_lshl_add_u32 %" #n ", %8, 16, %" #n "\n\t"
#define I_MAD24(n) "v_mad_u32_u24 %" #n ", %8, %9, %" #n "\n\t"
#define I_MAD24S(n) "v_mad_u32_u24 %" #n ", %8, %10, %" #n "\n\t"
#define I_ALIGNBIT(n) "v_alignbit_b32 %" #n ", %" #n ", %" #n ", 16\n\t"
#define I_ALIGNBITV(n) "v_alignbit_b32 %" #n ", %" #n ", %" #n ", %8\n\t"
#define I_PERM(n) "v_perm_b32 %" #n ", %" #n ", %" #n ", %9\n\t"
#define I_ADD(n) "v_add_u32 %" #n ", %8, %" #n "\n\t"
#define I_PKADD(n) "v_pk_add_u16 %" #n ", %8, %" #n " op_sel:[1,0] op_sel_hi:[0,1]\n\t"
#define I_PKADDP(n) "v_pk_add_u16 %" #n ", %8, %" #n "\n\t"
#define I_ADD3(n) "v_add3_u32 %" #n ", %8, %9, %" #n "\n\t"
#define I_LSHLOR(n) "v_lshl_or_b32 %" #n ", %8, 16, %" #n "\n\t"
#define I_MADU16(n) "v_mad_u16 %" #n ", %8, %9, %" #n " op_sel:[0,0,1,1]\n\t"

K(k_bitop3, I_BITOP3)
K(k_bcnt, I_BCNT)
K(k_sdwa, I_SDWA)
K(k_lshladd, I_LSHLADD)
K(k_mad24, I_MAD24)
K(k_mad24s, I_MAD24S)
K(k_alignbit, I_ALIGNBIT)
K(k_alignbitv, I_ALIGNBITV)
K(k_perm, I_PERM)
K(k_add, I_ADD)
K(k_pkadd, I_PKADD)
K(k_pkaddp, I_PKADDP)
K(k_add3, I_ADD3)
K(k_lshlor, I_LSHLOR)

template <class F>
int run(const char *name, F kern, uint32_t *out)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = 256 * 24;
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0));
        kern<<<grid, 256>>>(out, 65536u);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double insn_per_simd = static_cast<double>(grid) * 4 / 1024 * kIters * 32;
    printf("%-34s %7.3f ms  %.2f ns per wave-instruction per SIMD\n", name, best, best * 1e6 / insn_per_simd);
    return 0;
}

int main()
{
    uint32_t *out;
    CHECK(hipMalloc(&out, 256 * 24 * 256 * 4));
    for (int round = 0; round < 2; ++round) {
        if (run("v_bitop3_b32 vvv (reference)", k_bitop3, out)) return 1;
        if (run("v_bcnt_u32_b32", k_bcnt, out)) return 1;
        if (run("v_add_u32_sdwa WORD_0 -> hi", k_sdwa, out)) return 1;
        if (run("v_lshl_add_u32 16", k_lshladd, out)) return 1;
        if (run("v_mad_u32_u24 v,v,v", k_mad24, out)) return 1;
        if (run("v_mad_u32_u24 v,s,v", k_mad24s, out)) return 1;
        if (run("v_alignbit_b32 imm", k_alignbit, out)) return 1;
        if (run("v_alignbit_b32 vgpr shift", k_alignbitv, out)) return 1;
        if (run("v_perm_b32", k_perm, out)) return 1;
        if (run("v_add_u32", k_add, out)) return 1;
        if (run("v_pk_add_u16 op_sel lo->hi", k_pkadd, out)) return 1;
        if (run("v_pk_add_u16 plain", k_pkaddp, out)) return 1;
        if (run("v_add3_u32", k_add3, out)) return 1;
        if (run("v_lshl_or_b32 16", k_lshlor, out)) return 1;
    }
    return 0;
}
