// What does one wave-instruction of the pair loop's instruction mix COST on a gfx950 SIMD, in core-clock cycles?  (round 6)
// The K1 roofline of rounds 1-5 prices v_bitop3_b32 at 2 cycles per wave64 instruction (SIMD-32) and v_bcnt_u32_b32 at 4.  This probe
// measures it with the kernel's own clocks -- s_memtime (the shader clock) and s_memrealtime (100 MHz, constant) -- so the answer does
// not depend on what the clock happens to be: K1's launch shape (one wave per workgroup, 3 workgroups per SIMD, 164 VGPRs), a long
// unrolled run of ONE instruction form with independent destinations and register banks chosen as in gen_k1_loop.py.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench_issue.hip -o tools/bin/mb_issue && tools/bin/mb_issue
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int kIters = 512, kBody = 64;   // instructions per wave = kIters * kBody

// 16 independent accumulators c0..c15 (consecutive VGPRs: banks rotate), two lane operands p (bank of c + 1) and q
#define R16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
#define BODY(INSN)                                                                                                                  \
    asm volatile(R16(INSN) R16(INSN) R16(INSN) R16(INSN)                                                                            \
                 : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(c[8]),      \
                   "+v"(c[9]), "+v"(c[10]), "+v"(c[11]), "+v"(c[12]), "+v"(c[13]), "+v"(c[14]), "+v"(c[15])                          \
                 : "v"(p), "v"(q), "s"(s0));

#define K(NAME, INSN)                                                                                                               \
    __global__ __launch_bounds__(64, 3) void NAME(uint32_t *out, unsigned long long *clk, uint32_t s0)                              \
    {                                                                                                                               \
        uint32_t p = threadIdx.x * 2654435761u, q = p ^ 0x55aa55aau;                                                                \
        uint32_t c[16];                                                                                                             \
        for (int i = 0; i < 16; ++i) c[i] = i + threadIdx.x;                                                                        \
        uint32_t pad[96];                                          /* (keeps the kernel at K1's three waves per SIMD) */              \
        for (int i = 0; i < 96; ++i) pad[i] = out[(threadIdx.x + i * 64) & 4095];                                                    \
        __builtin_amdgcn_s_waitcnt(0);                                                                                              \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();                          \
        for (int it = 0; it < kIters; ++it) { BODY(INSN) }                                                                          \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();                          \
        uint32_t sum = 0;                                                                                                           \
        for (int i = 0; i < 16; ++i) sum += c[i];                                                                                   \
        for (int i = 0; i < 96; ++i) sum += pad[i];                                                                                 \
        out[blockIdx.x * 64 + threadIdx.x] = sum;                                                                                   \
        if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }                                 \
    }

#define I_ADD(n) "v_add_u32 %" #n ", %16, %" #n "\n\t"
#define I_XOR(n) "v_xor_b32 %" #n ", %16, %" #n "\n\t"
#define I_FMAC(n) "v_fmac_f32 %" #n ", %16, %17\n\t"
#define I_FMA(n) "v_fma_f32 %" #n ", %16, %17, %" #n "\n\t"
#define I_BITOP3(n) "v_bitop3_b32 %" #n ", %16, %17, %" #n " bitop3:0x8e\n\t"
#define I_BITOP3_2SRC(n) "v_bitop3_b32 %" #n ", %16, %16, %" #n " bitop3:0x8e\n\t"
#define I_BITOP3_S(n) "v_bitop3_b32 %" #n ", %16, %18, %" #n " bitop3:0x8e\n\t"
#define I_BFI(n) "v_bfi_b32 %" #n ", %16, %17, %" #n "\n\t"
#define I_ANDOR(n) "v_and_or_b32 %" #n ", %16, %17, %" #n "\n\t"
#define I_BCNT(n) "v_bcnt_u32_b32 %" #n ", %16, %" #n "\n\t"
#define I_ADD_E64(n) "v_add_u32_e64 %" #n ", %16, %" #n "\n\t"
#define I_MOV(n) "v_mov_b32 %" #n ", %16\n\t"

K(k_add, I_ADD)
K(k_xor, I_XOR)
K(k_fmac, I_FMAC)
K(k_fma, I_FMA)
K(k_bitop3, I_BITOP3)
K(k_bitop3_2src, I_BITOP3_2SRC)
K(k_bitop3_s, I_BITOP3_S)
K(k_bfi, I_BFI)
K(k_andor, I_ANDOR)
K(k_bcnt, I_BCNT)
K(k_add_e64, I_ADD_E64)
K(k_mov, I_MOV)

template <class F>
int run(const char *name, const char *enc, F kern, uint32_t *out, unsigned long long *clk, int waves_per_simd)
{
    const int grid = 1024 * waves_per_simd;   // one wave per workgroup: every SIMD gets waves_per_simd of them
    std::vector<unsigned long long> h(2 * grid);
    double best_cyc = 1e30, best_ns = 0, mhz = 0;
    for (int rep = 0; rep < 4; ++rep) {
        // (dynamic LDS limits a CU to 4 * waves_per_simd one-wave workgroups: the SIMDs get waves_per_simd each)
        kern<<<grid, 64, (160 * 1024 / (4 * waves_per_simd)) & ~1023>>>(out, clk, 0x12345678u);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost));
        double cyc = 0, rt = 0;
        for (int b = 0; b < grid; ++b) { cyc += static_cast<double>(h[2 * b]); rt += static_cast<double>(h[2 * b + 1]); }
        cyc /= grid; rt /= grid;   // per wave: shader-clock cycles and 100 MHz ticks for kIters * kBody instructions
        // a SIMD runs waves_per_simd such streams interleaved: cycles per wave-instruction per SIMD
        const double per = cyc / (static_cast<double>(kIters) * kBody * waves_per_simd);
        if (per < best_cyc) { best_cyc = per; best_ns = rt * 10.0 / (static_cast<double>(kIters) * kBody * waves_per_simd); mhz = cyc / (rt * 10.0) * 1e3; }
    }
    printf("%-34s %-5s waves/SIMD %d: %5.2f shader-clock cycles = %5.2f ns per wave-instruction per SIMD (clock %4.0f MHz)\n", name, enc, waves_per_simd, best_cyc, best_ns, mhz);
    return 0;
}

int main()
{
    uint32_t *out; unsigned long long *clk;
    CHECK(hipMalloc(&out, 1024 * 8 * 64 * 4 + 4096 * 4));
    CHECK(hipMemset(out, 0, 1024 * 8 * 64 * 4 + 4096 * 4));
    CHECK(hipMalloc(&clk, 1024 * 8 * 2 * 8));
    for (int w : {3, 1}) {
        if (run("v_add_u32 (2 sources)", "VOP2", k_add, out, clk, w)) return 1;
        if (run("v_xor_b32 (2 sources)", "VOP2", k_xor, out, clk, w)) return 1;
        if (run("v_mov_b32", "VOP1", k_mov, out, clk, w)) return 1;
        if (run("v_add_u32_e64 (2 sources)", "VOP3", k_add_e64, out, clk, w)) return 1;
        if (run("v_fmac_f32 (3 operands, dst = src2)", "VOP2", k_fmac, out, clk, w)) return 1;
        if (run("v_fma_f32 (3 sources)", "VOP3", k_fma, out, clk, w)) return 1;
        if (run("v_bitop3_b32 (3 VGPR sources)", "VOP3", k_bitop3, out, clk, w)) return 1;
        if (run("v_bitop3_b32 (src0 == src1)", "VOP3", k_bitop3_2src, out, clk, w)) return 1;
        if (run("v_bitop3_b32 (an SGPR source)", "VOP3", k_bitop3_s, out, clk, w)) return 1;
        if (run("v_bfi_b32 (3 sources)", "VOP3", k_bfi, out, clk, w)) return 1;
        if (run("v_and_or_b32 (3 sources)", "VOP3", k_andor, out, clk, w)) return 1;
        if (run("v_bcnt_u32_b32 (2 sources)", "VOP3", k_bcnt, out, clk, w)) return 1;
    }
    return 0;
}
