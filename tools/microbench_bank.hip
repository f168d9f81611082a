// Does v_bitop3_b32 pay for VGPR bank conflicts (register number mod 4)?  Explicit registers in inline asm.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench_bank.hip -o tools/bin/mb_bank && tools/bin/mb_bank
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int kIters = 2048;

// 8 chains; chain n uses dst/src2 = v[D0 + DS*n], src0 = v[A0 + AS*n], src1 = v[B0 + BS*(n%4)]
#define INS(d, a, b) "v_bitop3_b32 v" #d ", v" #a ", v" #b ", v" #d " bitop3:0x8e\n\t"
#define CLOB "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103"

#define KERNEL(NAME, BODY)                                                                   \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out)                               \
    {                                                                                        \
        asm volatile("v_mov_b32 v40, 0\n\t" ::: CLOB);                                       \
        for (int it = 0; it < kIters; ++it) { asm volatile(BODY BODY BODY BODY ::: CLOB); }  \
        uint32_t r;                                                                          \
        asm volatile("v_mov_b32 %0, v40" : "=v"(r)::CLOB);                                   \
        out[blockIdx.x * 256 + threadIdx.x] = r;                                             \
    }

// all three operands in different banks: d = 40+4n (bank 0), a = 73+4n (bank 1), b = 98,99.. hmm keep simple
// dst bank 0, src0 bank 1, src1 bank 2
#define B_DIFF INS(40, 73, 98) INS(44, 77, 98) INS(48, 81, 98) INS(52, 85, 98) INS(56, 89, 102) INS(60, 93, 102) INS(64, 97, 102) INS(68, 101, 102)
// dst bank 0, src0 bank 0, src1 bank 0
#define B_SAME INS(40, 72, 96) INS(44, 76, 96) INS(48, 80, 96) INS(52, 84, 96) INS(56, 88, 100) INS(60, 92, 100) INS(64, 96, 100) INS(68, 100, 96)
// dst bank 0, src0 bank 1, src1 bank 0  (src1 conflicts with dst/src2)
#define B_TWO INS(40, 73, 96) INS(44, 77, 96) INS(48, 81, 96) INS(52, 85, 96) INS(56, 89, 100) INS(60, 93, 100) INS(64, 97, 100) INS(68, 101, 100)
// dst bank 0, src0 bank 1, src1 bank 1  (src0 conflicts with src1)
#define B_AB INS(40, 73, 97) INS(44, 77, 97) INS(48, 81, 97) INS(52, 85, 97) INS(56, 89, 101) INS(60, 93, 101) INS(64, 97, 101) INS(68, 101, 97)

KERNEL(k_diff, B_DIFF)
KERNEL(k_same, B_SAME)
KERNEL(k_two, B_TWO)
KERNEL(k_ab, B_AB)

template <class F>
int run(const char *name, F kern, uint32_t *out, int wg_per_cu)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = 256 * wg_per_cu;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        kern<<<grid, 256>>>(out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double insn_per_simd = static_cast<double>(grid) * 4 / 1024 * kIters * 32;
    printf("%-28s wg/cu=%d  %7.3f ms  %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", name, wg_per_cu, best,
           best * 1e6 / insn_per_simd, best * 1e6 / insn_per_simd * 2.4);
    return 0;
}

int main()
{
    uint32_t *out;
    CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
    for (int w : {3, 8}) {
        if (run("dst/src0/src1 banks 0,1,2", k_diff, out, w)) return 1;
        if (run("all bank 0", k_same, out, w)) return 1;
        if (run("src1 = dst bank", k_two, out, w)) return 1;
        if (run("src0 = src1 bank", k_ab, out, w)) return 1;
    }
    return 0;
}
