// Prototype of the bit-sliced pair-count loop (round 2): positions are stored as bit planes over 32-sample
// blocks, [pos_j < lo_i] for 32 samples is a borrow chain of NB v_bitop3_b32 (majority of ~p, u, lt) and the
// count is one v_bcnt_u32_b32.  Measures variants of the operand feed on a config-3-sized triangle and checks
// sampled pairs against a scalar loop.
//   hipcc -O3 --offload-arch=gfx950 tools/k1b_proto.hip -o tools/bin/k1b_proto && tools/bin/k1b_proto
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#ifndef SKEW
#define SKEW 1  // A planes stored one word off the P planes: operand banks never coincide (even-aligned tuples)
#endif
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)


// lt' = carry of (~p) + u + lt : after bits 0..NB-1 (LSB first) lt = [p < u] for 32 samples at once
__device__ __forceinline__ uint32_t step(uint32_t p, uint32_t u, uint32_t lt) { return __builtin_amdgcn_bitop3_b32(p, u, lt, 0x8e); }

// P  lane operand  [nblk][4][Gp] uint4 : words 4q..4q+3 of gene j in block b at ((b * 4 + q) * Gp + j)
// A  uniform operand [nblk][Gp][4] uint4 : the 16 words of gene i in block b at ((b * Gp + i) * 4 + q)
template <int RI, int RJ, int NB, int WPS, bool SMEM, int ILV, bool PACK, int kUnitH>
__global__ __launch_bounds__(256, WPS) void k1b(const uint4 *__restrict__ P, const uint4 *__restrict__ A, int Gp, int nblk,
                                                int n_units, const uint32_t *__restrict__ unit_map, uint32_t *__restrict__ out,
                                                int dump_it, int dump_jc, uint32_t *__restrict__ dump)
{
    constexpr int CJ = 256 * RJ;
    const int slot = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int u = (q / kUnitH) * 8 + slot;
    if (u >= n_units) return;
    const uint32_t um = unit_map[u];
    const int it = static_cast<int>(um & 0xFFFFu) * kUnitH + q % kUnitH;
    const int jc = static_cast<int>(um >> 16);
    const int i0 = it * RI;
    if (i0 >= Gp || jc * CJ >= Gp) return;
    if (jc * CJ + CJ - 1 < i0) return;  // strictly below the diagonal
    const int j0 = jc * CJ + threadIdx.x;

    constexpr int NA = PACK ? RI / 2 : RI;
    uint32_t acc[RJ][NA];
#pragma unroll
    for (int r = 0; r < RJ; ++r)
#pragma unroll
        for (int h = 0; h < NA; ++h) acc[r][h] = 0;

    constexpr int SB = 4;  // blocks per LDS stage
    __shared__ uint4 sm[SMEM ? 1 : 2 * SB * RI * 4];
    uint4 st[SB * RI * 4 / 256];
    auto stage_load = [&](int b0) {
#pragma unroll
        for (int e = 0; e < SB * RI * 4 / 256; ++e) {
            const int idx = threadIdx.x + 256 * e;
            const int blk = idx / (RI * 4), rem = idx % (RI * 4);
            const int b = min(b0 + blk, nblk - 1);
            st[e] = A[(static_cast<size_t>(b) * Gp + i0) * 4 + rem];
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int e = 0; e < SB * RI * 4 / 256; ++e) sm[buf * (SB * RI * 4) + threadIdx.x + 256 * e] = st[e];
    };
    int buf = 0;
    if (!SMEM) {
        stage_load(0);
        stage_store(0);
        __syncthreads();
    }
    for (int b0 = 0; b0 < nblk; b0 += SB) {
        const bool more = b0 + SB < nblk;
        if (!SMEM && more) stage_load(b0 + SB);
        const int nb = min(SB, nblk - b0);
        for (int bb = 0; bb < nb; ++bb) {
            const int b = b0 + bb;
            uint32_t p[RJ][16];
#pragma unroll
            for (int r = 0; r < RJ; ++r)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const uint4 v = P[(static_cast<size_t>(b) * 4 + qq) * Gp + j0 + 256 * r];
                    p[r][4 * qq] = v.x; p[r][4 * qq + 1] = v.y; p[r][4 * qq + 2] = v.z; p[r][4 * qq + 3] = v.w;
                }
            if (ILV == 2) {
#pragma clang loop unroll(full)
                for (int i = 0; i < RI; i += 2) {
                    uint32_t a[16], c[16];
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const uint4 v = sm[buf * (SB * RI * 4) + (bb * RI + i) * 4 + qq];
                        a[4 * qq] = v.x; a[4 * qq + 1] = v.y; a[4 * qq + 2] = v.z; a[4 * qq + 3] = v.w;
                        const uint4 w = sm[buf * (SB * RI * 4) + (bb * RI + i + 1) * 4 + qq];
                        c[4 * qq] = w.x; c[4 * qq + 1] = w.y; c[4 * qq + 2] = w.z; c[4 * qq + 3] = w.w;
                    }
                    uint32_t l0, l1, l2, l3, m0, m1, m2, m3;
                    asm volatile("v_bitop3_b32 %0, %8, %12, %8 bitop3:0x0c\n\tv_bitop3_b32 %1, %9, %12, %9 bitop3:0x0c\n\t"
                                 "v_bitop3_b32 %2, %10, %12, %10 bitop3:0x0c\n\tv_bitop3_b32 %3, %11, %12, %11 bitop3:0x0c\n\t"
                                 "v_bitop3_b32 %4, %8, %13, %8 bitop3:0x0c\n\tv_bitop3_b32 %5, %9, %13, %9 bitop3:0x0c\n\t"
                                 "v_bitop3_b32 %6, %10, %13, %10 bitop3:0x0c\n\tv_bitop3_b32 %7, %11, %13, %11 bitop3:0x0c"
                                 : "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3), "=&v"(m0), "=&v"(m1), "=&v"(m2), "=&v"(m3)
                                 : "v"(p[0][0]), "v"(p[1][0]), "v"(p[2][0]), "v"(p[3][0]), "v"(a[SKEW ? 15 : 0]), "v"(c[SKEW ? 15 : 0]));
#pragma unroll
                    for (int k = 1; k < NB; ++k)
                        asm volatile("v_bitop3_b32 %0, %8, %12, %0 bitop3:0x8e\n\tv_bitop3_b32 %1, %9, %12, %1 bitop3:0x8e\n\t"
                                     "v_bitop3_b32 %2, %10, %12, %2 bitop3:0x8e\n\tv_bitop3_b32 %3, %11, %12, %3 bitop3:0x8e\n\t"
                                     "v_bitop3_b32 %4, %8, %13, %4 bitop3:0x8e\n\tv_bitop3_b32 %5, %9, %13, %5 bitop3:0x8e\n\t"
                                     "v_bitop3_b32 %6, %10, %13, %6 bitop3:0x8e\n\tv_bitop3_b32 %7, %11, %13, %7 bitop3:0x8e"
                                     : "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3), "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3)
                                     : "v"(p[0][k]), "v"(p[1][k]), "v"(p[2][k]), "v"(p[3][k]), "v"(a[SKEW ? k - 1 : k]), "v"(c[SKEW ? k - 1 : k]));
                    const uint32_t l[4] = {l0, l1, l2, l3}, m[4] = {m0, m1, m2, m3};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        acc[r][i >> 1] += __builtin_popcount(l[r]);
                        acc[r][i >> 1] += static_cast<uint32_t>(__builtin_popcount(m[r])) << 16;
                    }
                }
            } else
#pragma unroll
            for (int i = 0; i < RI; ++i) {
                uint32_t a[16];
                if (SMEM) {
                    const uint4 *ap = A + (static_cast<size_t>(b) * Gp + i0 + i) * 4;  // wave-uniform address -> s_load
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const uint4 v = ap[qq];
                        a[4 * qq] = v.x; a[4 * qq + 1] = v.y; a[4 * qq + 2] = v.z; a[4 * qq + 3] = v.w;
                    }
                } else {
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const uint4 v = sm[buf * (SB * RI * 4) + (bb * RI + i) * 4 + qq];  // broadcast read
                        a[4 * qq] = v.x; a[4 * qq + 1] = v.y; a[4 * qq + 2] = v.z; a[4 * qq + 3] = v.w;
                    }
                }
                if (ILV && RJ == 4) {
                    // four independent borrow chains, one instruction each per bit: no result is consumed by the next instruction
                    uint32_t l0, l1, l2, l3;
                    asm volatile("v_bitop3_b32 %0, %4, %8, %4 bitop3:0x0c\n\tv_bitop3_b32 %1, %5, %8, %5 bitop3:0x0c\n\t"
                                 "v_bitop3_b32 %2, %6, %8, %6 bitop3:0x0c\n\tv_bitop3_b32 %3, %7, %8, %7 bitop3:0x0c"
                                 : "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
                                 : "v"(p[0][0]), "v"(p[1][0]), "v"(p[2][0]), "v"(p[3][0]), "v"(a[SKEW ? 15 : 0]));
#pragma unroll
                    for (int k = 1; k < NB; ++k)
                        asm volatile("v_bitop3_b32 %0, %4, %8, %0 bitop3:0x8e\n\tv_bitop3_b32 %1, %5, %8, %1 bitop3:0x8e\n\t"
                                     "v_bitop3_b32 %2, %6, %8, %2 bitop3:0x8e\n\tv_bitop3_b32 %3, %7, %8, %3 bitop3:0x8e"
                                     : "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3)
                                     : "v"(p[0][k]), "v"(p[1][k]), "v"(p[2][k]), "v"(p[3][k]), "v"(a[SKEW ? k - 1 : k]));
                    const uint32_t l[4] = {l0, l1, l2, l3};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (PACK) {
                            if (i & 1) acc[r][i >> 1] += static_cast<uint32_t>(__builtin_popcount(l[r])) << 16;
                            else acc[r][i >> 1] += __builtin_popcount(l[r]);
                        } else acc[r][i] += __builtin_popcount(l[r]);
                    }
                } else {
#pragma unroll
                for (int r = 0; r < RJ; ++r) {
                    uint32_t lt = 0;
#pragma unroll
                    for (int k = 0; k < NB; ++k) lt = step(p[r][k], a[k], lt);
                    if (PACK) {
                        if (i & 1) acc[r][i >> 1] += static_cast<uint32_t>(__builtin_popcount(lt)) << 16;
                        else acc[r][i >> 1] += __builtin_popcount(lt);
                    } else acc[r][i] += __builtin_popcount(lt);
                }
                }
            }
        }
        if (!SMEM) {
            if (more) stage_store(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }
    uint32_t x = 0;
#pragma unroll
    for (int r = 0; r < RJ; ++r)
#pragma unroll
        for (int h = 0; h < NA; ++h) x += acc[r][h] * (2 * h + 1 + r);
    out[blockIdx.x * 256 + threadIdx.x] = x;
    if (it == dump_it && jc == dump_jc) {
#pragma unroll
        for (int r = 0; r < RJ; ++r)
#pragma unroll
            for (int h = 0; h < RI / 2; ++h)
                dump[(r * 256 + threadIdx.x) * (RI / 2) + h] = PACK ? acc[r][h] : (acc[r][2 * h] | (acc[r][min(2 * h + 1, NA - 1)] << 16));
    }
}


// Second form: the uniform operand is read from LDS one 16-byte quad (4 bit planes) at a time, two quads ahead of
// its use (the LDS latency of a broadcast ds_read_b128 is otherwise exposed once per gene row at 3-4 waves per
// SIMD); RJ independent borrow chains per instruction group.
template <int RJ> struct Chains;
template <> struct Chains<2> {
    static __device__ __forceinline__ void first(uint32_t (&l)[2], const uint32_t (&p)[2], uint32_t a)
    {
        asm volatile("v_bitop3_b32 %0, %2, %4, %2 bitop3:0x0c\n\tv_bitop3_b32 %1, %3, %4, %3 bitop3:0x0c"
                     : "=&v"(l[0]), "=&v"(l[1]) : "v"(p[0]), "v"(p[1]), "v"(a));
    }
    static __device__ __forceinline__ void next(uint32_t (&l)[2], const uint32_t (&p)[2], uint32_t a)
    {
        asm volatile("v_bitop3_b32 %0, %2, %4, %0 bitop3:0x8e\n\tv_bitop3_b32 %1, %3, %4, %1 bitop3:0x8e"
                     : "+v"(l[0]), "+v"(l[1]) : "v"(p[0]), "v"(p[1]), "v"(a));
    }
};
template <> struct Chains<3> {
    static __device__ __forceinline__ void first(uint32_t (&l)[3], const uint32_t (&p)[3], uint32_t a)
    {
        asm volatile("v_bitop3_b32 %0, %3, %6, %3 bitop3:0x0c\n\tv_bitop3_b32 %1, %4, %6, %4 bitop3:0x0c\n\tv_bitop3_b32 %2, %5, %6, %5 bitop3:0x0c"
                     : "=&v"(l[0]), "=&v"(l[1]), "=&v"(l[2]) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(a));
    }
    static __device__ __forceinline__ void next(uint32_t (&l)[3], const uint32_t (&p)[3], uint32_t a)
    {
        asm volatile("v_bitop3_b32 %0, %3, %6, %0 bitop3:0x8e\n\tv_bitop3_b32 %1, %4, %6, %1 bitop3:0x8e\n\tv_bitop3_b32 %2, %5, %6, %2 bitop3:0x8e"
                     : "+v"(l[0]), "+v"(l[1]), "+v"(l[2]) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(a));
    }
};
template <> struct Chains<4> {
    static __device__ __forceinline__ void first(uint32_t (&l)[4], const uint32_t (&p)[4], uint32_t a)
    {
        asm volatile("v_bitop3_b32 %0, %4, %8, %4 bitop3:0x0c\n\tv_bitop3_b32 %1, %5, %8, %5 bitop3:0x0c\n\t"
                     "v_bitop3_b32 %2, %6, %8, %6 bitop3:0x0c\n\tv_bitop3_b32 %3, %7, %8, %7 bitop3:0x0c"
                     : "=&v"(l[0]), "=&v"(l[1]), "=&v"(l[2]), "=&v"(l[3]) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(a));
    }
    static __device__ __forceinline__ void next(uint32_t (&l)[4], const uint32_t (&p)[4], uint32_t a)
    {
        asm volatile("v_bitop3_b32 %0, %4, %8, %0 bitop3:0x8e\n\tv_bitop3_b32 %1, %5, %8, %1 bitop3:0x8e\n\t"
                     "v_bitop3_b32 %2, %6, %8, %2 bitop3:0x8e\n\tv_bitop3_b32 %3, %7, %8, %3 bitop3:0x8e"
                     : "+v"(l[0]), "+v"(l[1]), "+v"(l[2]), "+v"(l[3]) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(a));
    }
};

template <int RI, int RJ, int NB, int WPS, int kUnitH, int FEED>
__global__ __launch_bounds__(256, WPS) void k1q(const uint4 *__restrict__ P, const uint4 *__restrict__ A, int Gp, int nblk,
                                                int n_units, const uint32_t *__restrict__ unit_map, uint32_t *__restrict__ out,
                                                int dump_it, int dump_jc, uint32_t *__restrict__ dump)
{
    constexpr int CJ = 256 * RJ;
    constexpr int NQ = (NB + 3) / 4;  // quads of bit planes in use
    const int slot = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int u = (q / kUnitH) * 8 + slot;
    if (u >= n_units) return;
    const uint32_t um = unit_map[u];
    const int it = static_cast<int>(um & 0xFFFFu) * kUnitH + q % kUnitH;
    const int jc = static_cast<int>(um >> 16);
    const int i0 = it * RI;
    if (i0 >= Gp || jc * CJ >= Gp) return;
    if (jc * CJ + CJ - 1 < i0) return;
    const int j0 = jc * CJ + threadIdx.x;
    uint32_t acc[RJ][RI / 2];
#pragma unroll
    for (int r = 0; r < RJ; ++r)
#pragma unroll
        for (int h = 0; h < RI / 2; ++h) acc[r][h] = 0;
    constexpr int SB = 4;
    __shared__ uint4 sm[2 * SB * RI * 4];
    uint4 st[SB * RI * 4 / 256];
    auto stage_load = [&](int b0) {
#pragma unroll
        for (int e = 0; e < SB * RI * 4 / 256; ++e) {
            const int idx = threadIdx.x + 256 * e;
            const int blk = idx / (RI * 4), rem = idx % (RI * 4);
            const int b = min(b0 + blk, nblk - 1);
            st[e] = A[(static_cast<size_t>(b) * Gp + i0) * 4 + rem];
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int e = 0; e < SB * RI * 4 / 256; ++e) sm[buf * (SB * RI * 4) + threadIdx.x + 256 * e] = st[e];
    };
    int buf = 0;
    stage_load(0);
    stage_store(0);
    __syncthreads();
    for (int b0 = 0; b0 < nblk; b0 += SB) {
        const bool more = b0 + SB < nblk;
        if (more) stage_load(b0 + SB);
        const int nb = min(SB, nblk - b0);
        for (int bb = 0; bb < nb; ++bb) {
            const int b = b0 + bb;
            uint32_t p[16][RJ];
#pragma unroll
            for (int r = 0; r < RJ; ++r)
#pragma unroll
                for (int qq = 0; qq < NQ; ++qq) {
                    const uint4 v = P[(static_cast<size_t>(FEED == 2 ? 0 : b) * 4 + qq) * Gp + j0 + 256 * r];
                    p[4 * qq][r] = v.x; p[4 * qq + 1][r] = v.y; p[4 * qq + 2][r] = v.z; p[4 * qq + 3][r] = v.w;
                }
            const uint4 *sa = sm + buf * (SB * RI * 4) + bb * RI * 4;
            constexpr int NS = RI * NQ;  // quad steps of this block; step t = gene t / NQ, quad t % NQ
            uint4 ab[3];
            ab[0] = sa[0];
            ab[1] = sa[NQ > 1 ? 1 : 4];
            uint32_t l[RJ];
#pragma clang loop unroll(full)
            for (int i = 0; i < RI; ++i) {
#pragma clang loop unroll(full)
                for (int qq = 0; qq < NQ; ++qq) {
                    const int t = i * NQ + qq;
                    if (t + 2 < NS && (FEED != 0 || t + 2 < 3)) ab[(t + 2) % 3] = sa[((t + 2) / NQ) * 4 + (t + 2) % NQ];
                    const uint4 a4 = ab[t % 3];
                    const uint32_t aw[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma clang loop unroll(full)
                    for (int e = 0; e < 4; ++e) {
                        const int k = 4 * qq + e;
                        if (k >= NB) continue;
                        if (k == 0) Chains<RJ>::first(l, p[0], aw[0]);
                        else Chains<RJ>::next(l, p[k], aw[e]);
                    }
                }
#pragma clang loop unroll(full)
                for (int r = 0; r < RJ; ++r) {
                    if (i & 1) acc[r][i >> 1] += static_cast<uint32_t>(__builtin_popcount(l[r])) << 16;
                    else acc[r][i >> 1] += __builtin_popcount(l[r]);
                }
            }
        }
        if (more) stage_store(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    uint32_t x = 0;
#pragma unroll
    for (int r = 0; r < RJ; ++r)
#pragma unroll
        for (int h = 0; h < RI / 2; ++h) x += acc[r][h] * (2 * h + 1 + r);
    out[blockIdx.x * 256 + threadIdx.x] = x;
    if (it == dump_it && jc == dump_jc) {
#pragma unroll
        for (int r = 0; r < RJ; ++r)
#pragma unroll
            for (int h = 0; h < RI / 2; ++h) dump[(r * 256 + threadIdx.x) * (RI / 2) + h] = acc[r][h];
    }
}


// Third form: like k1b (four chains, packed counts, skewed A words) but the 16-byte LDS reads of the tile operand are
// issued TWO QUADS AHEAD of their use, across gene rows: quad order per gene = 3 (plane 0), 0, 1, 2; quad 3 stays
// live for planes 13, 14.  D = prefetch distance in quads.
template <int NB, int WPS, int D>
__global__ __launch_bounds__(256, WPS) void k1s(const uint4 *__restrict__ P, const uint4 *__restrict__ A, int Gp, int nblk,
                                                int n_units, const uint32_t *__restrict__ unit_map, uint32_t *__restrict__ out,
                                                int dump_it, int dump_jc, uint32_t *__restrict__ dump)
{
    constexpr int RI = 32, RJ = 4, CJ = 1024, kUnitH = 32;
    const int slot = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int u = (q / kUnitH) * 8 + slot;
    if (u >= n_units) return;
    const uint32_t um = unit_map[u];
    const int it = static_cast<int>(um & 0xFFFFu) * kUnitH + q % kUnitH;
    const int jc = static_cast<int>(um >> 16);
    const int i0 = it * RI;
    if (i0 >= Gp || jc * CJ >= Gp) return;
    if (jc * CJ + CJ - 1 < i0) return;
    const int j0 = jc * CJ + threadIdx.x;
    uint32_t acc[RJ][RI / 2];
#pragma unroll
    for (int r = 0; r < RJ; ++r)
#pragma unroll
        for (int h = 0; h < RI / 2; ++h) acc[r][h] = 0;
    constexpr int SB = 4;
    __shared__ uint4 sm[2 * SB * RI * 4];
    uint4 st[SB * RI * 4 / 256];
    auto stage_load = [&](int b0) {
#pragma unroll
        for (int e = 0; e < SB * RI * 4 / 256; ++e) {
            const int idx = threadIdx.x + 256 * e;
            const int blk = idx / (RI * 4), rem = idx % (RI * 4);
            const int b = min(b0 + blk, nblk - 1);
            st[e] = A[(static_cast<size_t>(b) * Gp + i0) * 4 + rem];
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int e = 0; e < SB * RI * 4 / 256; ++e) sm[buf * (SB * RI * 4) + threadIdx.x + 256 * e] = st[e];
    };
    int buf = 0;
    stage_load(0);
    stage_store(0);
    __syncthreads();
    for (int b0 = 0; b0 < nblk; b0 += SB) {
        const bool more = b0 + SB < nblk;
        if (more) stage_load(b0 + SB);
        const int nb = min(SB, nblk - b0);
        for (int bb = 0; bb < nb; ++bb) {
            const int b = b0 + bb;
            uint32_t p[16][RJ];
#pragma unroll
            for (int r = 0; r < RJ; ++r)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const uint4 v = P[(static_cast<size_t>(b) * 4 + qq) * Gp + j0 + 256 * r];
                    p[4 * qq][r] = v.x; p[4 * qq + 1][r] = v.y; p[4 * qq + 2][r] = v.z; p[4 * qq + 3][r] = v.w;
                }
            const uint4 *sa = sm + buf * (SB * RI * 4) + bb * RI * 4;
            // step t = 4 i + s, s = 0..3 reads quad qs[s] of gene i
            constexpr int NS = RI * 4;
            auto quad_of = [](int t) { const int s = t & 3; return (t >> 2) * 4 + (s == 0 ? 3 : s - 1); };
            uint4 ring[D + 1];
#pragma unroll
            for (int d = 0; d < D; ++d) ring[d] = sa[quad_of(d)];
            uint32_t l[4];
            uint4 q3 = {0, 0, 0, 0};
#pragma clang loop unroll(full)
            for (int i = 0; i < RI; ++i) {
#pragma clang loop unroll(full)
                for (int s4 = 0; s4 < 4; ++s4) {
                    const int t = 4 * i + s4;
                    if (t + D < NS) ring[(t + D) % (D + 1)] = sa[quad_of(t + D)];
                    const uint4 a4 = ring[t % (D + 1)];
                    if (s4 == 0) {  // quad 3: word 3 = plane 0 now, words 0, 1 = planes 13, 14 at the end
                        q3 = a4;
                        asm volatile("v_bitop3_b32 %0, %4, %8, %4 bitop3:0x0c\n\tv_bitop3_b32 %1, %5, %8, %5 bitop3:0x0c\n\t"
                                     "v_bitop3_b32 %2, %6, %8, %6 bitop3:0x0c\n\tv_bitop3_b32 %3, %7, %8, %7 bitop3:0x0c"
                                     : "=&v"(l[0]), "=&v"(l[1]), "=&v"(l[2]), "=&v"(l[3])
                                     : "v"(p[0][0]), "v"(p[0][1]), "v"(p[0][2]), "v"(p[0][3]), "v"(a4.w));
                    } else {
                        const uint32_t aw[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma clang loop unroll(full)
                        for (int e = 0; e < 4; ++e) {
                            const int k = 4 * (s4 - 1) + e + 1;  // planes 1..12
                            asm volatile("v_bitop3_b32 %0, %4, %8, %0 bitop3:0x8e\n\tv_bitop3_b32 %1, %5, %8, %1 bitop3:0x8e\n\t"
                                         "v_bitop3_b32 %2, %6, %8, %2 bitop3:0x8e\n\tv_bitop3_b32 %3, %7, %8, %3 bitop3:0x8e"
                                         : "+v"(l[0]), "+v"(l[1]), "+v"(l[2]), "+v"(l[3])
                                         : "v"(p[k][0]), "v"(p[k][1]), "v"(p[k][2]), "v"(p[k][3]), "v"(aw[e]));
                        }
                    }
                }
#pragma clang loop unroll(full)
                for (int k = 13; k < NB; ++k) {
                    const uint32_t w = k == 13 ? q3.x : q3.y;
                    asm volatile("v_bitop3_b32 %0, %4, %8, %0 bitop3:0x8e\n\tv_bitop3_b32 %1, %5, %8, %1 bitop3:0x8e\n\t"
                                 "v_bitop3_b32 %2, %6, %8, %2 bitop3:0x8e\n\tv_bitop3_b32 %3, %7, %8, %3 bitop3:0x8e"
                                 : "+v"(l[0]), "+v"(l[1]), "+v"(l[2]), "+v"(l[3])
                                 : "v"(p[k][0]), "v"(p[k][1]), "v"(p[k][2]), "v"(p[k][3]), "v"(w));
                }
#pragma clang loop unroll(full)
                for (int r = 0; r < RJ; ++r) {
                    if (i & 1) acc[r][i >> 1] += static_cast<uint32_t>(__builtin_popcount(l[r])) << 16;
                    else acc[r][i >> 1] += __builtin_popcount(l[r]);
                }
            }
        }
        if (more) stage_store(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    uint32_t x = 0;
#pragma unroll
    for (int r = 0; r < RJ; ++r)
#pragma unroll
        for (int h = 0; h < RI / 2; ++h) x += acc[r][h] * (2 * h + 1 + r);
    out[blockIdx.x * 256 + threadIdx.x] = x;
    if (it == dump_it && jc == dump_jc) {
#pragma unroll
        for (int r = 0; r < RJ; ++r)
#pragma unroll
            for (int h = 0; h < RI / 2; ++h) dump[(r * 256 + threadIdx.x) * (RI / 2) + h] = acc[r][h];
    }
}

static uint64_t splitmix(uint64_t &s)
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

template <int RI, int RJ, int NB, int WPS, bool SMEM, int ILV, bool PACK, int kUnitH, bool QUAD = false, int FEED = 1, int KS = 0>
int run(const char *name, const uint4 *dP, const uint4 *dA, int Gp, int G, int nblk, const std::vector<uint16_t> &pos,
        const std::vector<uint16_t> &lo, int S)
{
    constexpr int CJ = 256 * RJ;
    const int NJ = (Gp + CJ - 1) / CJ, NIT = Gp / RI;
    std::vector<uint32_t> units;
    for (int p = 0; p < NJ; ++p) {
        const int ni = std::min(NIT, (CJ / RI) * (p + 1));
        for (int r = 0; r * kUnitH < ni; ++r) units.push_back(static_cast<uint32_t>(p) << 16 | static_cast<uint32_t>(r));
    }
    uint32_t *dU, *dOut, *dDump;
    const unsigned grid = static_cast<unsigned>((units.size() + 7) / 8 * 8 * kUnitH);
    CHECK(hipMalloc(&dU, units.size() * 4));
    CHECK(hipMemcpy(dU, units.data(), units.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&dOut, static_cast<size_t>(grid) * 256 * 4));
    CHECK(hipMalloc(&dDump, static_cast<size_t>(RJ) * 256 * (RI / 2) * 4));
    const int dump_it = 37, dump_jc = std::min(NJ - 1, 3);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0));
        if constexpr (KS > 0) k1s<NB, WPS, KS><<<grid, 256>>>(dP, dA, Gp, nblk, static_cast<int>(units.size()), dU, dOut, dump_it, dump_jc, dDump);
        else if constexpr (QUAD) k1q<RI, RJ, NB, WPS, kUnitH, FEED><<<grid, 256>>>(dP, dA, Gp, nblk, static_cast<int>(units.size()), dU, dOut, dump_it, dump_jc, dDump);
        else k1b<RI, RJ, NB, WPS, SMEM, ILV, PACK, kUnitH><<<grid, 256>>>(dP, dA, Gp, nblk, static_cast<int>(units.size()), dU, dOut, dump_it, dump_jc, dDump);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    CHECK(hipGetLastError());
    std::vector<uint32_t> dump(static_cast<size_t>(RJ) * 256 * (RI / 2));
    CHECK(hipMemcpy(dump.data(), dDump, dump.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int r = 0; r < RJ; ++r)
        for (int t = 0; t < 256; t += 37)
            for (int ii = 0; ii < RI; ++ii) {
                const int i = dump_it * RI + ii, j = dump_jc * CJ + t + 256 * r;
                uint32_t n = 0;
                for (int s = 0; s < S; ++s) n += pos[static_cast<size_t>(s) * Gp + j] < lo[static_cast<size_t>(s) * Gp + i];
                const uint32_t w = dump[(r * 256 + t) * (RI / 2) + (ii >> 1)];
                const uint32_t got = (ii & 1) ? (w >> 16) : (w & 0xFFFFu);
                if (got != n) { if (bad < 5) printf("  MISMATCH i=%d j=%d got %u want %u\n", i, j, got, n); ++bad; }
            }
    // pairs actually computed: tiles on or above the diagonal
    double tiles = 0;
    for (int itl = 0; itl < NIT; ++itl)
        for (int jc = 0; jc < NJ; ++jc)
            if (jc * CJ + CJ - 1 >= itl * RI) tiles += 1;
    const double cmp_done = tiles * RI * CJ * S;
    const double cmp_useful = 0.5 * G * (G - 1.0) * S;
    printf("%-34s %7.3f ms  %6.2f Tcmp/s executed, %6.2f Tcmp/s of the G(G-1)/2 triangle  %s\n", name, best, cmp_done / best / 1e9,
           cmp_useful / best / 1e9, bad ? "WRONG" : "ok");
    (void)hipFree(dU); (void)hipFree(dOut); (void)hipFree(dDump);
    return 0;
}

#ifndef SKEW
#define SKEW 1
#endif
int main()
{
    const int G = 20000, Gp = 21504, S = 1024, nblk = S / 32, NB = 15;
    std::vector<uint16_t> pos(static_cast<size_t>(S) * Gp), lo(static_cast<size_t>(S) * Gp);
    uint64_t seed = 12345;
    for (size_t t = 0; t < pos.size(); t += 4) {
        uint64_t z = splitmix(seed);
        for (int e = 0; e < 4; ++e) { pos[t + e] = (z >> (16 * e)) & 0x7FFF; }
        z = splitmix(seed);
        for (int e = 0; e < 4; ++e) { lo[t + e] = (z >> (16 * e)) & 0x7FFF; }
    }
    // bit planes
    std::vector<uint32_t> P(static_cast<size_t>(nblk) * 4 * Gp * 4), A(static_cast<size_t>(nblk) * Gp * 16);
    for (int b = 0; b < nblk; ++b)
        for (int g = 0; g < Gp; ++g) {
            uint32_t wp[16] = {0}, wa[16] = {0};
            for (int s = 0; s < 32; ++s) {
                const uint16_t vp = pos[static_cast<size_t>(b * 32 + s) * Gp + g], va = lo[static_cast<size_t>(b * 32 + s) * Gp + g];
                for (int k = 0; k < NB; ++k) { wp[k] |= static_cast<uint32_t>((vp >> k) & 1) << s; wa[k] |= static_cast<uint32_t>((va >> k) & 1) << s; }
            }
            for (int k = 0; k < 16; ++k) {
                P[((static_cast<size_t>(b) * 4 + k / 4) * Gp + g) * 4 + (k & 3)] = wp[k];
                A[(static_cast<size_t>(b) * Gp + g) * 16 + (SKEW ? (k + 15) % 16 : k)] = wa[k];
            }
        }
    uint4 *dP, *dA;
    CHECK(hipMalloc(&dP, P.size() * 4)); CHECK(hipMalloc(&dA, A.size() * 4));
    CHECK(hipMemcpy(dP, P.data(), P.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    printf("G=%d Gp=%d S=%d NB=%d (one side of S samples; the real kernel runs two sides of S/2)\n", G, Gp, S, NB);
    if (run<32, 4, 15, 3, false, 1, true, 32>("LDS RJ=4 3w ilv pack skew", dP, dA, Gp, G, nblk, pos, lo, S)) return 1;
    if (run<32, 4, 15, 3, false, 1, true, 32, false, 1, 2>("k1s prefetch 2 quads, 3w", dP, dA, Gp, G, nblk, pos, lo, S)) return 1;
    if (run<32, 4, 15, 3, false, 1, true, 32, false, 1, 3>("k1s prefetch 3 quads, 3w", dP, dA, Gp, G, nblk, pos, lo, S)) return 1;
    if (run<32, 4, 15, 3, false, 1, true, 32, false, 1, 1>("k1s prefetch 1 quad, 3w", dP, dA, Gp, G, nblk, pos, lo, S)) return 1;
    return 0;
}
