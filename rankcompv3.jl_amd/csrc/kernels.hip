// Hand-written gfx950 kernels of the REO hot path.
//
//   K1  k1w_pairs   pair compare -> per-group counts -> stable-REO class ->
//                   4 bit planes per ordered pair      (src/RankCompV3.jl:363-392)
//                   wave form (round 3, the default for two groups): one wave per workgroup, the count loop a generated,
//                   hand-scheduled asm statement (gen_k1_loop.py -> k1_loop_gen.inc), up to 262 143 genes
//                   (an item = 32 gene rows x 256 genes x one side; the items of a launch's last round as two 16-row halves)
//       k1w_pairs_wide   the wave form with 32-bit totals (more than 65 535 samples, two groups)
//       k1w_group_counts + k1_classify: one-vs-rest over more than two groups (:375-390) -- every group counted once
//                   (the wave form's loop, once per group and item), counts kept in HBM, one cheap classification per
//                   comparison;  k1w_pairs_multi: the recounting form of that above 65 535 genes
//       k1_pairs, k1_pairs_wide, k1_group_counts: the workgroup forms of round 2 (REO_K1_WAVE=0; more than two groups
//                   without shared counts; more than two groups with more than 65 535 samples)
//   K2  k2_tally    class table x reference mask -> per-gene tallies   (:403)
//       (delta form: the same launch updates the counters from the rows of the genes whose mask bit changed)
//   K3  k3_*        McCullagh test, trimmed std, normal p, BH, new mask, loop control (:404-425,225-259): the
//                   sorting path; kl_head + kl_rank the light passes, two launches each (no sort: quantile
//                   windows + BH cut from per-XCD histograms + a list of the genes near the cut); kl_persist the same
//                   as one persistent launch (opt-in)
//       x_pack / x_expand_*  exchange of the class table between shards: all-gather of forward words
//
// All file:line citations are relative to /root/reference.
//
// Data layout in HBM
//   P   uint4 [nblk][4][Gp]  bit planes of pos (position of gene g in its sample's sorted order) over blocks of
//                   32 samples: planes 4q..4q+3 of gene g in block b at (b * 4 + q) * Gp + g (lane operand;
//                   groups padded to whole blocks)
//   AL  uint4 [nblk][Gp][4]  the 16 plane words of lo (first position of g's tie band) of gene g in block b,
//                   plane k in word (k + 15) % 16 (tile operand: staged through LDS)
//   AH  likewise for hi (one past the last position of g's tie band); padding samples have lo = hi = 0
//   (more than 65 535 genes: the big layout of transform.hip, t_slice_big -- P [nblk][5][Gp], AL / AH [nblk][Gp][8], plane k in word k)
//   table u32 [G][4][Wp]  bit planes cL cH tL tH of row i: bit j of plane cL is
//                   set iff pair (i,j) is "i<j stable" in ctrl (ic==1), cH iff
//                   ic==3, tL/tH likewise for treat.  4 bits per ORDERED pair,
//                   the diagonal is all-zero (like the reference's R, :363).
//
// Wave = 64 lanes everywhere; no warp-32 idiom is used.
#include <algorithm>
#include <chrono>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>

#include "reo_internal.h"

namespace reo {

namespace {

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// Loads and stores of data that travels between workgroups of ONE launch (the persistent light kernel): relaxed
// agent-scope atomics = global_load / global_store with sc1, which bypass the CU's L1 and are served coherently
// across the XCDs' L2s (measured hand-off form: MI355X_MICROARCH.md, "Valid forms"; tools/microbench_gridbar.hip reads
// every workgroup's value in every round and finds none stale).  COH = false: plain accesses (separate launches).
template <bool COH, class T>
__device__ __forceinline__ T ldc(const T *p)
{
    if (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <bool COH, class T>
__device__ __forceinline__ void stc(T *p, T v)
{
    if (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// Workgroup barrier for data exchanged through LDS only.  __syncthreads() also waits for every outstanding global
// load and STORE of the wave (s_waitcnt vmcnt(0): a write round trip of a microsecond or two in these latency-bound
// kernels); this waits for the LDS traffic alone.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Binomial(n_eq, 1/2) draw standing in for n_eq calls of rand(Bool) in
// is_greater (:72-73): one bit of a counter-based stream keyed by
// (seed, i, j, group) per tied sample.
__device__ uint32_t tie_wins(uint64_t seed, uint32_t i, uint32_t j, uint32_t g, uint32_t n_eq)
{
    uint64_t base = mix64(seed ^ mix64((static_cast<uint64_t>(i) << 32) | j)) + (static_cast<uint64_t>(g) << 40);
    uint32_t wins = 0;
    for (uint64_t w = 0; n_eq > 0; ++w) {
        uint64_t bits = mix64(base + w);
        uint32_t take = n_eq < 64 ? n_eq : 64;
        if (take < 64) bits &= (1ULL << take) - 1;
        wins += __popcll(bits);
        n_eq -= take;
    }
    return wins;
}

// ---------------------------------------------------------------------------
// K1 inner loop, bit-sliced.  The transform hands over pos / lo / hi as bit planes over blocks of 32 samples
// (plane k, word of a block: bit s = bit k of the 16-bit number of sample 32 b + s).  For one pair and one block
//     lt = [pos_j < lo_i] for 32 samples  =  the borrow of lo_i - pos_j, bit by bit from the LSB:
//     lt <- majority(~p_k, u_k, lt)                       one v_bitop3_b32 (truth table 0x8e) per bit plane
//     n_gt += popcount(lt)                                one v_bcnt_u32_b32 per 32 samples
// i.e. (NB + 1) / 32 instructions per comparison instead of the two packed float ops of round 1.  Measured on
// MI355X (tools/microbench_bitop.hip, tools/history/k1b_proto.hip): v_bitop3_b32 with three VGPR sources issues at full
// rate (1.0-1.2 ns per wave-instruction per SIMD), at half rate with an SGPR source or when its three sources share
// a VGPR bank, and v_bcnt / v_lshl_add at half rate; v_bfi_b32 + v_xor_b32 (two ops per bit) is no faster than
// the float form.  Lane = gene j (RJ genes per lane, 64 apart), the 32 genes i of the tile are wave-uniform and
// their planes are staged through LDS (ordinary 16-byte vector loads one stage ahead, double-buffered, one
// barrier per stage; read back with broadcast ds_read_b128).  Counts are kept packed, two 16-bit counts per
// register (rows 2h and 2h+1): a side never has more than 65 535 samples.
// n_gt(i,j) = #{s : pos_j < lo_i},  n_ge(i,j) = #{s : pos_j < hi_i}  (ties: n_eq = n_ge - n_gt).
constexpr int kStageB = 4;  // 32-sample blocks per LDS stage of the tile operand

// four independent borrow chains per bit plane (no result is consumed by the next instruction):
// chain c combines lane operand p[c] with tile operand a[c]
__device__ __forceinline__ void chains_first(uint32_t (&l)[4], uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3,
                                             uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3)
{
    asm volatile("v_bitop3_b32 %0, %4, %8, %4 bitop3:0x0c\n\t"    // ~p & u
                 "v_bitop3_b32 %1, %5, %9, %5 bitop3:0x0c\n\t"
                 "v_bitop3_b32 %2, %6, %10, %6 bitop3:0x0c\n\t"
                 "v_bitop3_b32 %3, %7, %11, %7 bitop3:0x0c"
                 : "=&v"(l[0]), "=&v"(l[1]), "=&v"(l[2]), "=&v"(l[3])
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(a0), "v"(a1), "v"(a2), "v"(a3));
}

__device__ __forceinline__ void chains_next(uint32_t (&l)[4], uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3,
                                            uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3)
{
    asm volatile("v_bitop3_b32 %0, %4, %8, %0 bitop3:0x8e\n\t"    // majority(~p, u, lt)
                 "v_bitop3_b32 %1, %5, %9, %1 bitop3:0x8e\n\t"
                 "v_bitop3_b32 %2, %6, %10, %2 bitop3:0x8e\n\t"
                 "v_bitop3_b32 %3, %7, %11, %3 bitop3:0x8e"
                 : "+v"(l[0]), "+v"(l[1]), "+v"(l[2]), "+v"(l[3])
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(a0), "v"(a1), "v"(a2), "v"(a3));
}

// the 16 plane words of one gene as registers; A planes are stored one word off (plane k in word (k + 15) % 16)
struct Planes16 {
    uint32_t w[16];
    __device__ __forceinline__ void set(int q, uint4 v) { w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w; }
};
__device__ __forceinline__ constexpr int a_word(int k) { return (k + 15) & 15; }

// unsigned count of row ii (0..31) from the packed accumulators
__device__ __forceinline__ uint32_t unpack16(const uint32_t (&acc)[kTileI / 2], int ii)
{
    return (ii & 1) ? (acc[ii >> 1] >> 16) : (acc[ii >> 1] & 0xFFFFu);
}

// Whole-workgroup pair loop over blocks [bb, be) (every thread must call it: it has barriers).
// gt[r][h] / ge[r][h]: packed counts of rows 2h, 2h+1 against the lane's gene r.  Tie-free: 4 genes per lane, one
// chain each; with ties: 2 genes per lane, two chains each (lo and hi).  idle (wave-uniform): none of this wave's
// genes forms a real pair with the tile -- the wave only helps staging and keeps the barriers.
template <int RJ, int NB, bool TIES>
__device__ __forceinline__ void count_pass(const uint4 *__restrict__ P, const uint4 *__restrict__ AL, const uint4 *__restrict__ AH,
                                           int Gp, int i0, int jl, int bb, int be, uint32_t (&gt)[RJ][kTileI / 2],
                                           uint32_t (&ge)[TIES ? RJ : 1][kTileI / 2], uint4 *sm_lo, uint4 *sm_hi, bool idle)
{
    static_assert((TIES && RJ == 2) || (!TIES && RJ == 4), "four chains per bit plane");
    constexpr int RI = kTileI, NQ = (NB + 3) / 4;
    constexpr int kStageQ = kStageB * RI * 4;        // uint4 per stage
    constexpr int kPerThread = kStageQ / 256;
#pragma unroll
    for (int r = 0; r < RJ; ++r)
#pragma unroll
        for (int h = 0; h < RI / 2; ++h) { gt[r][h] = 0; if (TIES) ge[r][h] = 0; }
    if (bb >= be) return;
    uint4 sl[kPerThread], sh[TIES ? kPerThread : 1];
    auto stage_load = [&](int b0) {
#pragma unroll
        for (int e = 0; e < kPerThread; ++e) {
            const int idx = threadIdx.x + 256 * e;
            const int b = min(b0 + idx / (RI * 4), be - 1);
            const size_t o = (static_cast<size_t>(b) * Gp + i0) * 4 + idx % (RI * 4);
            sl[e] = AL[o];
            if (TIES) sh[e] = AH[o];
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int e = 0; e < kPerThread; ++e) {
            sm_lo[buf * kStageQ + threadIdx.x + 256 * e] = sl[e];
            if (TIES) sm_hi[buf * kStageQ + threadIdx.x + 256 * e] = sh[e];
        }
    };
    stage_load(bb);
    __syncthreads();  // the previous pass may still be reading the stage buffers
    stage_store(0);
    __syncthreads();
    int buf = 0;
    for (int b0 = bb; b0 < be; b0 += kStageB) {
        const bool more = b0 + kStageB < be;
        if (more) stage_load(b0 + kStageB);  // in flight during this stage's compute
        const int nb = idle ? 0 : min(kStageB, be - b0);
        for (int s = 0; s < nb; ++s) {
            Planes16 p[RJ];
#pragma unroll
            for (int r = 0; r < RJ; ++r)
#pragma unroll
                for (int q = 0; q < NQ; ++q) p[r].set(q, P[(static_cast<size_t>(b0 + s) * 4 + q) * Gp + jl + 64 * r]);
            const uint4 *al = sm_lo + buf * kStageQ + s * RI * 4;
            const uint4 *ah = sm_hi + buf * kStageQ + s * RI * 4;
#pragma clang loop unroll(full)
            for (int i = 0; i < RI; ++i) {
                Planes16 a, c;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q < 3 && 4 * q + 1 > NB - 1) continue;  // words of planes beyond NB (word 15 = plane 0 is always read)
                    a.set(q, al[i * 4 + q]);                    // broadcast read: every lane the same address
                    if (TIES) c.set(q, ah[i * 4 + q]);
                }
                uint32_t l[4];
                if (TIES) {
                    chains_first(l, p[0].w[0], p[1].w[0], p[0].w[0], p[1].w[0], a.w[a_word(0)], a.w[a_word(0)], c.w[a_word(0)], c.w[a_word(0)]);
#pragma unroll
                    for (int k = 1; k < NB; ++k)
                        chains_next(l, p[0].w[k], p[1].w[k], p[0].w[k], p[1].w[k], a.w[a_word(k)], a.w[a_word(k)], c.w[a_word(k)], c.w[a_word(k)]);
                } else {
                    chains_first(l, p[0].w[0], p[1].w[0], p[RJ - 2].w[0], p[RJ - 1].w[0], a.w[a_word(0)], a.w[a_word(0)], a.w[a_word(0)], a.w[a_word(0)]);
#pragma unroll
                    for (int k = 1; k < NB; ++k)
                        chains_next(l, p[0].w[k], p[1].w[k], p[RJ - 2].w[k], p[RJ - 1].w[k], a.w[a_word(k)], a.w[a_word(k)], a.w[a_word(k)], a.w[a_word(k)]);
                }
                // counts of row i: low half of the packed register for even rows, high half for odd rows
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) {
                    uint32_t &dst = TIES ? (ch < 2 ? gt[ch][i >> 1] : ge[ch - 2][i >> 1]) : gt[ch % RJ][i >> 1];
                    if (i & 1) dst += static_cast<uint32_t>(__builtin_popcount(l[ch])) << 16;
                    else dst += static_cast<uint32_t>(__builtin_popcount(l[ch]));
                }
            }
        }
        if (more) stage_store(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
}

struct K1Args {
    const uint4 *P;    // pos planes (lane operand)
    const uint4 *AL;   // lo planes (tile operand)
    const uint4 *AH;   // hi planes
    uint32_t *table;
    const uint32_t *unit_map;  // work unit -> panel << 16 | i-range
    int G, Gp, Wp;
    int cb, ce, tb, te;  // ctrl / treat ranges in 32-sample blocks
    int gc, gt;          // their group ids (tie-stream key)
    int nc, nt;          // group sizes gsi1, gsi2 (:358-359)
    int m1, m2;          // threshold[1,k], threshold[2,k] (:362)
    uint64_t seed;
    int n_units, Q;      // units owned by this shard; j-chunks per panel
    const int32_t *goff; // MULTI: group offsets in blocks (ngroups + 1)
    int ngroups;
    const uint32_t *items;  // wave form: side << 31 | wave chunk << 16 | i-tile, one per workgroup
    unsigned long long *stamps;  // diagnostic (REO_K1_STAMPS=1): four s_memrealtime marks per item, else null
    const int32_t *gate;         // pipelined upload: the transform's flags (0 NaN in the input, 1 some tie so far, 4 / 5 another form of the
                                 // ranking is needed), read by k1w_pairs_gated, which picks the tie form itself -- the host need not wait
                                 // for the flags before it launches a side.  Else null.
    // RANGE ITEMS (round 6; k1w_pairs, two groups): a launch may cover a RANGE of its side's sample blocks (cb..ce / tb..te are then that
    // range), so that the pair kernel starts before a whole group has arrived (transform.hip, eager_upload).  The counts of a range wait
    // for the next one in `park` -- slot blockIdx.x (the launches of a side share ONE item list): kParkSlot words = the four genes' packed
    // 16-bit count registers lane by lane, n_gt then n_ge -- and only the launch of the side's LAST range classifies.
    uint32_t *park;              // [items][kParkSlot], or null
    int32_t *park_ge;            // [items]: the slot's n_ge half is valid (the tie form stored it; else n_ge = n_gt: no tie in those samples)
    int park_mode;               // bit 0: add the parked counts of the blocks before this range; bit 1: park the sums instead of classifying
};
constexpr size_t kParkSlot = 2 * 64 * 64;   // words per item: 64 count registers x 64 lanes, twice (32 KB)

// true when every gene j of the wave (64 RJ consecutive genes from jw) is padding (>= G) or lies in a 64-gene
// block left of the tile's block: such pairs are never emitted (emit_side)
template <int RJ>
__device__ __forceinline__ bool wave_idle(int jw, int i0, int G)
{
    return jw >= G || ((jw + 64 * RJ - 1) >> 6) < (i0 >> 6);
}

// word = 2 * word + (bit `lane` of mask): v_addc_co_u32 takes the lane mask as its per-lane carry-in
__device__ __forceinline__ void shift_in(uint32_t &w, unsigned long long mask)
{
    unsigned long long carry_out;
    asm("v_addc_co_u32 %0, %1, %0, %0, %2" : "+v"(w), "=s"(carry_out) : "s"(mask));
}

// lane `l` of v = the wave-uniform x
__device__ __forceinline__ void write_lane(uint32_t &v, uint32_t x, int l)
{
    asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(x), "i"(l));  // l must fold to a constant (unrolled loops)
}

// Epilogue of one side (control: planes 0/1, treat: planes 2/3) of a tile.  val(r, ii) is the count n of
// pair (i0+ii, jl+64r) on this side; state H <=> n >= m, L <=> size - n >= m (:376-377).  Each predicate
// is one v_cmp whose lane mask IS the forward word of row i0+ii (columns of this wave's 64-gene block);
// the mirror word of row j (:386: L and H swap) grows by one bit per row with an add-with-carry from the
// same mask.  Only pairs i < j < G are real; everything else contributes zero bits.  The diagonal 64x64
// blocks are written by several tiles and use atomicOr on the pre-zeroed table.
// NEAR: 0 no row of the tile reaches the diagonal inside this wave's 64 genes, 1 some do (d > ii per row), 2 decided at
// run time per row (one copy of the code: the workgroup kernels, whose epilogues are unrolled over their genes)
template <int RI, int NEAR, typename F>
__device__ __forceinline__ void emit_rows(bool near, int d, unsigned long long lanes_ok, int hi_thr, int lo_thr, uint32_t &wL, uint32_t &wH,
                                          uint32_t &fLlo, uint32_t &fLhi, uint32_t &fHlo, uint32_t &fHhi, F val)
{
#pragma unroll
    for (int ii = RI - 1; ii >= 0; --ii) {
        const int n = val(ii);
        const unsigned long long ok = (NEAR == 1 || (NEAR == 2 && near)) ? (__ballot(d > ii) & lanes_ok) : lanes_ok;
        const unsigned long long mH = __ballot(n >= hi_thr) & ok;
        const unsigned long long mL = __ballot(n <= lo_thr) & ok & ~mH;
        shift_in(wH, mH);
        shift_in(wL, mL);
        write_lane(fHlo, static_cast<uint32_t>(mH), ii);
        write_lane(fHhi, static_cast<uint32_t>(mH >> 32), ii);
        write_lane(fLlo, static_cast<uint32_t>(mL), ii);
        write_lane(fLhi, static_cast<uint32_t>(mL >> 32), ii);
    }
}

template <int RI, bool SPLIT, typename F>
__device__ __forceinline__ void emit_gene(const K1Args &a, int i0, int j, int bi, int lane, int pl, int hi_thr, int lo_thr, F val)
{
    const int bj = __builtin_amdgcn_readfirstlane(j >> 6);  // wave-uniform, and the compiler should know it
    if ((bj << 6) >= a.Gp || bj < bi) return;
    const int d = j - i0;                          // rows i0+ii with ii < d are above the diagonal
    const bool near = (bj << 6) - i0 < RI;         // wave-uniform: some lane has d < 32
    const unsigned long long lanes_ok = __ballot(j < a.G);
    uint32_t wL = 0, wH = 0;
    uint32_t fLlo = 0, fLhi = 0, fHlo = 0, fHhi = 0;  // lane ii: forward words of row i0+ii
    // two copies of the 32 rows, chosen once (near is wave-uniform and rare): inside a copy the rows are straight-line code,
    // so the compare -> lane mask -> writelane / add-with-carry chains of neighbouring rows overlap instead of queueing
    // behind a branch per row
    if (!SPLIT) emit_rows<RI, 2>(near, d, lanes_ok, hi_thr, lo_thr, wL, wH, fLlo, fLhi, fHlo, fHhi, val);
    else if (near) emit_rows<RI, 1>(near, d, lanes_ok, hi_thr, lo_thr, wL, wH, fLlo, fLhi, fHlo, fHhi, val);
    else emit_rows<RI, 0>(near, d, lanes_ok, hi_thr, lo_thr, wL, wH, fLlo, fLhi, fHlo, fHhi, val);
    const bool diag = (bj == bi);
    if (lane < RI && i0 + lane < a.G) {
        uint32_t *row = a.table + (static_cast<size_t>(i0 + lane) * kPlanes + pl) * a.Wp + 2 * bj;
        if (!diag) {
            *reinterpret_cast<uint2 *>(row) = uint2{fLlo, fLhi};
            *reinterpret_cast<uint2 *>(row + a.Wp) = uint2{fHlo, fHhi};
        } else {
            if (fLlo) atomicOr(row, fLlo);
            if (fLhi) atomicOr(row + 1, fLhi);
            if (fHlo) atomicOr(row + a.Wp, fHlo);
            if (fHhi) atomicOr(row + a.Wp + 1, fHhi);
        }
    }
    if (j < a.G) {  // mirror: pair (j, i) is in state 2 - state(i, j)
        uint32_t *row = a.table + (static_cast<size_t>(j) * kPlanes + pl) * a.Wp + (i0 >> 5);
        if (RI == 16) {  // a half-height item owns one 16-bit half of the word (the other half: its twin, or nobody)
            const int half = (i0 >> 4) & 1;
            if (!diag) {
                reinterpret_cast<uint16_t *>(row)[half] = static_cast<uint16_t>(wH);
                reinterpret_cast<uint16_t *>(row + a.Wp)[half] = static_cast<uint16_t>(wL);
            } else {
                if (wH) atomicOr(row, wH << (16 * half));
                if (wL) atomicOr(row + a.Wp, wL << (16 * half));
            }
        } else if (!diag) {
            row[0] = wH; row[a.Wp] = wL;
        } else {
            if (wH) atomicOr(row, wH);
            if (wL) atomicOr(row + a.Wp, wL);
        }
    }
}

template <int RI, int RJ, typename F>
__device__ __forceinline__ void emit_side(const K1Args &a, int i0, int jl, int bi, int lane, int pl, int hi_thr, int lo_thr, F val)
{
#pragma unroll
    for (int r = 0; r < RJ; ++r)
        emit_gene<RI, false>(a, i0, jl + 64 * r, bi, lane, pl, hi_thr, lo_thr, [&](int ii) { return val(r, ii); });
}

// block -> tile mapping shared by k1_pairs, k1_group_counts and k1_classify.  Work order (speed only, never
// correctness): a unit = kUnitH i-tiles x Q j-chunks.  Workgroups are dealt round-robin over the 8 XCDs, so
// workgroup b belongs to "XCD slot" b & 7; each slot walks whole units, i-tile-major inside a unit, which keeps
// the unit's pos panel (Q chunks of 256 RJ genes x nblk blocks x 64 B) in that XCD's L2 while the tile operand
// streams past once.  jl = this lane's first gene (its others are jl + 64 r): a wave owns 64 RJ consecutive genes.
template <int RI, int RJ>
__device__ __forceinline__ bool tile_of_block(const K1Args &a, int &i0, int &jl)
{
    const int slot = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int bu = kUnitH * a.Q;
    const int u = (q / bu) * 8 + slot;
    if (u >= a.n_units) return false;
    const uint32_t um = a.unit_map[u];
    const int wq = q % bu;
    const int it = static_cast<int>(um & 0xFFFFu) * kUnitH + wq / a.Q;
    const int jc = static_cast<int>(um >> 16) * a.Q + wq % a.Q;
    i0 = it * RI;
    constexpr int CJ = kTileJ * RJ;
    if (i0 >= a.Gp || jc * CJ >= a.Gp) return false;
    jl = jc * CJ + (threadIdx.x >> 6) * (64 * RJ) + (threadIdx.x & 63);
    return ((jc * CJ + CJ - 1) >> 6) >= (i0 >> 6);  // whole workgroups only: barriers inside
}

// count of pair (i0+ii, jl+64r) in group g from the packed accumulators, tie coins included (:72-77)
template <int RJ, bool TIES>
__device__ __forceinline__ int count_with_coins(const uint32_t (&gt)[RJ][kTileI / 2], const uint32_t (&ge)[TIES ? RJ : 1][kTileI / 2],
                                                uint64_t seed, int i0, int jl, int r, int ii, int g)
{
    int nre = static_cast<int>(unpack16(gt[r], ii));
    if (TIES) {
        const uint32_t neq = unpack16(ge[r], ii) - static_cast<uint32_t>(nre);
        if (neq) nre += tie_wins(seed, i0 + ii, jl + 64 * r, g, neq);
    }
    return nre;
}

// MULTI = one-vs-rest with more than two groups (:375-390) without the shared per-group counts: the treat side
// is every other group, counted group by group because the tie coins are keyed by group.
template <int NB, bool TIES, bool MULTI>
__global__ __launch_bounds__(256, MULTI ? 2 : 3) void k1_pairs(K1Args a)  // waves per SIMD wanted -> VGPR cap
{
    constexpr int RI = kTileI, RJ = TIES ? kRJTies : kRJ;
    int i0, jl;
    if (!tile_of_block<RI, RJ>(a, i0, jl)) return;
    const int lane = threadIdx.x & 63, bi = i0 >> 6;
    __shared__ uint4 sm_lo[2 * kStageB * RI * 4];
    __shared__ uint4 sm_hi[TIES ? 2 * kStageB * RI * 4 : 1];
    const bool idle = wave_idle<RJ>(jl & ~63, i0, a.G);
    uint32_t gt[RJ][RI / 2], ge[TIES ? RJ : 1][RI / 2];

    // control side (:376): nothing of it has to survive the treat-side loop
    count_pass<RJ, NB, TIES>(a.P, a.AL, a.AH, a.Gp, i0, jl, a.cb, a.ce, gt, ge, sm_lo, sm_hi, idle);
    emit_side<RI, RJ>(a, i0, jl, bi, lane, 0, a.m1, a.nc - a.m1,
                      [&](int r, int ii) { return count_with_coins<RJ, TIES>(gt, ge, a.seed, i0, jl, r, ii, a.gc); });
    // treat side (:377)
    if (!MULTI) {
        count_pass<RJ, NB, TIES>(a.P, a.AL, a.AH, a.Gp, i0, jl, a.tb, a.te, gt, ge, sm_lo, sm_hi, idle);
        emit_side<RI, RJ>(a, i0, jl, bi, lane, 2, a.m2, a.nt - a.m2,
                          [&](int r, int ii) { return count_with_coins<RJ, TIES>(gt, ge, a.seed, i0, jl, r, ii, a.gt); });
    } else {
        uint32_t tot[RJ][RI / 2];  // not = sum(nre) - nre[k]  (:374), two 16-bit sums per register (each at most S < 65536)
#pragma unroll
        for (int r = 0; r < RJ; ++r)
#pragma unroll
            for (int h = 0; h < RI / 2; ++h) tot[r][h] = 0;
        for (int g = 0; g < a.ngroups; ++g) {
            if (g == a.gc) continue;
            count_pass<RJ, NB, TIES>(a.P, a.AL, a.AH, a.Gp, i0, jl, a.goff[g], a.goff[g + 1], gt, ge, sm_lo, sm_hi, idle);
#pragma unroll
            for (int r = 0; r < RJ; ++r)
#pragma unroll
                for (int h = 0; h < RI / 2; ++h) {
                    if (TIES) {
                        const uint32_t n0 = count_with_coins<RJ, TIES>(gt, ge, a.seed, i0, jl, r, 2 * h, g);
                        const uint32_t n1 = count_with_coins<RJ, TIES>(gt, ge, a.seed, i0, jl, r, 2 * h + 1, g);
                        tot[r][h] += n0 | (n1 << 16);
                    } else tot[r][h] += gt[r][h];  // no carry between the halves
                }
        }
        emit_side<RI, RJ>(a, i0, jl, bi, lane, 2, a.m2, a.nt - a.m2, [&](int r, int ii) { return static_cast<int>(unpack16(tot[r], ii)); });
    }
}


// ---------------------------------------------------------------------------
// K1, wave form (round 3; the default for two groups): ONE WAVE PER WORKGROUP, one work item = (tile of 32 gene rows i,
// 64 RJ consecutive genes j, one side).  The count loop is one generated, hand-scheduled asm statement
// (gen_k1_loop.py -> k1_loop_gen.inc): the wave stages its own tile operand by LDS-DMA into a private 2-slot ring and
// reads it back one row ahead, so there is no barrier, no idle wave held by one, and no wait for LDS or (beyond the
// first block) for the lane operand inside the row loop.  The items are an explicit list made by the host (launch_k1:
// unit by unit, side-major, then i-tile-major, wave chunks fastest; only items that hold a real pair).  Workgroups go to
// the 8 XCDs round-robin, so every XCD gets every 8th item of that order: the same number of items each (walking whole
// units per XCD, as the workgroup form does, left the XCD with the most diagonal-free units 22 % more work than the
// average -- tools/k1w_probe.hip), and all of them inside one unit's pos panel at any time.
#include "k1_loop_gen.inc"

template <int NB>
__device__ __forceinline__ void k1_loop(u32x16 &c0, u32x16 &c1, u32x16 &c2, u32x16 &c3, const void *pb, uint32_t ps, const void *ab,
                                        uint32_t as, uint32_t nblk, uint32_t poff, uint32_t aoff, uint32_t lds)
{
    if (NB == 12) k1_loop_nb12_free(c0, c1, c2, c3, pb, ps, ab, ab, as, nblk, poff, aoff, lds);
    else if (NB == 15) k1_loop_nb15_free(c0, c1, c2, c3, pb, ps, ab, ab, as, nblk, poff, aoff, lds);
    else if (NB == 16) k1_loop_nb16_free(c0, c1, c2, c3, pb, ps, ab, ab, as, nblk, poff, aoff, lds);
    else if (NB == 17) k1_loop_nb17_free(c0, c1, c2, c3, pb, ps, ab, ab, as, nblk, poff, aoff, lds);   // (more than 65 535 genes:
    else k1_loop_nb18_free(c0, c1, c2, c3, pb, ps, ab, ab, as, nblk, poff, aoff, lds);                 //  the big plane layout)
}

// the same for a half-height item (16 gene rows, 8 packed count registers per gene: the _h loops)
template <int NB>
__device__ __forceinline__ void k1_loop(u32x8 &c0, u32x8 &c1, u32x8 &c2, u32x8 &c3, const void *pb, uint32_t ps, const void *ab,
                                        uint32_t as, uint32_t nblk, uint32_t poff, uint32_t aoff, uint32_t lds)
{
    if (NB == 12) k1_loop_nb12_free_h(c0, c1, c2, c3, pb, ps, ab, ab, as, nblk, poff, aoff, lds);
    else if (NB == 15) k1_loop_nb15_free_h(c0, c1, c2, c3, pb, ps, ab, ab, as, nblk, poff, aoff, lds);
    else if (NB == 16) k1_loop_nb16_free_h(c0, c1, c2, c3, pb, ps, ab, ab, as, nblk, poff, aoff, lds);
    else if (NB == 17) k1_loop_nb17_free_h(c0, c1, c2, c3, pb, ps, ab, ab, as, nblk, poff, aoff, lds);
    else k1_loop_nb18_free_h(c0, c1, c2, c3, pb, ps, ab, ab, as, nblk, poff, aoff, lds);
}

// One item of k1w_pairs: RI = 32 gene rows from i0, or a half-height item of 16 (the items of a launch's last, partly filled
// round are dealt as two halves each, launch_k1: the launch then ends half an item's time earlier).
template <int NB, bool TIES, int RI>
__device__ __forceinline__ void k1w_item(const K1Args &a, uint4 *ring, int i0, int jw, int side, unsigned long long t_begin)
{
    constexpr int RJ = kRJ, NE = TIES ? 2 : 1;
    constexpr bool BIG = NB > 16;
    constexpr int LQ = BIG ? 5 : 4, ROWB = BIG ? 128 : 64;  // pos quads per block; bytes of an edge row
    typedef typename std::conditional<RI == 32, u32x16, u32x8>::type Counts;
    const int lane = threadIdx.x, jl = jw + lane, bi = i0 >> 6;
    const int bb = side ? a.tb : a.cb, be = side ? a.te : a.ce;
    Counts gt0, gt1, gt2, gt3;                    // n_gt of the lane's four genes: packed, rows 2h and 2h+1
    // n_ge (tie-rich data), the first pass's counts, wait for the second pass: in the private segment (scratch) at three waves per
    // SIMD -- or, above 16 planes, where the loop's own registers leave room for two waves only, in 64 more registers.  Measured A/B
    // (profiles/r5_ties_park_ab.txt): registers everywhere cost NB = 15 its third wave, 5.1 -> 5.5 ms at config 3 and 36.9 -> 39.8 at
    // config 4; at NB = 17 (70 000 x 1 000) registers win, 67.2 -> 65.9 ms.
    constexpr bool PARK_REGS = NB > 16;
    uint32_t park[(TIES && !PARK_REGS) ? RJ * (RI / 2) : 1];
    Counts ge0 = 0, ge1 = 0, ge2 = 0, ge3 = 0;
    unsigned long long t_loop = 0, t_emit = 0;
    if (a.stamps) t_loop = __builtin_amdgcn_s_memrealtime();
    if (be > bb) {
        const char *pb = reinterpret_cast<const char *>(a.P) + static_cast<size_t>(bb) * LQ * a.Gp * 16;
        const size_t aoff = (static_cast<size_t>(bb) * a.Gp + i0) * ROWB;
        const uint32_t lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(ring));
        // ONE copy of the loop's code for both passes; nothing but the parked counts (memory) lives across the second pass
#pragma clang loop unroll(disable)
        for (int e = 0; e < NE; ++e) {
            const char *ab = reinterpret_cast<const char *>(e ? a.AL : (TIES ? a.AH : a.AL)) + aoff;  // ties: hi first, lo last
            k1_loop<NB>(gt0, gt1, gt2, gt3, pb, static_cast<uint32_t>(a.Gp) * 16u, ab, static_cast<uint32_t>(a.Gp) * static_cast<uint32_t>(ROWB),
                        static_cast<uint32_t>(be - bb), static_cast<uint32_t>(jl) * 16u, static_cast<uint32_t>(lane) * 16u, lds);
            if (TIES && e == 0) {
                if constexpr (PARK_REGS) { ge0 = gt0; ge1 = gt1; ge2 = gt2; ge3 = gt3; }
                else {
                    constexpr int W = (TIES && !PARK_REGS) ? RI / 2 : 0;
#pragma unroll
                    for (int h = 0; h < W; ++h) { park[h] = gt0[h]; park[W + h] = gt1[h]; park[2 * W + h] = gt2[h]; park[3 * W + h] = gt3[h]; }
                }
            }
        }
    } else {
        gt0 = 0; gt1 = 0; gt2 = 0; gt3 = 0;
#pragma unroll
        for (int h = 0; h < ((TIES && !PARK_REGS) ? RJ * (RI / 2) : 1); ++h) park[h] = 0;
    }
    if (a.stamps) t_emit = __builtin_amdgcn_s_memrealtime();
    if (a.park_mode) {   // (wave-uniform) range items: the counts of the side's earlier sample blocks come in, or these go out
        constexpr int W4 = RI / 8;   // uint4 per gene and lane
        uint4 *slot = reinterpret_cast<uint4 *>(a.park + static_cast<size_t>(blockIdx.x) * kParkSlot) + lane;   // [gene c][q][64 lanes]
        auto add4 = [&](Counts &v, const uint4 *p) {
#pragma unroll
            for (int q = 0; q < W4; ++q) { const uint4 w = p[q * 64]; v[4 * q] += w.x; v[4 * q + 1] += w.y; v[4 * q + 2] += w.z; v[4 * q + 3] += w.w; }   // (no carry between the halves: a count is at most S < 65 536)
        };
        auto put4 = [&](const Counts &v, uint4 *p) {
#pragma unroll
            for (int q = 0; q < W4; ++q) p[q * 64] = uint4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
        };
        if (a.park_mode & 1) {
            // n_ge of the earlier blocks: parked by the tie form, or equal to their n_gt (the tie-free form ran: no tie in those samples)
            const uint4 *ge_from = slot + ((TIES && __builtin_amdgcn_readfirstlane(a.park_ge[blockIdx.x])) ? 4 * W4 * 64 : 0);
            if constexpr (TIES && PARK_REGS) { add4(ge0, ge_from); add4(ge1, ge_from + W4 * 64); add4(ge2, ge_from + 2 * W4 * 64); add4(ge3, ge_from + 3 * W4 * 64); }
            else if constexpr (TIES) {   // (n_ge waits in the private segment: a real loop, one line at a time -- unrolled it took 64 more registers)
#pragma clang loop unroll(disable)
                for (int cq = 0; cq < 4 * W4; ++cq) { const uint4 w = ge_from[cq * 64]; park[4 * cq] += w.x; park[4 * cq + 1] += w.y; park[4 * cq + 2] += w.z; park[4 * cq + 3] += w.w; }
            }
            add4(gt0, slot); add4(gt1, slot + W4 * 64); add4(gt2, slot + 2 * W4 * 64); add4(gt3, slot + 3 * W4 * 64);
        }
        if (a.park_mode & 2) {
            put4(gt0, slot); put4(gt1, slot + W4 * 64); put4(gt2, slot + 2 * W4 * 64); put4(gt3, slot + 3 * W4 * 64);
            if constexpr (TIES && PARK_REGS) { uint4 *ge_to = slot + 4 * W4 * 64; put4(ge0, ge_to); put4(ge1, ge_to + W4 * 64); put4(ge2, ge_to + 2 * W4 * 64); put4(ge3, ge_to + 3 * W4 * 64); }
            else if constexpr (TIES) {
                uint4 *ge_to = slot + 4 * W4 * 64;
#pragma clang loop unroll(disable)
                for (int cq = 0; cq < 4 * W4; ++cq) ge_to[cq * 64] = uint4{park[4 * cq], park[4 * cq + 1], park[4 * cq + 2], park[4 * cq + 3]};
            }
            if (lane == 0) a.park_ge[blockIdx.x] = TIES ? 1 : 0;
            return;   // the side's last range classifies
        }
    }
    const int g = side ? a.gt : a.gc;
    const int m = side ? a.m2 : a.m1, n = side ? a.nt : a.nc;
    // the four genes one after the other in a real loop (the count registers rotate): a quarter of the code of the
    // unrolled form (41 KB beside a 20 KB count loop)
#pragma clang loop unroll(disable)
    for (int r = 0; r < RJ; ++r) {
        const Counts cur = gt0;
        Counts cge = 0;
        if constexpr (TIES && PARK_REGS) cge = ge0;
        else if constexpr (TIES) {
#pragma unroll
            for (int h = 0; h < RI / 2; ++h) cge[h] = park[r * (RI / 2) + h];  // (dynamic r: the array stays in memory)
        }
        emit_gene<RI, true>(a, i0, jl + 64 * r, bi, lane, side ? 2 : 0, m, n - m, [&](int ii) {
            const uint32_t w = cur[ii >> 1];
            int nre = static_cast<int>((ii & 1) ? (w >> 16) : (w & 0xFFFFu));
            if (TIES) {  // tie coins (:72-77): n = n_gt + Binomial(n_eq, 1/2) lies in [n_gt, n_ge].  The coins are drawn only
                         // where they can change the class, i.e. where that interval straddles a threshold (0.4 % of the
                         // pairs of count-like data; a wave skips the hash unless one of its 64 pairs needs it) -- any
                         // other pair is in the same class for every outcome, so the table is the oracle's bit for bit.
                const uint32_t w2 = cge[ii >> 1];
                const int nge = static_cast<int>((ii & 1) ? (w2 >> 16) : (w2 & 0xFFFFu));
                const bool amb = (nre < m && nge >= m) || (nre <= n - m && nge > n - m);
                if (amb) nre += tie_wins(a.seed, i0 + ii, jl + 64 * r, g, static_cast<uint32_t>(nge - nre));
            }
            return nre;
        });
        gt0 = gt1; gt1 = gt2; gt2 = gt3;
        if constexpr (TIES && PARK_REGS) { ge0 = ge1; ge1 = ge2; ge2 = ge3; }
    }
    if (a.stamps && lane == 0) {
        unsigned long long *st = a.stamps + static_cast<size_t>(blockIdx.x) * 4;
        st[0] = t_begin; st[1] = t_loop; st[2] = t_emit; st[3] = __builtin_amdgcn_s_memrealtime();
    }
}

// Tie-rich data (two band edges per pair): the SAME loop runs twice per item, against the lo planes (n_gt) and then
// against the hi planes (n_ge); the first pass's 64 count registers wait in the wave's private segment (16 stores and
// loads per item).  Two chains per pair inside one loop would halve the genes per lane, i.e. double the LDS reads per
// bit op -- the round-2 form, whose LDS pipe was busy 45 % of the cycles.
// More than 65 535 genes (NB = 17, 18): the big plane layout of transform.hip (five pos quads per gene and block, edge
// rows of 8 uint4), 180 registers, two waves per SIMD.
template <int NB, bool TIES>
__device__ __forceinline__ void k1w_body(const K1Args &a, uint4 *ring)
{
    constexpr int RI = kTileI, RJ = kRJ;
    const unsigned long long t_begin = a.stamps ? __builtin_amdgcn_s_memrealtime() : 0;
    // item: side << 31 | wave chunk << 16 | half-height << 15 | which half << 14 | i-tile
    const uint32_t item = a.items[blockIdx.x];
    const int jw = __builtin_amdgcn_readfirstlane(static_cast<int>((item >> 16) & 0x7FFFu) * (64 * RJ));
    const int side = __builtin_amdgcn_readfirstlane(static_cast<int>(item >> 31));
    const int tile0 = static_cast<int>(item & 0x3FFFu) * RI;
    if (__builtin_amdgcn_readfirstlane(static_cast<int>(item & 0x8000u))) {
        const int i0 = __builtin_amdgcn_readfirstlane(tile0 + ((item & 0x4000u) ? RI / 2 : 0));
        k1w_item<NB, TIES, RI / 2>(a, ring, i0, jw, side, t_begin);
    } else {
        k1w_item<NB, TIES, RI>(a, ring, __builtin_amdgcn_readfirstlane(tile0), jw, side, t_begin);
    }
}

template <int NB, bool TIES>
__global__ __launch_bounds__(64, NB > 16 ? 2 : 3) void k1w_pairs(K1Args a)
{
    constexpr int RI = kTileI;
    constexpr int ROWB = NB > 16 ? 128 : 64;
    __shared__ uint4 ring[2 * RI * ROWB / 16];  // two slots of one block's tile operand: 2 x 2 KB (4 KB)
    k1w_body<NB, TIES>(a, ring);
}

// The same with the tie form chosen on the DEVICE from the transform's flags (K1Args::gate): the pipelined upload launches a side of
// the pair kernel behind the ranking of its group without the host having read the flags (transform.hip, eager_upload).  "Some tie so
// far" covers every sample of the side; a flagged sample (a NaN, another form of the ranking needed) makes the launch
// return at once.  One path runs per launch, so the other costs nothing but its place in the code object.
template <int NB>
__global__ __launch_bounds__(64, NB > 16 ? 2 : 3) void k1w_pairs_gated(K1Args a)
{
    constexpr int RI = kTileI;
    constexpr int ROWB = NB > 16 ? 128 : 64;
    __shared__ uint4 ring[2 * RI * ROWB / 16];
    const int bad = a.gate[0] | a.gate[4] | a.gate[5], ties = a.gate[1];   // (wave-uniform: scalar loads)
    if (bad) return;
    if (ties) k1w_body<NB, true>(a, ring);
    else k1w_body<NB, false>(a, ring);
}

// ---------------------------------------------------------------------------
// Wide form of the pair loop for more than 65 535 samples (single-cell mode without pseudo-bulking, :608-616): one
// 32-bit count per register.  Tie-free: 2 genes per lane, two gene rows per step (chains (p0,a_i) (p1,a_i) (p0,a_i+1)
// (p1,a_i+1)); with ties: 1 gene per lane, chains (p0,lo_i) (p0,hi_i) (p0,lo_i+1) (p0,hi_i+1).  The tile operand is
// used by half as many chains as in the packed form, so the broadcast LDS reads bind before the VALU does; this
// path exists for completeness, not speed.
constexpr int kRJWide = 2, kRJWideTies = 1;

template <int RJ, int NB, bool TIES>
__device__ __forceinline__ void count_pass_wide(const uint4 *__restrict__ P, const uint4 *__restrict__ AL, const uint4 *__restrict__ AH,
                                                int Gp, int i0, int jl, int bb, int be, uint32_t (&gt)[RJ][kTileI],
                                                uint32_t (&ge)[TIES ? RJ : 1][kTileI], uint4 *sm_lo, uint4 *sm_hi, bool idle)
{
    static_assert((TIES && RJ == 1) || (!TIES && RJ == 2), "four chains per bit plane");
    constexpr int RI = kTileI, NQ = (NB + 3) / 4;
    constexpr int kStageQ = kStageB * RI * 4;
    constexpr int kPerThread = kStageQ / 256;
#pragma unroll
    for (int r = 0; r < RJ; ++r)
#pragma unroll
        for (int ii = 0; ii < RI; ++ii) { gt[r][ii] = 0; if (TIES) ge[r][ii] = 0; }
    if (bb >= be) return;
    uint4 sl[kPerThread], sh[TIES ? kPerThread : 1];
    auto stage_load = [&](int b0) {
#pragma unroll
        for (int e = 0; e < kPerThread; ++e) {
            const int idx = threadIdx.x + 256 * e;
            const int b = min(b0 + idx / (RI * 4), be - 1);
            const size_t o = (static_cast<size_t>(b) * Gp + i0) * 4 + idx % (RI * 4);
            sl[e] = AL[o];
            if (TIES) sh[e] = AH[o];
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int e = 0; e < kPerThread; ++e) {
            sm_lo[buf * kStageQ + threadIdx.x + 256 * e] = sl[e];
            if (TIES) sm_hi[buf * kStageQ + threadIdx.x + 256 * e] = sh[e];
        }
    };
    stage_load(bb);
    __syncthreads();
    stage_store(0);
    __syncthreads();
    int buf = 0;
    for (int b0 = bb; b0 < be; b0 += kStageB) {
        const bool more = b0 + kStageB < be;
        if (more) stage_load(b0 + kStageB);
        const int nb = idle ? 0 : min(kStageB, be - b0);
        for (int s = 0; s < nb; ++s) {
            Planes16 p[RJ];
#pragma unroll
            for (int r = 0; r < RJ; ++r)
#pragma unroll
                for (int q = 0; q < NQ; ++q) p[r].set(q, P[(static_cast<size_t>(b0 + s) * 4 + q) * Gp + jl + 64 * r]);
            const uint4 *al = sm_lo + buf * kStageQ + s * RI * 4;
            const uint4 *ah = sm_hi + buf * kStageQ + s * RI * 4;
#pragma clang loop unroll(full)
            for (int i = 0; i < RI; i += 2) {
                Planes16 a0, a1, c0, c1;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q < 3 && 4 * q + 1 > NB - 1) continue;
                    a0.set(q, al[i * 4 + q]); a1.set(q, al[(i + 1) * 4 + q]);
                    if (TIES) { c0.set(q, ah[i * 4 + q]); c1.set(q, ah[(i + 1) * 4 + q]); }
                }
                uint32_t l[4];
                if (TIES) {
                    chains_first(l, p[0].w[0], p[0].w[0], p[0].w[0], p[0].w[0], a0.w[a_word(0)], c0.w[a_word(0)], a1.w[a_word(0)], c1.w[a_word(0)]);
#pragma unroll
                    for (int k = 1; k < NB; ++k)
                        chains_next(l, p[0].w[k], p[0].w[k], p[0].w[k], p[0].w[k], a0.w[a_word(k)], c0.w[a_word(k)], a1.w[a_word(k)], c1.w[a_word(k)]);
                    gt[0][i] += __builtin_popcount(l[0]); ge[0][i] += __builtin_popcount(l[1]);
                    gt[0][i + 1] += __builtin_popcount(l[2]); ge[0][i + 1] += __builtin_popcount(l[3]);
                } else {
                    chains_first(l, p[0].w[0], p[RJ - 1].w[0], p[0].w[0], p[RJ - 1].w[0], a0.w[a_word(0)], a0.w[a_word(0)], a1.w[a_word(0)], a1.w[a_word(0)]);
#pragma unroll
                    for (int k = 1; k < NB; ++k)
                        chains_next(l, p[0].w[k], p[RJ - 1].w[k], p[0].w[k], p[RJ - 1].w[k], a0.w[a_word(k)], a0.w[a_word(k)], a1.w[a_word(k)], a1.w[a_word(k)]);
                    gt[0][i] += __builtin_popcount(l[0]); gt[RJ - 1][i] += __builtin_popcount(l[1]);
                    gt[0][i + 1] += __builtin_popcount(l[2]); gt[RJ - 1][i + 1] += __builtin_popcount(l[3]);
                }
            }
        }
        if (more) stage_store(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
}

template <int NB, bool TIES, bool MULTI>
__global__ __launch_bounds__(256, 2) void k1_pairs_wide(K1Args a)
{
    constexpr int RI = kTileI, RJ = TIES ? kRJWideTies : kRJWide;
    int i0, jl;
    if (!tile_of_block<RI, RJ>(a, i0, jl)) return;
    const int lane = threadIdx.x & 63, bi = i0 >> 6;
    __shared__ uint4 sm_lo[2 * kStageB * RI * 4];
    __shared__ uint4 sm_hi[TIES ? 2 * kStageB * RI * 4 : 1];
    const bool idle = wave_idle<RJ>(jl & ~63, i0, a.G);
    uint32_t gt[RJ][RI], ge[TIES ? RJ : 1][RI];
    auto count_of = [&](int r, int ii, int g) -> int {
        int nre = static_cast<int>(gt[r][ii]);
        if (TIES) {
            const uint32_t neq = ge[r][ii] - gt[r][ii];
            if (neq) nre += tie_wins(a.seed, i0 + ii, jl + 64 * r, g, neq);
        }
        return nre;
    };
    count_pass_wide<RJ, NB, TIES>(a.P, a.AL, a.AH, a.Gp, i0, jl, a.cb, a.ce, gt, ge, sm_lo, sm_hi, idle);
    emit_side<RI, RJ>(a, i0, jl, bi, lane, 0, a.m1, a.nc - a.m1, [&](int r, int ii) { return count_of(r, ii, a.gc); });
    if (!MULTI) {
        count_pass_wide<RJ, NB, TIES>(a.P, a.AL, a.AH, a.Gp, i0, jl, a.tb, a.te, gt, ge, sm_lo, sm_hi, idle);
        emit_side<RI, RJ>(a, i0, jl, bi, lane, 2, a.m2, a.nt - a.m2, [&](int r, int ii) { return count_of(r, ii, a.gt); });
    } else {
        int tot[RJ][RI];  // not = sum(nre) - nre[k]  (:374)
#pragma unroll
        for (int r = 0; r < RJ; ++r)
#pragma unroll
            for (int ii = 0; ii < RI; ++ii) tot[r][ii] = 0;
        for (int g = 0; g < a.ngroups; ++g) {
            if (g == a.gc) continue;
            count_pass_wide<RJ, NB, TIES>(a.P, a.AL, a.AH, a.Gp, i0, jl, a.goff[g], a.goff[g + 1], gt, ge, sm_lo, sm_hi, idle);
#pragma unroll
            for (int r = 0; r < RJ; ++r)
#pragma unroll
                for (int ii = 0; ii < RI; ++ii) tot[r][ii] += count_of(r, ii, g);
        }
        emit_side<RI, RJ>(a, i0, jl, bi, lane, 2, a.m2, a.nt - a.m2, [&](int r, int ii) { return tot[r][ii]; });
    }
}

// ---------------------------------------------------------------------------
// One-vs-rest with C > 2 groups (:375-390,396-436): the C comparisons need the same per-group counts
// nre_g(i,j) (tie coins are keyed by group, not by comparison), so they are counted once, kept in
// HBM, and each comparison only classifies:  c-side = nre_k,  t-side = sum_g nre_g - nre_k  (:374).
// Layout: plane g (g = C holds the sum) = [Gp/32 i-tiles][4 quarters][Gp genes j][8] u16, element
// (it, q, j, e) = pair (32 it + 8 q + e, j): one 16-byte load or store per lane, coalesced over j.
__device__ __forceinline__ size_t gc_index(int it, int q, int j, int Gp) { return (static_cast<size_t>(it * 4 + q) * Gp + j) * 8; }

template <int NB, bool TIES>
__global__ __launch_bounds__(256, 2) void k1_group_counts(K1Args a, uint16_t *__restrict__ planes, size_t plane_elems)
{
    constexpr int RI = kTileI, RJ = TIES ? kRJTies : kRJ;
    int i0, jl;
    if (!tile_of_block<RI, RJ>(a, i0, jl)) return;
    __shared__ uint4 sm_lo[2 * kStageB * RI * 4];
    __shared__ uint4 sm_hi[TIES ? 2 * kStageB * RI * 4 : 1];
    const bool idle = wave_idle<RJ>(jl & ~63, i0, a.G);
    uint32_t gt[RJ][RI / 2], ge[TIES ? RJ : 1][RI / 2];
    uint32_t tot[RJ][RI / 2];  // two u16 sums per register (sums are at most S < 65536)
#pragma unroll
    for (int r = 0; r < RJ; ++r)
#pragma unroll
        for (int h = 0; h < RI / 2; ++h) tot[r][h] = 0;
    const int it = i0 / RI;
    for (int g = 0; g < a.ngroups; ++g) {
        count_pass<RJ, NB, TIES>(a.P, a.AL, a.AH, a.Gp, i0, jl, a.goff[g], a.goff[g + 1], gt, ge, sm_lo, sm_hi, idle);
        uint16_t *plane = planes + static_cast<size_t>(g) * plane_elems;
#pragma unroll
        for (int r = 0; r < RJ; ++r) {
            const int j = jl + 64 * r;
            uint32_t pk[RI / 2];
#pragma unroll
            for (int h = 0; h < RI / 2; ++h) {
                if (TIES) {
                    const uint32_t n0 = count_with_coins<RJ, TIES>(gt, ge, a.seed, i0, jl, r, 2 * h, g);
                    const uint32_t n1 = count_with_coins<RJ, TIES>(gt, ge, a.seed, i0, jl, r, 2 * h + 1, g);
                    pk[h] = n0 | (n1 << 16);
                } else pk[h] = gt[r][h];
                tot[r][h] += pk[h];  // no carry between the halves: each half-sum stays below 2^16
            }
            if (j < a.Gp) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<uint4 *>(plane + gc_index(it, q, j, a.Gp)) = uint4{pk[4 * q], pk[4 * q + 1], pk[4 * q + 2], pk[4 * q + 3]};
            }
        }
    }
    uint16_t *plane = planes + static_cast<size_t>(a.ngroups) * plane_elems;
#pragma unroll
    for (int r = 0; r < RJ; ++r) {
        const int j = jl + 64 * r;
        if (j < a.Gp) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<uint4 *>(plane + gc_index(it, q, j, a.Gp)) = uint4{tot[r][4 * q], tot[r][4 * q + 1], tot[r][4 * q + 2], tot[r][4 * q + 3]};
        }
    }
}

// Classify comparison k from the stored counts.  HBM-bound: 2 x 64 B read per (tile, gene j),
// the class-table words written as in k1_pairs.
template <int RJ>
__global__ __launch_bounds__(256) void k1_classify(K1Args a, const uint16_t *__restrict__ planes, size_t plane_elems)
{
    constexpr int RI = kTileI;
    int i0, jl;
    if (!tile_of_block<RI, RJ>(a, i0, jl)) return;
    const int lane = threadIdx.x & 63;
    const int it = i0 / RI;
    const uint16_t *pk = planes + static_cast<size_t>(a.gc) * plane_elems;
    const uint16_t *pt = planes + static_cast<size_t>(a.ngroups) * plane_elems;
    uint32_t wk[RJ][RI / 2], wt[RJ][RI / 2];  // two u16 counts per register
#pragma unroll
    for (int r = 0; r < RJ; ++r) {
        const int j = jl + 64 * r;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint4 vk = {0, 0, 0, 0}, vt = {0, 0, 0, 0};
            if (j < a.Gp) {
                vk = *reinterpret_cast<const uint4 *>(pk + gc_index(it, q, j, a.Gp));
                vt = *reinterpret_cast<const uint4 *>(pt + gc_index(it, q, j, a.Gp));
            }
            wk[r][4 * q] = vk.x; wk[r][4 * q + 1] = vk.y; wk[r][4 * q + 2] = vk.z; wk[r][4 * q + 3] = vk.w;
            wt[r][4 * q] = vt.x; wt[r][4 * q + 1] = vt.y; wt[r][4 * q + 2] = vt.z; wt[r][4 * q + 3] = vt.w;
        }
    }
    auto nk_of = [&](int r, int ii) -> int { return static_cast<int>(unpack16(wk[r], ii)); };
    auto nt_of = [&](int r, int ii) -> int { return static_cast<int>(unpack16(wt[r], ii)) - nk_of(r, ii); };
    emit_side<RI, RJ>(a, i0, jl, i0 >> 6, lane, 0, a.m1, a.nc - a.m1, nk_of);
    emit_side<RI, RJ>(a, i0, jl, i0 >> 6, lane, 2, a.m2, a.nt - a.m2, nt_of);
}

// More than 65 535 samples (single-cell mode without pseudo-bulking, :608-616), wave form: a side's blocks in runs of at most
// 2 047 (65 504 samples: the packed 16-bit counts of the loop cannot overflow inside a run), the runs' counts added into
// 32-bit totals that wait in the private segment; classification from the totals.  Two groups.
constexpr int kWideRun = 2047;  // 32-sample blocks per run of the count loop

template <int NB, bool TIES>
__global__ __launch_bounds__(64, NB > 16 ? 2 : 3) void k1w_pairs_wide(K1Args a)
{
    constexpr int RI = kTileI, RJ = kRJ, NE = TIES ? 2 : 1;
    constexpr bool BIG = NB > 16;
    constexpr int LQ = BIG ? 5 : 4, ROWB = BIG ? 128 : 64;
    __shared__ uint4 ring[2 * RI * ROWB / 16];
    const uint32_t item = a.items[blockIdx.x];  // side << 31 | wave chunk << 16 | i-tile
    const int i0 = __builtin_amdgcn_readfirstlane(static_cast<int>(item & 0xFFFFu) * RI);
    const int jw = __builtin_amdgcn_readfirstlane(static_cast<int>((item >> 16) & 0x7FFFu) * (64 * RJ));
    const int side = __builtin_amdgcn_readfirstlane(static_cast<int>(item >> 31));
    const int lane = threadIdx.x, jl = jw + lane, bi = i0 >> 6;
    const int bb = side ? a.tb : a.cb, be = side ? a.te : a.ce;
    const uint32_t lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(&ring[0]));
    uint32_t tgt[RJ * RI], tge[TIES ? RJ * RI : 1];   // 32-bit totals of n_gt (n_ge); indexed by loop counters: memory
#pragma unroll
    for (int h = 0; h < RJ * RI; ++h) { tgt[h] = 0; if (TIES) tge[h] = 0; }
#pragma clang loop unroll(disable)
    for (int c0 = bb; c0 < be; c0 += kWideRun) {
        const int nrun = min(kWideRun, be - c0);
        const char *pb = reinterpret_cast<const char *>(a.P) + static_cast<size_t>(c0) * LQ * a.Gp * 16;
        const size_t aoff = (static_cast<size_t>(c0) * a.Gp + i0) * ROWB;
#pragma clang loop unroll(disable)
        for (int e = 0; e < NE; ++e) {
            const char *ab = reinterpret_cast<const char *>(e ? a.AL : (TIES ? a.AH : a.AL)) + aoff;  // ties: hi (n_ge) first, lo (n_gt) last
            u32x16 c[4];
            k1_loop<NB>(c[0], c[1], c[2], c[3], pb, static_cast<uint32_t>(a.Gp) * 16u, ab, static_cast<uint32_t>(a.Gp) * static_cast<uint32_t>(ROWB),
                        static_cast<uint32_t>(nrun), static_cast<uint32_t>(jl) * 16u, static_cast<uint32_t>(lane) * 16u, lds);
            uint32_t *dst = (TIES && e == 0) ? tge : tgt;
#pragma unroll
            for (int r = 0; r < RJ; ++r)
#pragma unroll
                for (int h = 0; h < RI / 2; ++h) {
                    dst[r * RI + 2 * h] += c[r][h] & 0xFFFFu;
                    dst[r * RI + 2 * h + 1] += c[r][h] >> 16;
                }
        }
    }
    const int g = side ? a.gt : a.gc;
    const int m = side ? a.m2 : a.m1, n = side ? a.nt : a.nc;
#pragma clang loop unroll(disable)
    for (int r = 0; r < RJ; ++r) {
        uint32_t cg[RI], ce[TIES ? RI : 1];
#pragma unroll
        for (int ii = 0; ii < RI; ++ii) { cg[ii] = tgt[r * RI + ii]; if (TIES) ce[ii] = tge[r * RI + ii]; }
        emit_gene<RI, true>(a, i0, jl + 64 * r, bi, lane, side ? 2 : 0, m, n - m, [&](int ii) {
            int nre = static_cast<int>(cg[ii]);
            if (TIES) {  // coins only where they can change the class (as in k1w_pairs)
                const int nge = static_cast<int>(ce[ii]);
                const bool amb = (nre < m && nge >= m) || (nre <= n - m && nge > n - m);
                if (amb) nre += tie_wins(a.seed, i0 + ii, jl + 64 * r, g, static_cast<uint32_t>(nge - nre));
            }
            return nre;
        });
    }
}

// The per-group counts by the wave form (round 3): one wave per workgroup, one item = (tile of 32 rows, 256 genes), the
// generated count loop of k1w_pairs run once per group (twice with ties: hi planes, then lo planes), the counts of group
// g written to plane g as they come and summed into the last plane.  Stored counts carry their tie coins (every pair:
// the classification of a comparison adds counts of several groups, so no coin can be skipped here).
template <int NB, bool TIES>
__global__ __launch_bounds__(64, NB > 16 ? 2 : 3) void k1w_group_counts(K1Args a, uint16_t *__restrict__ planes, size_t plane_elems)
{
    constexpr int RI = kTileI, RJ = kRJ, NE = TIES ? 2 : 1;
    constexpr bool BIG = NB > 16;
    constexpr int LQ = BIG ? 5 : 4, ROWB = BIG ? 128 : 64;
    __shared__ uint4 ring[2 * RI * ROWB / 16];
    const uint32_t item = a.items[blockIdx.x];  // wave chunk << 16 | i-tile
    const int i0 = __builtin_amdgcn_readfirstlane(static_cast<int>(item & 0xFFFFu) * RI);
    const int jw = __builtin_amdgcn_readfirstlane(static_cast<int>((item >> 16) & 0x7FFFu) * (64 * RJ));
    const int lane = threadIdx.x, jl = jw + lane, it = i0 / RI;
    const uint32_t lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(&ring[0]));
    uint32_t tot[RJ * (RI / 2)];   // sums over the groups: two u16 per word (a sum is at most S < 65536); indexed by a loop counter: memory
    uint32_t park[TIES ? RJ * (RI / 2) : 1];
#pragma unroll
    for (int h = 0; h < RJ * (RI / 2); ++h) tot[h] = 0;
#pragma clang loop unroll(disable)
    for (int g = 0; g < a.ngroups; ++g) {
        const int bb = __builtin_amdgcn_readfirstlane(a.goff[g]), be = __builtin_amdgcn_readfirstlane(a.goff[g + 1]);
        u32x16 gt0 = 0, gt1 = 0, gt2 = 0, gt3 = 0;
        if (be > bb) {
            const char *pb = reinterpret_cast<const char *>(a.P) + static_cast<size_t>(bb) * LQ * a.Gp * 16;
            const size_t aoff = (static_cast<size_t>(bb) * a.Gp + i0) * ROWB;
#pragma clang loop unroll(disable)
            for (int e = 0; e < NE; ++e) {
                const char *ab = reinterpret_cast<const char *>(e ? a.AL : (TIES ? a.AH : a.AL)) + aoff;
                k1_loop<NB>(gt0, gt1, gt2, gt3, pb, static_cast<uint32_t>(a.Gp) * 16u, ab, static_cast<uint32_t>(a.Gp) * static_cast<uint32_t>(ROWB),
                            static_cast<uint32_t>(be - bb), static_cast<uint32_t>(jl) * 16u, static_cast<uint32_t>(lane) * 16u, lds);
                if (TIES && e == 0) {
#pragma unroll
                    for (int h = 0; h < RI / 2; ++h) {
                        park[h] = gt0[h]; park[(TIES ? 1 : 0) * (RI / 2) + h] = gt1[h];
                        park[(TIES ? 2 : 0) * (RI / 2) + h] = gt2[h]; park[(TIES ? 3 : 0) * (RI / 2) + h] = gt3[h];
                    }
                }
            }
        } else if (TIES) {
#pragma unroll
            for (int h = 0; h < RJ * (RI / 2); ++h) park[h] = 0;
        }
        uint16_t *plane = planes + static_cast<size_t>(g) * plane_elems;
#pragma clang loop unroll(disable)
        for (int r = 0; r < RJ; ++r) {
            const int j = jl + 64 * r;
            uint32_t pk[RI / 2];
#pragma unroll
            for (int h = 0; h < RI / 2; ++h) {
                uint32_t w = gt0[h];
                if (TIES) {  // n = n_gt + Binomial(n_eq, 1/2), :72-77
                    const uint32_t w2 = park[r * (RI / 2) + h];
                    uint32_t n0 = w & 0xFFFFu, n1 = w >> 16;
                    const uint32_t e0 = (w2 & 0xFFFFu) - n0, e1 = (w2 >> 16) - n1;
                    if (e0) n0 += tie_wins(a.seed, i0 + 2 * h, j, g, e0);
                    if (e1) n1 += tie_wins(a.seed, i0 + 2 * h + 1, j, g, e1);
                    w = n0 | (n1 << 16);
                }
                pk[h] = w;
                tot[r * (RI / 2) + h] += w;  // no carry between the halves: each half-sum stays below 2^16
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<uint4 *>(plane + gc_index(it, q, j, a.Gp)) = uint4{pk[4 * q], pk[4 * q + 1], pk[4 * q + 2], pk[4 * q + 3]};
            gt0 = gt1; gt1 = gt2; gt2 = gt3;
        }
    }
    uint16_t *plane = planes + static_cast<size_t>(a.ngroups) * plane_elems;
#pragma clang loop unroll(disable)
    for (int r = 0; r < RJ; ++r) {
        const int j = jl + 64 * r;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<uint4 *>(plane + gc_index(it, q, j, a.Gp)) =
                uint4{tot[r * (RI / 2) + 4 * q], tot[r * (RI / 2) + 4 * q + 1], tot[r * (RI / 2) + 4 * q + 2], tot[r * (RI / 2) + 4 * q + 3]};
    }
}

// One comparison of a one-vs-rest run WITHOUT the shared count planes (they did not fit, or REO_SHARE_GROUP_COUNTS=0), wave
// form: one item = (tile, chunk), the generated loop once per group; the counts of group k stay apart, those of every
// other group are summed (not = sum(nre) - nre[k], :374), and both sides are classified at the end (:376-377).
template <int NB, bool TIES>
__global__ __launch_bounds__(64, NB > 16 ? 2 : 3) void k1w_pairs_multi(K1Args a)
{
    constexpr int RI = kTileI, RJ = kRJ, NE = TIES ? 2 : 1;
    constexpr bool BIG = NB > 16;
    constexpr int LQ = BIG ? 5 : 4, ROWB = BIG ? 128 : 64;
    __shared__ uint4 ring[2 * RI * ROWB / 16];
    const uint32_t item = a.items[blockIdx.x];  // wave chunk << 16 | i-tile
    const int i0 = __builtin_amdgcn_readfirstlane(static_cast<int>(item & 0xFFFFu) * RI);
    const int jw = __builtin_amdgcn_readfirstlane(static_cast<int>((item >> 16) & 0x7FFFu) * (64 * RJ));
    const int lane = threadIdx.x, jl = jw + lane, bi = i0 >> 6;
    const uint32_t lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(&ring[0]));
    uint32_t nk[RJ * (RI / 2)], tot[RJ * (RI / 2)];   // packed u16 pairs; indexed by loop counters: memory
    uint32_t park[TIES ? RJ * (RI / 2) : 1];
#pragma unroll
    for (int h = 0; h < RJ * (RI / 2); ++h) { nk[h] = 0; tot[h] = 0; }
#pragma clang loop unroll(disable)
    for (int g = 0; g < a.ngroups; ++g) {
        const int bb = __builtin_amdgcn_readfirstlane(a.goff[g]), be = __builtin_amdgcn_readfirstlane(a.goff[g + 1]);
        if (be <= bb) continue;
        u32x16 gt0, gt1, gt2, gt3;
        const char *pb = reinterpret_cast<const char *>(a.P) + static_cast<size_t>(bb) * LQ * a.Gp * 16;
        const size_t aoff = (static_cast<size_t>(bb) * a.Gp + i0) * ROWB;
#pragma clang loop unroll(disable)
        for (int e = 0; e < NE; ++e) {
            const char *ab = reinterpret_cast<const char *>(e ? a.AL : (TIES ? a.AH : a.AL)) + aoff;
            k1_loop<NB>(gt0, gt1, gt2, gt3, pb, static_cast<uint32_t>(a.Gp) * 16u, ab, static_cast<uint32_t>(a.Gp) * static_cast<uint32_t>(ROWB),
                        static_cast<uint32_t>(be - bb), static_cast<uint32_t>(jl) * 16u, static_cast<uint32_t>(lane) * 16u, lds);
            if (TIES && e == 0) {
#pragma unroll
                for (int h = 0; h < RI / 2; ++h) {
                    park[h] = gt0[h]; park[(TIES ? 1 : 0) * (RI / 2) + h] = gt1[h];
                    park[(TIES ? 2 : 0) * (RI / 2) + h] = gt2[h]; park[(TIES ? 3 : 0) * (RI / 2) + h] = gt3[h];
                }
            }
        }
#pragma clang loop unroll(disable)
        for (int r = 0; r < RJ; ++r) {
            const int j = jl + 64 * r;
#pragma unroll
            for (int h = 0; h < RI / 2; ++h) {
                uint32_t w = gt0[h];
                if (TIES) {  // the coins are keyed by group (:72-77): every group's count carries its own
                    const uint32_t w2 = park[r * (RI / 2) + h];
                    uint32_t n0 = w & 0xFFFFu, n1 = w >> 16;
                    const uint32_t e0 = (w2 & 0xFFFFu) - n0, e1 = (w2 >> 16) - n1;
                    if (e0) n0 += tie_wins(a.seed, i0 + 2 * h, j, g, e0);
                    if (e1) n1 += tie_wins(a.seed, i0 + 2 * h + 1, j, g, e1);
                    w = n0 | (n1 << 16);
                }
                if (g == a.gc) nk[r * (RI / 2) + h] = w; else tot[r * (RI / 2) + h] += w;  // (no carry between the halves: sums stay below 2^16)
            }
            gt0 = gt1; gt1 = gt2; gt2 = gt3;
        }
    }
#pragma clang loop unroll(disable)
    for (int r = 0; r < RJ; ++r) {
        uint32_t ck[RI / 2], ct[RI / 2];
#pragma unroll
        for (int h = 0; h < RI / 2; ++h) { ck[h] = nk[r * (RI / 2) + h]; ct[h] = tot[r * (RI / 2) + h]; }
        emit_gene<RI, true>(a, i0, jl + 64 * r, bi, lane, 0, a.m1, a.nc - a.m1, [&](int ii) { return static_cast<int>((ii & 1) ? (ck[ii >> 1] >> 16) : (ck[ii >> 1] & 0xFFFFu)); });
        emit_gene<RI, true>(a, i0, jl + 64 * r, bi, lane, 2, a.m2, a.nt - a.m2, [&](int ii) { return static_cast<int>((ii & 1) ? (ct[ii >> 1] >> 16) : (ct[ii >> 1] & 0xFFFFu)); });
    }
}

// Parity hook: the raw counts of a block of ordered pairs, one thread per pair, the same borrow chain over the
// same planes as count_pass (plain loads, no staging: blocks of a few hundred genes).
__global__ __launch_bounds__(256) void k1_counts(const uint4 *__restrict__ P, const uint4 *__restrict__ AL,
                                                 const uint4 *__restrict__ AH, int Gp, int nbits,
                                                 const int32_t *__restrict__ goff, int ngroups, int ci0, int ci1, int cj0, int cj1,
                                                 uint16_t *__restrict__ out_gt, uint16_t *__restrict__ out_eq)
{
    const int j = cj0 + blockIdx.x * 256 + threadIdx.x, i = ci0 + blockIdx.y;
    if (j >= cj1 || i >= ci1) return;
    const int nj = cj1 - cj0;
    for (int g = 0; g < ngroups; ++g) {
        uint32_t n_gt = 0, n_ge = 0;
        for (int b = goff[g]; b < goff[g + 1]; ++b) {
            Planes16 p, lo, hi;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                p.set(q, P[(static_cast<size_t>(b) * 4 + q) * Gp + j]);
                lo.set(q, AL[(static_cast<size_t>(b) * Gp + i) * 4 + q]);
                hi.set(q, AH[(static_cast<size_t>(b) * Gp + i) * 4 + q]);
            }
            uint32_t lt = 0, le = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (k >= nbits) break;
                lt = __builtin_amdgcn_bitop3_b32(p.w[k], lo.w[a_word(k)], lt, 0x8e);
                le = __builtin_amdgcn_bitop3_b32(p.w[k], hi.w[a_word(k)], le, 0x8e);
            }
            n_gt += __builtin_popcount(lt);
            n_ge += __builtin_popcount(le);
        }
        const size_t o = (static_cast<size_t>(i - ci0) * nj + (j - cj0)) * ngroups + g;
        out_gt[o] = static_cast<uint16_t>(n_gt);
        out_eq[o] = static_cast<uint16_t>(n_ge - n_gt);
    }
}

// The same on the big plane layout (more than 65 535 genes: five pos quads, edge rows of 8 uint4, plane k in word k).
__global__ __launch_bounds__(256) void k1_counts_big(const uint4 *__restrict__ P, const uint4 *__restrict__ AL,
                                                     const uint4 *__restrict__ AH, int Gp, int nbits,
                                                     const int32_t *__restrict__ goff, int ngroups, int ci0, int ci1, int cj0, int cj1,
                                                     uint16_t *__restrict__ out_gt, uint16_t *__restrict__ out_eq)
{
    const int j = cj0 + blockIdx.x * 256 + threadIdx.x, i = ci0 + blockIdx.y;
    if (j >= cj1 || i >= ci1) return;
    const int nj = cj1 - cj0;
    for (int g = 0; g < ngroups; ++g) {
        uint32_t n_gt = 0, n_ge = 0;
        for (int b = goff[g]; b < goff[g + 1]; ++b) {
            uint32_t p[20], lo[20], hi[20];
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const uint4 pv = P[(static_cast<size_t>(b) * 5 + q) * Gp + j];
                const uint4 lv = AL[(static_cast<size_t>(b) * Gp + i) * 8 + q], hv = AH[(static_cast<size_t>(b) * Gp + i) * 8 + q];
                p[4 * q] = pv.x; p[4 * q + 1] = pv.y; p[4 * q + 2] = pv.z; p[4 * q + 3] = pv.w;
                lo[4 * q] = lv.x; lo[4 * q + 1] = lv.y; lo[4 * q + 2] = lv.z; lo[4 * q + 3] = lv.w;
                hi[4 * q] = hv.x; hi[4 * q + 1] = hv.y; hi[4 * q + 2] = hv.z; hi[4 * q + 3] = hv.w;
            }
            uint32_t lt = 0, le = 0;
#pragma unroll
            for (int k = 0; k < 20; ++k) {
                if (k >= nbits) break;
                lt = __builtin_amdgcn_bitop3_b32(p[k], lo[k], lt, 0x8e);
                le = __builtin_amdgcn_bitop3_b32(p[k], hi[k], le, 0x8e);
            }
            n_gt += __builtin_popcount(lt);
            n_ge += __builtin_popcount(le);
        }
        const size_t o = (static_cast<size_t>(i - ci0) * nj + (j - cj0)) * ngroups + g;
        out_gt[o] = static_cast<uint16_t>(n_gt);
        out_eq[o] = static_cast<uint16_t>(n_ge - n_gt);
    }
}

// Parity hook: decode the bit planes of a block into class codes 0..8.
__global__ void k_decode(const uint32_t *__restrict__ table, int Wp, int i0, int i1, int j0, int j1,
                         uint8_t *__restrict__ code)
{
    const int j = j0 + blockIdx.x * blockDim.x + threadIdx.x;
    const int i = i0 + blockIdx.y;
    if (j >= j1 || i >= i1) return;
    const uint32_t *row = table + static_cast<size_t>(i) * kPlanes * Wp + (j >> 5);
    const uint32_t sh = j & 31;
    const int l = (row[0] >> sh) & 1, h = (row[Wp] >> sh) & 1;
    const int tl = (row[2 * Wp] >> sh) & 1, th = (row[3 * Wp] >> sh) & 1;
    const int ic = l ? 0 : (h ? 2 : 1), it = tl ? 0 : (th ? 2 : 1);
    code[static_cast<size_t>(i - i0) * (j1 - j0) + (j - j0)] = (i == j) ? 255 : static_cast<uint8_t>(3 * ic + it);
}

// Consistency of a class table that came through an exchange: a pair is in at most one of the states L / H on each side,
// nothing sits on the diagonal or past the last gene.  One wave per row; flag[0] |= 1 on a violation.  (A caller-supplied
// collective that delivers wrong words would otherwise hand the iteration passes tallies that break their invariants.)
// diagnostic (REO_K1_STAMPS): one s_memrealtime mark on the stream's timeline, before and after the pair kernel
__global__ void k_time_mark(unsigned long long *out) { if (threadIdx.x == 0) *out = __builtin_amdgcn_s_memrealtime(); }

__global__ __launch_bounds__(256) void k_check_table(const uint32_t *__restrict__ table, int G, int Wp, int32_t *__restrict__ flag)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= G) return;
    const uint32_t *r = table + static_cast<size_t>(row) * kPlanes * Wp;
    uint32_t bad = 0;
    for (int w = lane; w < Wp; w += 64) {
        const uint32_t cl = r[w], ch = r[Wp + w], tl = r[2 * Wp + w], th = r[3 * Wp + w];
        uint32_t valid = 0xFFFFFFFFu;                                   // columns < G, not the diagonal
        const int c0 = w * 32;
        if (c0 + 32 > G) valid = c0 >= G ? 0u : (0xFFFFFFFFu >> (c0 + 32 - G));
        if ((row >> 5) == w) valid &= ~(1u << (row & 31));
        bad |= (cl & ch) | (tl & th) | ((cl | ch | tl | th) & ~valid);
    }
    if (__ballot(bad != 0) && lane == 0) atomicOr(flag, 1);
}

// bytes -> bit mask (one workgroup-wide ballot per 64 genes)
__global__ __launch_bounds__(256) void k_pack_ref(const uint8_t *__restrict__ bytes, int Gp,
                                                  uint32_t *__restrict__ bits)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    const unsigned long long m = __ballot(g < Gp && bytes[g] != 0);
    if ((threadIdx.x & 63) == 0) {
        bits[g >> 5] = static_cast<uint32_t>(m);
        bits[(g >> 5) + 1] = static_cast<uint32_t>(m >> 32);
    }
}

// Start of reo_identify_degs in ONE launch (it was a memset, two copies, k_pack_ref and three more memsets: seven stream
// operations of 3 - 10 us each in front of the first pass): the caller's reference mask and the initial loop state are
// read from pinned host memory, the mask goes to device memory as bytes (padded with zeros to Gp) and as bits, and the
// result matrix, the K2 mode log and the BH-rank histograms are cleared.
struct IterInitArgs {
    const uint8_t *host_ref;    // pinned, G bytes
    const uint32_t *host_state; // pinned, IterState
    uint8_t *refbytes;
    uint32_t *refbits, *state;
    uint32_t *zero[3];
    size_t zero_n[3];           // in 32-bit words
    int G, Gp, state_words;
};
__global__ __launch_bounds__(256) void k_iter_init(IterInitArgs a)
{
    const int g = blockIdx.x * 256 + threadIdx.x;  // grid = Gp / 256
    const uint8_t b = g < a.G ? a.host_ref[g] : 0;
    a.refbytes[g] = b;
    const unsigned long long m = __ballot(b != 0);
    if ((threadIdx.x & 63) == 0) {
        a.refbits[g >> 5] = static_cast<uint32_t>(m);
        a.refbits[(g >> 5) + 1] = static_cast<uint32_t>(m >> 32);
    }
    if (blockIdx.x == 0 && static_cast<int>(threadIdx.x) < a.state_words) a.state[threadIdx.x] = a.host_state[threadIdx.x];
    const size_t nthreads = static_cast<size_t>(gridDim.x) * 256;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        for (size_t i = static_cast<size_t>(g); i < a.zero_n[k]; i += nthreads) a.zero[k][i] = 0;
}

// ---------------------------------------------------------------------------
// K2: one wave per gene row.  Streams the row's four planes (16 B per lane per
// load, fully coalesced), ANDs with the reference mask and popcounts.  Raw
// counters: 0 cL, 1 cH, 2 tL, 3 tH (marginals), 4 LL, 5 LH, 6 HL, 7 HH.  The 9 tallies
// follow from them in k3_derive.  HBM-bound: 16 VALU ops per 32 pairs.
__device__ __forceinline__ void tally_word(uint32_t cl, uint32_t ch, uint32_t tl, uint32_t th, uint32_t m,
                                           uint32_t (&c)[kRaw])
{
    cl &= m; ch &= m; tl &= m; th &= m;
    c[0] += __popc(cl); c[1] += __popc(ch); c[2] += __popc(tl); c[3] += __popc(th);
    c[4] += __popc(cl & tl); c[5] += __popc(cl & th); c[6] += __popc(ch & tl); c[7] += __popc(ch & th);
}

__device__ __forceinline__ void tally_rows(const uint4 *__restrict__ table, const uint4 *__restrict__ refbits, int G, int Wq,
                                           int32_t *__restrict__ raw)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= G) return;
    const uint4 *r = table + static_cast<size_t>(row) * kPlanes * Wq;
    uint32_t c[kRaw] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int q = lane; q < Wq; q += 64) {
        const uint4 m = refbits[q];
        const uint4 cl = r[q], ch = r[Wq + q], tl = r[2 * Wq + q], th = r[3 * Wq + q];
        tally_word(cl.x, ch.x, tl.x, th.x, m.x, c);
        tally_word(cl.y, ch.y, tl.y, th.y, m.y, c);
        tally_word(cl.z, ch.z, tl.z, th.z, m.z, c);
        tally_word(cl.w, ch.w, tl.w, th.w, m.w, c);
    }
#pragma unroll
    for (int t = 0; t < kRaw; ++t) {
        uint32_t v = c[t];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        c[t] = v;
    }
    if (lane == 0) {
        int4 *o = reinterpret_cast<int4 *>(raw + static_cast<size_t>(row) * kRaw);
        o[0] = make_int4(c[0], c[1], c[2], c[3]);
        o[1] = make_int4(c[4], c[5], c[6], c[7]);
    }
}

// Incremental form of K2.  From one pass to the next the reference set usually changes by a handful
// of genes (the k3_finalize of the previous pass lists them).  The counters are linear in the mask, and
// by the mirror rule (:386) column j of the table is row j with L and H swapped, so gene j entering
// (leaving) the reference set adds (removes), for every gene i, the class bits found at bit i of ROW j:
// one contiguous row per changed gene instead of the whole table.  Exact (integer sums).
template <bool COH, int U = 4>
__device__ __forceinline__ void delta_counts(const uint32_t *__restrict__ table, int Wp, const uint32_t *list, int n, int i, int (&d)[kRaw])
{
    const int w = i >> 5, sh = i & 31;
#pragma unroll
    for (int q = 0; q < kRaw; ++q) d[q] = 0;
    for (int e0 = 0; e0 < n; e0 += U) {  // U list entries per step: 4 U independent loads in flight (one round trip per step)
        uint32_t ent[U], w0[U], w1[U], w2[U], w3[U];
#pragma unroll
        for (int u = 0; u < U; ++u) ent[u] = ldc<COH>(list + min(e0 + u, n - 1));  // wave-uniform
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (U > 4 && e0 + u >= n) { w0[u] = w1[u] = w2[u] = w3[u] = 0; continue; }  // (uniform)
            const uint32_t *row = table + static_cast<size_t>(ent[u] >> 1) * kPlanes * Wp + w;
            w0[u] = row[0]; w1[u] = row[Wp]; w2[u] = row[2 * Wp]; w3[u] = row[3 * Wp];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (e0 + u >= n) continue;
            const int sgn = (ent[u] & 1u) ? 1 : -1;
            // pair (i, j) seen from gene i: cL(i,j) = cH(j,i), cH(i,j) = cL(j,i), likewise for the treat side
            const int cl = (w1[u] >> sh) & 1, ch = (w0[u] >> sh) & 1;
            const int tl = (w3[u] >> sh) & 1, th = (w2[u] >> sh) & 1;
            d[0] += sgn * cl; d[1] += sgn * ch; d[2] += sgn * tl; d[3] += sgn * th;
            d[4] += sgn * (cl & tl); d[5] += sgn * (cl & th); d[6] += sgn * (ch & tl); d[7] += sgn * (ch & th);
        }
    }
}

// delta_counts for a workgroup of 256 threads that owns genes 256 b .. 256 b + 255 (thread = gene) and a list of at most
// kDeltaMax entries in LDS.  A pass of the light path changes about twenty reference genes; four loads per entry in every
// thread are 320 load instructions per CU and a round trip per eight entries.  Here the 8 words x 4 planes that hold the
// workgroup's bits of a changed row are fetched once -- every entry's at the same time, one round trip, 32 n / 256 loads per
// thread -- into LDS ([entry][word] as uint4 of the four planes), and each thread then reads one uint4 per entry.  The
// eight sums are kept in packed 8-bit fields, separately for entering and leaving genes: A = cL | cH << 16,
// B = tL | tH << 8, A * B = LL | LH << 8 | HL << 16 | HH << 24.  Whole workgroup (one barrier); buf: kDeltaMax * 8 uint4.
__device__ __forceinline__ void delta_counts_block(const uint32_t *__restrict__ table, int Wp, const uint32_t *list, int n, int (&d)[kRaw], uint4 *buf)
{
    static_assert(kDeltaMax <= 255 && kDeltaMax * 32 <= 256 * 16, "8-bit fields; at most 16 fetches per thread");
    const int items = n * 32;
    uint32_t v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int id = threadIdx.x + 256 * k;
        v[k] = 0;
        if (id < items) {
            const uint32_t ent = list[id >> 5];
            v[k] = table[(static_cast<size_t>(ent >> 1) * kPlanes + ((id >> 3) & 3)) * Wp + blockIdx.x * 8 + (id & 7)];
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int id = threadIdx.x + 256 * k;
        if (id < items) reinterpret_cast<uint32_t *>(buf)[((id >> 5) * 8 + (id & 7)) * 4 + ((id >> 3) & 3)] = v[k];
    }
    lds_barrier();
    const int wi = threadIdx.x >> 5, sh = threadIdx.x & 31;
    uint32_t accA[2] = {0, 0}, accB[2] = {0, 0}, accX[2] = {0, 0};  // [0] leaving, [1] entering
#pragma unroll 4
    for (int e = 0; e < n; ++e) {
        const uint4 w = buf[e * 8 + wi];
        // pair (i, j) seen from gene i: cL(i,j) = cH(j,i), cH(i,j) = cL(j,i), likewise for the treat side
        const uint32_t A = ((w.y >> sh) & 1u) | (((w.x >> sh) & 1u) << 16);
        const uint32_t B = ((w.w >> sh) & 1u) | (((w.z >> sh) & 1u) << 8);
        const int s = list[e] & 1u;  // (uniform)
        accA[s] += A; accB[s] += B; accX[s] += A * B;
    }
    auto f = [](uint32_t x, int at) { return static_cast<int>((x >> at) & 0xFFu); };
    d[0] = f(accA[1], 0) - f(accA[0], 0);  d[1] = f(accA[1], 16) - f(accA[0], 16);
    d[2] = f(accB[1], 0) - f(accB[0], 0);  d[3] = f(accB[1], 8) - f(accB[0], 8);
    d[4] = f(accX[1], 0) - f(accX[0], 0);  d[5] = f(accX[1], 8) - f(accX[0], 8);
    d[6] = f(accX[1], 16) - f(accX[0], 16); d[7] = f(accX[1], 24) - f(accX[0], 24);
}

__device__ __forceinline__ void delta_gene(const uint32_t *__restrict__ table, int Wp, const uint32_t *__restrict__ list,
                                           int n, int32_t *__restrict__ raw, int i)
{
    int d[kRaw];
    delta_counts<false>(table, Wp, list, n, i, d);
    int4 *o = reinterpret_cast<int4 *>(raw + static_cast<size_t>(i) * kRaw);
    int4 a = o[0], b = o[1];
    a.x += d[0]; a.y += d[1]; a.z += d[2]; a.w += d[3];
    b.x += d[4]; b.y += d[5]; b.z += d[6]; b.w += d[7];
    o[0] = a; o[1] = b;
}

__device__ __forceinline__ void delta_genes(const uint32_t *__restrict__ table, int G, int Wp, const uint32_t *__restrict__ list,
                                            int n, int32_t *__restrict__ raw)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < G) delta_gene(table, Wp, list, n, raw, i);
}

// ---------------------------------------------------------------------------
// Arguments shared by every kernel of the iteration passes (:396-425).  The pass index, and with it the parity of
// the double-buffered reference mask and change list, lives in device memory (IterState), so the host can enqueue
// passes without knowing which of them will run.
struct IterArgs {
    IterState *st;
    const uint32_t *table;
    int G, Gp, Wp;
    int n_iter, n_conv;
    int a0, b0;                  // 0-based bounds of the slice of :411
    double pval_deg, padj_deg;
    uint32_t *refbits[2];        // [Wp] mask of the pass of that parity, as bits ...
    uint8_t *refbytes[2];        // [Gp] ... and as bytes
    int32_t *raw;                // [G][8]
    uint32_t *delta_list;        // [2][Gp] genes whose mask bit changes for the pass of that parity
    double *result;              // [15][G]
    double *chunk_v; uint16_t *chunk_i; double *chunk_spl;
    double *sorted_d; double *sorted_spl; double *sorted_p;
    uint32_t *rank_s; uint32_t *rank_a;
    double *part; double *blockmin; double *scal;
    int32_t *trace; int32_t *modes;
    double *cand;                // [2][kCandMax] light passes: values inside the two quantile windows
    int32_t *hist;               // [2][hist_stride] light passes: histogram of the BH ranks m_i (second copy: two-launch form, by launch parity)
    int hist_stride;
    int32_t *mrank;              // [Gp]        light passes: m_i
    int replay;                  // 1: recompute the outputs of the last executed pass from its tallies, change no state
    int k2_idx;                  // index of this k2_tally launch (for the stage timers)
    int xcc_local;               // light passes: the histogram atomics may stay in the XCD's L2 (reo_create's self-test passed)
    int32_t *olist;              // one-launch light passes: [2][kOneStride] genes near the BH cut with their delta1, by workgroup
    int hist_below;              // light passes, two-launch form: see hist_first
    uint8_t *snap;               // [2][Gp] light passes, two-launch form: mask snapshots of the cycle watch (kl_head)
    int32_t *clist; int band;    // light passes: genes near the BH cut listed by kl_rank ([2][kListStride]); half width of "near" in ranks
    int window, light_min_g;     // light passes: ranks on either side of a quantile that its window is made to hold; smallest G that uses them
    unsigned long long *stamps;  // diagnostic builds (-DREO_STAMPS): s_memrealtime marks of workgroup 0, else unused
    IterState *host_st;          // pinned host memory: the kernels that end a pass or a batch mirror what the host reads of IterState
                                 // there (passes, done, need_full, last_full, delta_cnt), so no copy is queued behind a batch; null: no mirror
};

#ifdef REO_STAMPS
#define STAMP(a, k) do { if (blockIdx.x == 0 && threadIdx.x == 0) (a).stamps[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define STAMP(a, k) do { } while (0)
#endif

// A gene's tallies broke their invariant: tell the host (both copies of the state: the one it copies and the pinned
// mirror it reads).  Any thread may call this; the word only ever goes from 0 to a code, and no kernel of the passes
// branches on it, so the plain stores race with nothing.
__device__ __forceinline__ void raise_fault(const IterArgs &a, int code)
{
    a.st->fault = code;
    if (a.host_st) a.host_st->fault = code;
}

// scal[]: 0 se of the last pass; 1..4 quantile windows wa_lo wa_hi wb_lo wb_hi for the next light pass;
// 5..8 their half widths (value units) around the exact quantiles
constexpr int kCandMax = 64;     // candidates per quantile window (one per lane of the wave that sorts them)
constexpr int kWindow = 24;      // ranks on either side of the quantile that the window is made to hold (REO_LIGHT_WINDOW overrides: tests)
constexpr int kLightMinG = 4096; // below this the sorting passes are cheap (REO_LIGHT_MIN_G overrides: tests)

// the pass the kernels of the sorting path work on: the next one, or (replay) the last executed one
__device__ __forceinline__ bool full_pass_active(const IterArgs &a, int &t, int &nref)
{
    const IterState *st = a.st;
    if (a.replay) { t = st->passes - 1; nref = st->nref_prev; return t >= 0; }
    t = st->passes; nref = st->nref;
    return !st->done && st->passes < a.n_iter && st->need_full;
}

// The K2 stage of a sorting pass: one launch, the device picks the form (whole-table scan, or update from the rows
// of the genes whose mask bit changed).  Skipped when a light pass has already brought the counters up to date.
__global__ __launch_bounds__(256) void k2_tally(IterArgs a)
{
    int t, nref;
    if (!full_pass_active(a, t, nref) || a.st->raw_pass == t) return;
    const int cur = t & 1;
    const int n = a.st->delta_cnt[cur];
    const bool full = n > kDeltaMax;
    if (a.modes && blockIdx.x == 0 && threadIdx.x == 0) a.modes[a.k2_idx] = full ? 1 : 0;  // for the stage timers
    if (full) tally_rows(reinterpret_cast<const uint4 *>(a.table), reinterpret_cast<const uint4 *>(a.refbits[cur]), a.G, a.Wp / 4, a.raw);
    else if (static_cast<int>(blockIdx.x) * 256 < a.G) delta_genes(a.table, a.G, a.Wp, a.delta_list + static_cast<size_t>(cur) * a.Gp, n, a.raw);
}

// stand-alone form for reo_tally: always a whole-table scan with the mask of parity 0
__global__ __launch_bounds__(256) void k2_scan(const uint32_t *__restrict__ table, const uint4 *__restrict__ refbits, int G, int Wp,
                                               int32_t *__restrict__ raw)
{
    tally_rows(reinterpret_cast<const uint4 *>(table), refbits, G, Wp / 4, raw);
}

// ---------------------------------------------------------------------------
// K3.  McCullagh's test for a 3x3 table, :225-259.
//   N = [[a b][b d]], n = (a, d), R = (R1, R2).  With an integer determinant a*d - b*b != 0 the weights
//   omega2 = inv(N) n are taken in closed form (exact integers, one division each).
//   The integer-singular tables are NOT all singular for the reference: :242 tests abs(det(N)) <= eps() on the
//   LU factors of Float64.(N).  a, d >= b >= 0, so a*d == b*b leaves two cases:
//     b == 0 (then a == 0 or d == 0): a pivot is exactly zero, det(lu) == 0.0 -> (1, 0, 0, 0, 0), as :243;
//     a == b == d > 0: the elimination's l21 = b * (1.0 / b) is not 1 for about one b in ten (49, 98, 103,
//       107, 161, ...), u22 = b - l21 * b is then a few ulps of b, det = b * u22 > eps, and the reference goes
//       on with the inverse of that factorisation.  What it gets is rounding noise of its LAPACK; what is
//       reproduced here, operation by operation and without fused multiply-adds (the library is built with
//       -ffp-contract=off), is the arithmetic of the oracle's pivoted Gauss-Jordan elimination
//       (oracle/reo_oracle.c, oracle_mccullagh), so that the HIP path and its checker agree in this corner
//       too.  delta1 is robust there (both log terms are equal and the weights sum to 1); nu, and with it
//       delta2 / se / z1, depend on the noise (omega2 sums to 1 or to 2).  DESIGN.md section 2.
// FULL = false: delta1 only (the passes in between need nothing else: delta2, se, z1 and the test's own p-value
// never reach the next pass -- :412 overwrites the p-value -- and are recomputed for the pass that ends the loop).
template <bool FULL>
__device__ __forceinline__ void mccullagh3(const int32_t *n, double *out)
{
    const long long n12 = n[1], n13 = n[2], n21 = n[3], n23 = n[5], n31 = n[6], n32 = n[7];
    const long long a = n12 + n13 + n21 + n31;  // N11  (:232)
    const long long b = n13 + n31;              // N12
    const long long d = n13 + n23 + n31 + n32;  // N22
    const long long R1 = n12 + n13, R2 = n13 + n23;  // :239
    const long long det = a * d - b * b;
    const double fa = static_cast<double>(a), fd = static_cast<double>(d);
    double w1 = 0.0, w2 = 0.0;  // omega2 = inv(N) n  (:245-246)
    if (det != 0) {
        const double fdet = static_cast<double>(det);
        w1 = static_cast<double>(d * (a - b)) / fdet;
        w2 = static_cast<double>(a * (d - b)) / fdet;
    } else {
        bool singular = b == 0;  // a zero pivot: det(lu(N)) is exactly 0.0
        if (!singular) {         // a == b == d > 0: the float elimination of :242, in the oracle's order of operations
            const double B = fa, inv = 1.0 / B, l = B * inv;  // row 0 scaled by the reciprocal of its pivot: (l, l | inv, 0)
            const double u = B - B * l;                       // row 1 minus B * row 0: (., u | -l, 1)
            singular = u == 0.0 || fabs(B * u) <= 2.220446049250313e-16;  // det = B * u against eps()
            if (!singular) {
                const double inv2 = 1.0 / u;
                const double i10 = (0.0 - l) * inv2, i11 = inv2;          // row 1 scaled by the reciprocal of u
                const double i00 = inv - l * i10, i01 = 0.0 - l * i11;    // row 0 minus l * row 1
                w1 = i00 * B + i01 * B;                                   // inverse times n = (B, B)
                w2 = i10 * B + i11 * B;
            }
        }
        if (singular) { out[0] = 1.0; out[1] = out[2] = out[3] = out[4] = 0.0; return; }  // :243
    }
    const double nu = 1.0 / (fa * w1 + fd * w2);                // :247
    const double r1 = static_cast<double>(R1), r2 = static_cast<double>(R2);
    const double d1 = (fa * w1 * nu) * log((r1 + 0.5) / (fa - r1 + 0.5)) +
                      (fd * w2 * nu) * log((r2 + 0.5) / (fd - r2 + 0.5));  // :248-249
    out[1] = d1;
    if (!FULL) return;
    const double A = w1 * r1 + w2 * r2, B = w1 * (fa - r1) + w2 * (fd - r2);
    const double d2 = log((0.5 + A) / (0.5 + B));               // :250
    const double v1 = 4.0 * (1.0 + 0.25 * d1 * d1) * nu;        // :251
    const double v2 = 4.0 * (1.0 + 0.25 * d2 * d2) * nu;        // :252
    const double se = sqrt((v1 + v2) * 0.5);                    // :253
    const double z1 = d1 / se;                                  // :254
    double p = erfc(fabs(z1) * 0.70710678118654752440);         // 2*min(cdf, ccdf), :255
    out[0] = p > 1.0 ? 1.0 : p;
    out[2] = d2; out[3] = se; out[4] = z1;
}

__global__ void k_mccullagh(const int32_t *__restrict__ cont, int64_t n, double *__restrict__ out)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t c[9];
    for (int t = 0; t < 9; ++t) c[t] = cont[i * 9 + t];
    double o[5];
    mccullagh3<true>(c, o);
    for (int t = 0; t < 5; ++t) out[i * 5 + t] = o[t];
}

// one gene: raw counters (registers) -> 9 tallies (:403).  Returns false when a tally is negative.  Five of the nine are
// differences of counters, which a class table keeps non-negative because a pair holds ONE state per side; a table that
// breaks this (only a caller's exchange hook can deliver one, and reo_build_pairs scans what a hook delivers) would give
// McCullagh's formulas R > n: log of a negative number, NaN delta1 -- and NaN is what the rank searches of the sorting
// path cannot take (round 3's GPU fault: k3_abs_rank stored sorted_p[G + r]).  With nine non-negative
// tallies 0 <= R1 <= a, 0 <= R2 <= d, a, d >= b >= 0 hold by construction and every number below is finite, so the
// callers replace the statistics of a gene that fails by those of an empty table and raise IterState.fault.
__device__ __forceinline__ bool tallies_from(const int4 &r0, const int4 &r1, int total, int32_t (&c)[9])
{
    const int cLt = r0.x, cHt = r0.y, tLt = r0.z, tHt = r0.w, LL = r1.x, LH = r1.y, HL = r1.z, HH = r1.w;
    c[0] = LL; c[2] = LH; c[6] = HL; c[8] = HH;
    c[1] = cLt - LL - LH;
    c[7] = cHt - HL - HH;
    c[3] = tLt - LL - HL;
    c[5] = tHt - LH - HH;
    c[4] = total - (cLt + cHt + c[3] + c[5]);
    return (c[0] | c[1] | c[2] | c[3] | c[4] | c[5] | c[6] | c[7] | c[8]) >= 0;  // (no sign bit anywhere)
}

__device__ __forceinline__ bool tallies_of(const int32_t *__restrict__ raw, bool in_ref, int nref, int i, int32_t (&c)[9])
{
    const int4 r0 = reinterpret_cast<const int4 *>(raw)[2 * i], r1 = reinterpret_cast<const int4 *>(raw)[2 * i + 1];
    return tallies_from(r0, r1, nref - (in_ref ? 1 : 0), c);  // the diagonal is never set (:363,385)
}


// stand-alone form for reo_tally (tallies only)
__global__ __launch_bounds__(256) void k3_tallies(const int32_t *__restrict__ raw, const uint8_t *__restrict__ refbytes, int nref, int G,
                                                  int32_t *__restrict__ cont)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= G) return;
    int32_t c[9];
    tallies_of(raw, refbytes[i] != 0, nref, i, c);
#pragma unroll
    for (int t = 0; t < 9; ++t) cont[static_cast<size_t>(i) * 9 + t] = c[t];
}

// sorting pass: tallies -> McCullagh -> result columns 3..15 (:404-405)
__global__ __launch_bounds__(256) void k3_derive(IterArgs a)
{
    int t, nref;
    if (!full_pass_active(a, t, nref)) return;
    const int cur = t & 1;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0 && !a.replay) {
        a.st->delta_cnt[1 - cur] = 0;  // k3_finalize of this pass fills that list
        a.st->raw_pass = t;
    }
    if (i >= a.G) return;
    int32_t c[9];
    const bool ok = tallies_of(a.raw, a.refbytes[cur][i] != 0, nref, i, c);
    double o[5] = {1.0, 0.0, 0.0, 0.0, 0.0};
    if (ok) mccullagh3<true>(c, o);
    if (!ok || !(fabs(o[1]) < INFINITY)) {  // (the second test cannot fire on non-negative tallies: belt and braces)
        o[0] = 1.0; o[1] = o[2] = o[3] = o[4] = 0.0;
        raise_fault(a, kFaultTallies);
    }
    const size_t Gs = a.G;
    double *result = a.result;
    result[i] = o[0];
    result[Gs + i] = 1.0;
#pragma unroll
    for (int q = 0; q < 9; ++q) result[(2 + q) * Gs + i] = static_cast<double>(c[q]);
    result[11 * Gs + i] = o[1]; result[12 * Gs + i] = o[2];
    result[13 * Gs + i] = o[3]; result[14 * Gs + i] = o[4];
}

// Sum over the 256 threads of a workgroup, the same bits in every thread and every workgroup: an
// xor-butterfly inside each wave (a + b == b + a, so all lanes of a wave agree at every step), then the
// four wave sums in a fixed order.  Two barriers (the second lets `red` be reused at once).
__device__ __forceinline__ double block_sum_256(double v, double *red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    lds_barrier();
    const double r = (red[0] + red[1]) + (red[2] + red[3]);
    lds_barrier();
    return r;
}

// two sums at once (the same bits as two calls of block_sum_256: the same operations in the same order on each value)
__device__ __forceinline__ void block_sum2_256(double v, double w, double *red, double &rv, double &rw)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const double pv = __shfl_xor(v, o, 64), pw = __shfl_xor(w, o, 64); v += pv; w += pw; }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = v; red[4 + (threadIdx.x >> 6)] = w; }
    lds_barrier();
    rv = (red[0] + red[1]) + (red[2] + red[3]);
    rw = (red[4] + red[5]) + (red[6] + red[7]);
    lds_barrier();
}

// ---- ranking: the sort of :409 and the order BH needs (:413) ---------------
// (1) k3_sort_chunks: bitonic sort of 1024-gene chunks by delta1;
// (2) k3_merge_rank: rank of a gene = its position in its own chunk + the
//     number of smaller elements of every other chunk (binary searches; chunks
//     hold contiguous gene ranges, so "smaller gene index" is "earlier chunk");
//     this also scatters delta1 into globally sorted order;
// (3) k3_abs_rank: rank of |delta1| descending (= rank of pval ascending, pval
//     being a decreasing function of |delta1|, :412) from three binary searches
//     in the sorted array, plus per-block moments of the 5 %-95 % slice.
// Ties are broken consistently, so both ranks are permutations.

// compare-exchange with the partner's element: keep the smaller value if keep_min, else the larger.
// Equal values need no tie-break here: any consistent order among them still makes the ranks
// permutations, and every quantity derived from them (sorted values, BH, trimmed std) is identical.
// Value of lane (l ^ J): quad permutes (DPP, no LDS traffic) for J = 1, 2; ds_swizzle (no address register) for
// J = 4, 8, 16; ds_bpermute for 32.  __shfl_xor always takes the last route (two dependent LDS round trips per double);
// the 21 stages of the sort below were 2.2 us of kl_rank that way.
template <int J>
__device__ __forceinline__ int lane_xor_b32(int v)
{
    if constexpr (J == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
    else if constexpr (J == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    else if constexpr (J < 32) return __builtin_amdgcn_ds_swizzle(v, (J << 10) | 0x1F);          // bit mode: and 0x1F, or 0, xor J
    else return __shfl_xor(v, J, 64);
}
template <int J>
__device__ __forceinline__ double lane_xor(double x)
{
    const long long b = __double_as_longlong(x);
    const int lo = lane_xor_b32<J>(static_cast<int>(b)), hi = lane_xor_b32<J>(static_cast<int>(b >> 32));
    return __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo));
}

__device__ __forceinline__ void cmpx(double &v, uint32_t &i, double pv, uint32_t pi, bool keep_min)
{
    const bool take = ((pv < v) & keep_min) | ((pv > v) & !keep_min);  // (bitwise: no divergent branches)
    v = take ? pv : v;
    i = take ? pi : i;
}

// Bitonic sort of one kSortChunk-gene chunk, one element per thread: exchange distances 1..32 go through wave
// shuffles, and only the stages with distance >= 64 (partner in another wave) use LDS + a barrier.
__global__ __launch_bounds__(kSortChunk) void k3_sort_chunks(IterArgs a)
{
    int tp, nref;
    if (!full_pass_active(a, tp, nref)) return;
    __shared__ double sv[kSortChunk];
    __shared__ uint16_t si[kSortChunk];
    const double *d1 = a.result + 11 * static_cast<size_t>(a.G);
    const int t = threadIdx.x, base = blockIdx.x * kSortChunk;
    double v = base + t < a.G ? d1[base + t] : INFINITY;  // padding sorts to the end of the last chunk
    uint32_t id = t;
    for (int k = 2; k <= kSortChunk; k <<= 1) {
        const bool up = (t & k) == 0;
        for (int j = k >> 1; j >= 64; j >>= 1) {  // partner in another wave
            lds_barrier();
            sv[t] = v; si[t] = static_cast<uint16_t>(id);
            lds_barrier();
            cmpx(v, id, sv[t ^ j], si[t ^ j], ((t & j) == 0) == up);
        }
        // inside the wave: quad permutes / swizzles (lane_xor), starting at distance min(k / 2, 32)
        const int j0 = (k >> 1) < 32 ? (k >> 1) : 32;  // workgroup-uniform
        auto stage = [&](auto J) {
            constexpr int jj = decltype(J)::value;
            const double pv = lane_xor<jj>(v);
            const uint32_t pi = static_cast<uint32_t>(lane_xor_b32<jj>(static_cast<int>(id)));
            cmpx(v, id, pv, pi, ((t & jj) == 0) == up);
        };
        if (j0 >= 32) stage(std::integral_constant<int, 32>{});
        if (j0 >= 16) stage(std::integral_constant<int, 16>{});
        if (j0 >= 8) stage(std::integral_constant<int, 8>{});
        if (j0 >= 4) stage(std::integral_constant<int, 4>{});
        if (j0 >= 2) stage(std::integral_constant<int, 2>{});
        stage(std::integral_constant<int, 1>{});
    }
    a.chunk_v[base + t] = v; a.chunk_i[base + t] = static_cast<uint16_t>(id);
    // every 32nd element once more, packed: k3_merge_rank stages these into LDS with contiguous loads
    if ((t & 31) == 0) a.chunk_spl[blockIdx.x * (kSortChunk / 32) + (t >> 5)] = v;
}

// A search in a sorted array a[0..n) for the count of elements that sort before v (`<` when strict, `<=`
// otherwise), in two levels: a coarse search over every 2^LOG-th element staged in LDS narrows it to a
// half-open index range [l, h) of fewer than 2^LOG elements, then LOG halving steps in global memory finish.
// The two levels are separate calls so that a thread with several searches can run their global steps in
// lockstep: the loads of one step are independent, so k searches cost one round trip per step, not k.
// EVERY load of a lockstep step is issued unconditionally; a search that is switched off or already finished
// must therefore be given an in-bounds dummy (see the callers: this is where round 1's memory fault came from).
struct Range { int l, h; };

template <int LOG>
__device__ __forceinline__ Range coarse_range(const double *spl, int nspl, int n, double v, bool strict)
{
    int lo = 0, hi = nspl;  // splitter m = a[m << LOG]
    while (lo < hi) { const int m = (lo + hi) >> 1; const double w = spl[m]; if (strict ? (w < v) : (w <= v)) lo = m + 1; else hi = m; }
    if (lo == 0) return Range{0, 0};
    return Range{((lo - 1) << LOG) + 1, min(n, lo << LOG)};
}

__device__ __forceinline__ void halve(Range &r, double w, double v, bool strict)
{
    if (r.l < r.h) { const int m = (r.l + r.h) >> 1; if (strict ? (w < v) : (w <= v)) r.l = m + 1; else r.h = m; }
}

// Rank of an element = its position in its own chunk + the number of smaller elements in every
// other chunk.  A group of kMergeLanes lanes serves one element: each lane searches different
// chunks (two-level: every 32nd element of every chunk sits in LDS as a splitter; log2(kSortChunk/32)
// LDS steps pick the 32-element segment, five global steps finish inside it), then the counts are
// summed across the group with shuffles.  Chunks hold contiguous gene ranges, so "smaller gene
// index" is "earlier chunk": equal values of earlier chunks sort first.
constexpr int kMergeLanes = 16;
constexpr int kMergeThreads = 256;   // 16 elements per workgroup share one copy of the splitters
constexpr int kSplit = kSortChunk / 32;  // splitters per chunk

__global__ __launch_bounds__(kMergeThreads) void k3_merge_rank(IterArgs a, int nchunk)
{
    int tp, nref;
    if (!full_pass_active(a, tp, nref)) return;
    const int G = a.G;
    const double *cv = a.chunk_v;
    __shared__ double slice[kMergeThreads / kMergeLanes];
    extern __shared__ double spl[];  // [chunk][kSplit]: 32 splitters per chunk (dynamic: 16 KB at 65 535 genes, 64 KB at 262 143)
    for (int t = threadIdx.x; t < nchunk * kSplit; t += kMergeThreads) spl[t] = a.chunk_spl[t];  // = cv[chunk][32 m], packed
    __syncthreads();
    const int e = blockIdx.x * (kMergeThreads / kMergeLanes) + threadIdx.x / kMergeLanes;  // element (position in the chunked array)
    const int sub = threadIdx.x % kMergeLanes;
    const int c = e / kSortChunk, p = e % kSortChunk;
    const bool live = e < nchunk * kSortChunk;
    const int gene = live ? c * kSortChunk + a.chunk_i[e] : G;
    const double v = live ? cv[e] : 0.0;
    int count = 0;
    if (gene < G) {
        // this lane's chunks, two at a time so that the five global steps of both searches overlap
        for (int cc = sub; cc < nchunk; cc += 2 * kMergeLanes) {
            const int c2 = cc + kMergeLanes;
            const bool on1 = cc != c, on2 = c2 < nchunk && c2 != c;
            const int n1 = min(kSortChunk, G - cc * kSortChunk), n2 = on2 ? min(kSortChunk, G - c2 * kSortChunk) : 0;
            const bool s1 = cc > c, s2 = c2 > c;  // equal values of earlier chunks sort first
            Range r1 = on1 ? coarse_range<5>(spl + cc * kSplit, (n1 + 31) >> 5, n1, v, s1) : Range{0, 0};
            Range r2 = on2 ? coarse_range<5>(spl + c2 * kSplit, (n2 + 31) >> 5, n2, v, s2) : Range{0, 0};
            // c2 may lie past the last chunk: the switched-off search reads chunk cc instead (in bounds)
            const double *ch1 = cv + cc * kSortChunk, *ch2 = cv + (on2 ? c2 : cc) * kSortChunk;
#pragma unroll
            for (int step = 0; step < 5; ++step) {
                const double w1 = ch1[min((r1.l + r1.h) >> 1, kSortChunk - 1)], w2 = ch2[min((r2.l + r2.h) >> 1, kSortChunk - 1)];
                halve(r1, w1, v, s1); halve(r2, w2, v, s2);
            }
            count += r1.l + r2.l;
        }
    }
#pragma unroll
    for (int o = kMergeLanes >> 1; o > 0; o >>= 1) count += __shfl_xor(count, o, 64);
    bool in = false;
    if (sub == 0 && gene < G) {
        // (a permutation of 0..G-1 whenever every delta1 is finite, which k3_derive sees to; the clamp keeps the three
        //  stores below inside their allocations whatever the values are)
        const int rank = min(p + count, G - 1);
        a.rank_s[gene] = rank;
        a.sorted_d[rank] = v;
        if ((rank & 63) == 0) a.sorted_spl[rank >> 6] = v;  // packed splitters for k3_abs_rank
        in = rank >= a.a0 && rank <= a.b0;  // inside the 5 %-95 % slice of :411
    }
    // moments (count, mean, M2) of this block's elements that fall into the slice; combined in k3_abs_rank
    if (sub == 0) slice[threadIdx.x / kMergeLanes] = in ? v : NAN;
    __syncthreads();
    if (threadIdx.x < 64) {  // first wave: fixed-order butterfly over the workgroup's elements (NaN = not in the slice)
        constexpr int kElems = kMergeThreads / kMergeLanes;
        static_assert(kElems <= 64, "one wave reduces the workgroup's slice values");
        const double x = static_cast<int>(threadIdx.x) < kElems ? slice[threadIdx.x] : NAN;
        const bool has = x == x;
        double n = has ? 1.0 : 0.0, sum = has ? x : 0.0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { n += __shfl_xor(n, o, 64); sum += __shfl_xor(sum, o, 64); }
        const double mean = n > 0.0 ? sum / n : 0.0;
        double m2 = has ? (x - mean) * (x - mean) : 0.0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m2 += __shfl_xor(m2, o, 64);
        if (threadIdx.x == 0) { a.part[3 * blockIdx.x] = n; a.part[3 * blockIdx.x + 1] = mean; a.part[3 * blockIdx.x + 2] = m2; }
    }
}

// Chan's combination of per-block moments (n_b, mean_b, M2_b), the same bits in every thread of every workgroup:
// n = sum n_b, mean = sum n_b mean_b / n, M2 = sum [M2_b + n_b (mean_b - mean)^2], fixed-order tree sums.
__device__ __forceinline__ void combine_moments(const double *__restrict__ part, int npart, double *red, double &n, double &mean, double &m2)
{
    double nb = 0.0, sb = 0.0;
    for (int t = threadIdx.x; t < npart; t += 256) { const double n_ = part[3 * t]; nb += n_; sb += n_ * part[3 * t + 1]; }
    n = block_sum_256(nb, red);
    mean = block_sum_256(sb, red) / n;
    double qb = 0.0;
    for (int t = threadIdx.x; t < npart; t += 256) {
        const double n_ = part[3 * t], d_ = part[3 * t + 1] - mean;
        qb += part[3 * t + 2] + n_ * d_ * d_;
    }
    m2 = block_sum_256(qb, red);
}

// pvalue(Normal(0, se), delta1, tail = :both), :412
__device__ __forceinline__ double normal_p(double v, double se)
{
    if (se == 0.0) return 0.0;  // Normal(0,0): cdf/ccdf degenerate to a step, the smaller tail is 0
    const double p = erfc(fabs(v) / se * 0.70710678118654752440);
    return p > 1.0 ? 1.0 : p;
}

// |delta1| ranks (= rank of pval ascending), then se = std of the 5 %-95 % slice of the sorted delta1
// (n-1 estimator, :409-411) from the per-block moments of k3_merge_rank (every workgroup combines them
// the same way), then pval (:412) -> column 1, and into rank order for the BH step.
__global__ __launch_bounds__(256) void k3_abs_rank(IterArgs a, int npart)
{
    int tp, nref;
    if (!full_pass_active(a, tp, nref)) return;
    const int G = a.G;
    const double *sorted_d = a.sorted_d;
    __shared__ double red[256];
    extern __shared__ double spl[];  // every 64th element of the sorted vector (dynamic: 8 KB at 65 535 genes, 32 KB at 262 143)
    const int nspl = (G + 63) >> 6;
    for (int t = threadIdx.x; t < nspl; t += 256) spl[t] = a.sorted_spl[t];  // = sorted_d[64 t], packed
    double n, mean, m2;
    combine_moments(a.part, npart, red, n, mean, m2);  // (its first barrier also publishes spl)
    const double se = sqrt(m2 / (n - 1.0));
    if (blockIdx.x == 0 && threadIdx.x == 0) a.scal[0] = se;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= G) return;
    const double v = a.result[11 * static_cast<size_t>(G) + i];
    const int r = a.rank_s[i];
    // three counts in the sorted vector: < v, <= -v, <= v; their six global steps run in lockstep
    Range r1 = coarse_range<6>(spl, nspl, G, v, true), r2 = coarse_range<6>(spl, nspl, G, -v, false),
          r3 = coarse_range<6>(spl, nspl, G, v, false);
#pragma unroll
    for (int step = 0; step < 6; ++step) {
        const double w1 = sorted_d[min((r1.l + r1.h) >> 1, G - 1)], w2 = sorted_d[min((r2.l + r2.h) >> 1, G - 1)],
                     w3 = sorted_d[min((r3.l + r3.h) >> 1, G - 1)];
        halve(r1, w1, v, true); halve(r2, w2, -v, false); halve(r3, w3, v, false);
    }
    const int lbv = r1.l, ubn = r2.l, ubv = r3.l;
    int rank;
    if (v > 0.0) {  // larger |w|: w > v or w < -v; ties: the negatives -v first, then equals of v in sorted order
        rank = (G - ubv) + ubn + (r - lbv);
    } else {        // larger |w|: w < v or w > -v
        rank = lbv + (G - ubn) + (r - lbv);
    }
    // rank is a permutation of 0..G-1 when sorted_d is sorted and v is one of its (finite) elements.  Whatever the three
    // searches return otherwise -- a NaN v fails every comparison: all three counts 0, rank = G + r, the out-of-bounds
    // store of round 3's fault -- the clamp keeps this store, and k3_finalize's sorted_p[rank_a[i]] and tail[rank >> 10],
    // inside their allocations.
    rank = min(max(rank, 0), G - 1);
    a.rank_a[i] = rank;
    const double p = normal_p(v, se);
    a.result[i] = p;
    a.sorted_p[rank] = p;
}

// Benjamini-Hochberg step-up (:413), part 1: p_(r) * (n/r) and the reverse
// cumulative minimum inside blocks of 1024 ranks (in place) + the block minima.
__global__ __launch_bounds__(1024) void k3_bh_local(IterArgs a)
{
    int tp, nref;
    if (!full_pass_active(a, tp, nref)) return;
    const int G = a.G;
    __shared__ double wmin[16];
    const int r = blockIdx.x * 1024 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double v = r < G ? a.sorted_p[r] * (static_cast<double>(G) / static_cast<double>(r + 1)) : INFINITY;
    // reverse cumulative minimum inside the wave (min is exact: any order gives the same bits) ...
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double w = __shfl_down(v, o, 64);
        if (lane + o < 64) v = w < v ? w : v;
    }
    if (lane == 0) wmin[wave] = v;  // minimum of the whole wave
    __syncthreads();
    // ... then over the waves behind this one
    double t = INFINITY;
    for (int k = wave + 1; k < 16; ++k) { const double w = wmin[k]; t = w < t ? w : t; }
    v = t < v ? t : v;
    if (r < G) a.sorted_p[r] = v;
    if (threadIdx.x == 0) a.blockmin[blockIdx.x] = v;
}

// The end of a sorting pass (k3_finalize): the non-DEG mask
// inds (:417) of gene i as bytes and bits for the next pass, the list of genes whose mask bit changes (the next
// pass updates its tallies from their rows alone, delta_genes), and -- by the last workgroup to finish -- the loop
// control of :418-424 on the device-side iteration state.  Every thread of the grid (Gp threads) must call it.
// Returns -1 except in thread 0 of the last workgroup, where it returns 1 when more genes changed than a tally
// update can take (the next pass must scan the table) and 0 otherwise; that thread then sets st->need_full.
__device__ __forceinline__ int publish_mask(const IterArgs &a, int t, int i, bool ind, bool in_cut)
{
    IterState *st = a.st;
    const int cur = t & 1, nxt = 1 - cur;
    if (i < a.Gp) a.refbytes[nxt][i] = ind ? 1 : 0;
    const bool changed = i < a.G && ind != (a.refbytes[cur][i] != 0);
    const unsigned long long cm = __ballot(changed);
    if (cm) {
        int basepos = 0;
        if ((threadIdx.x & 63) == 0) basepos = atomicAdd(&st->delta_cnt[nxt], __popcll(cm));
        basepos = __shfl(basepos, 0, 64);
        const int at = basepos + __popcll(cm & ((1ULL << (threadIdx.x & 63)) - 1ULL));
        if (changed && at < kDeltaMax) a.delta_list[static_cast<size_t>(nxt) * a.Gp + at] = (static_cast<uint32_t>(i) << 1) | (ind ? 1u : 0u);
    }
    const unsigned long long m = __ballot(ind), mc = __ballot(in_cut);
    __shared__ int wave_nn[4], wave_kc[4];
    if ((threadIdx.x & 63) == 0) {
        if (i < a.Gp) {
            a.refbits[nxt][i >> 5] = static_cast<uint32_t>(m);
            a.refbits[nxt][(i >> 5) + 1] = static_cast<uint32_t>(m >> 32);
        }
        wave_nn[threadIdx.x >> 6] = __popcll(m);
        wave_kc[threadIdx.x >> 6] = __popcll(mc);
    }
    __syncthreads();
    if (threadIdx.x != 0) return -1;
    // one atomic per workgroup, not per wave: serialised updates of one word were a third of this kernel
    const int nn_blk = wave_nn[0] + wave_nn[1] + wave_nn[2] + wave_nn[3];
    if (nn_blk) atomicAdd(&st->nn_acc, nn_blk);
    const int kc_blk = wave_kc[0] + wave_kc[1] + wave_kc[2] + wave_kc[3];  // genes inside the BH cut: their number IS the cut (padj <= padj_deg <=> rank <= k*)
    if (kc_blk) atomicAdd(&st->kcut_acc, kc_blk);
    __threadfence();
    const int tk = atomicAdd(&st->ticket, 1);
    if (tk != static_cast<int>(gridDim.x) - 1) return -1;
    __threadfence();
    const int nn = atomicAdd(&st->nn_acc, 0);  // sum(inds), :417-418
    a.trace[2 * t] = a.G - nn;
    a.trace[2 * t + 1] = nn;
    st->passes = t + 1;
    st->cyc_pow = 0;  // (cycle watch of the light passes: this mask step is not in its books -- the next light pass starts a new snapshot)
    st->nref_prev = st->nref;
    const int diff = st->nref - nn;
    if ((diff < 0 ? -diff : diff) < a.n_conv) {
        st->done = 1;       // :419-422
    } else {
        st->i_iter += 1;    // :423
        st->nref = nn;      // ref_gene_vec = inds, :424
    }
    st->last_full = 1;
    st->kstar = atomicAdd(&st->kcut_acc, 0);  // the one-launch light passes centre their band on it
    st->kcut_acc = 0;
    st->nn_acc = 0;
    st->ticket = 0;
    return atomicAdd(&st->delta_cnt[nxt], 0) > kDeltaMax ? 1 : 0;
}

// BH part 2 + mask update of a sorting pass.  padj -> column 2 (:416), inds (:417), loop control (:418-424); the last
// workgroup also lays the quantile windows for a following light pass around ranks a0 and b0 of the sorted delta1.
// replay: padj only.
__global__ __launch_bounds__(256) void k3_finalize(IterArgs a)
{
    int t, nref;
    if (!full_pass_active(a, t, nref)) return;
    const int G = a.G;
    constexpr int kMaxBlocks = (kMaxGenes + 1023) / 1024;
    __shared__ double tail[kMaxBlocks + 1];
    const int nb = (G + 1023) / 1024;
    __shared__ double bm[kMaxBlocks];
    for (int b = threadIdx.x; b < nb; b += 256) bm[b] = a.blockmin[b];
    __syncthreads();
    for (int b = threadIdx.x; b <= nb; b += 256) {  // tail[b] = minimum over the blocks after b
        double run = INFINITY;
        for (int k = b + 1; k < nb; ++k) { const double m = bm[k]; run = m < run ? m : run; }
        tail[b] = run;
    }
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    bool ind = false, in_cut = false;
    if (i < G) {
        const uint32_t r = a.rank_a[i];
        double q = a.sorted_p[r];
        const double tl = tail[r >> 10];
        q = tl < q ? tl : q;
        q = q < 1.0 ? q : 1.0;
        a.result[static_cast<size_t>(G) + i] = q;
        in_cut = q <= a.padj_deg;
        ind = !(a.result[i] <= a.pval_deg && in_cut);
    }
    if (a.replay) return;
    const int over = publish_mask(a, t, i, ind, in_cut);
    if (over < 0) return;
    // quantile windows: the values kWindow ranks on either side of the slice bounds, kept as widths around the
    // exact quantiles so that a light pass can re-centre them on its own quantiles
    const int W = a.window;
    bool ok = G >= a.light_min_g && a.a0 - W >= 0 && a.b0 + W < G && a.a0 + W < a.b0 - W;
    if (ok) {
        const double va = a.sorted_d[a.a0], vb = a.sorted_d[a.b0];
        const double wa_lo = a.sorted_d[a.a0 - W], wa_hi = a.sorted_d[a.a0 + W];
        const double wb_lo = a.sorted_d[a.b0 - W], wb_hi = a.sorted_d[a.b0 + W];
        ok = wa_hi < wb_lo;
        a.scal[1] = wa_lo; a.scal[2] = wa_hi; a.scal[3] = wb_lo; a.scal[4] = wb_hi;
        a.scal[5] = va - wa_lo; a.scal[6] = wa_hi - va; a.scal[7] = vb - wb_lo; a.scal[8] = wb_hi - vb;
    }
    a.st->need_full = (over || !ok) ? 1 : 0;
    if (a.host_st) {  // (this thread wrote every one of these fields, here or in publish_mask)
        IterState *h = a.host_st;
        h->passes = t + 1; h->done = a.st->done; h->need_full = (over || !ok) ? 1 : 0; h->last_full = 1;
        h->delta_cnt[(t + 1) & 1] = a.st->delta_cnt[(t + 1) & 1];
    }
}

// ---------------------------------------------------------------------------
// Light passes.  A pass needs three things of the G values delta1: the trimmed standard deviation (:409-411), the
// BH decision padj <= padj_deg (:413,417) and the counts of :418 -- none of which needs the sorted vector:
//  * the slice bounds are two order statistics.  From one pass to the next delta1 moves little, so the values that
//    can hold rank a0 (b0) are those inside a narrow window around the previous pass's quantile: the pass counts
//    the values below each window, collects the (at most 64) values inside, and every workgroup sorts just those, reads
//    the exact order statistics off them and combines the slice's moments (block partials of the values strictly
//    between the windows + the window values inside the slice).  A window that fails to hold its order statistic
//    makes the pass fall back to the sorting path;
//  * BH: with m_i = min{r : p_i (n/r) <= padj_deg} (the same floating-point expression as the step-up rule) and
//    H(r) = #{i : m_i <= r}, the rule's cut is k* = max{r : H(r) >= r} and padj_i <= padj_deg <=> m_i <= k*:
//    a histogram and one scan instead of a sort (proof in DESIGN.md).
// Only the pass that ends the loop needs padj values and the other output columns: they are recomputed from its
// tallies by the sorting path (replay).  Two launches per pass (kl_head, kl_rank) instead of seven, or one persistent
// launch (kl_persist); none of them searches or sorts G values.

// Bitonic sort of the 64 doubles a wave holds (one per lane), ascending by lane: 21 compare-exchange stages, no barrier.
template <int K, int J>
__device__ __forceinline__ void sort_stage(double &x, int t)
{
    const bool up = (t & K) == 0 || K == 64;
    const bool keep_min = ((t & J) == 0) == up;
    const double px = lane_xor<J>(x);
    const bool take = ((px < x) & keep_min) | ((px > x) & !keep_min);  // (bitwise: the ?: form became divergent branches to far-away blocks)
    x = take ? px : x;
}
__device__ __forceinline__ double sort64(double x)
{
    const int t = threadIdx.x & 63;
    sort_stage<2, 1>(x, t);
    sort_stage<4, 2>(x, t); sort_stage<4, 1>(x, t);
    sort_stage<8, 4>(x, t); sort_stage<8, 2>(x, t); sort_stage<8, 1>(x, t);
    sort_stage<16, 8>(x, t); sort_stage<16, 4>(x, t); sort_stage<16, 2>(x, t); sort_stage<16, 1>(x, t);
    sort_stage<32, 16>(x, t); sort_stage<32, 8>(x, t); sort_stage<32, 4>(x, t); sort_stage<32, 2>(x, t); sort_stage<32, 1>(x, t);
    sort_stage<64, 32>(x, t); sort_stage<64, 16>(x, t); sort_stage<64, 8>(x, t); sort_stage<64, 4>(x, t); sort_stage<64, 2>(x, t); sort_stage<64, 1>(x, t);
    return x;
}

// sum over the 64 lanes of a wave, the same bits in every lane (xor butterfly, 32 first)
__device__ __forceinline__ double wave_sum(double v)
{
    v += lane_xor<32>(v); v += lane_xor<16>(v); v += lane_xor<8>(v); v += lane_xor<4>(v); v += lane_xor<2>(v); v += lane_xor<1>(v);
    return v;
}

// the BH cut k* = max{r in 1..G : H(r) >= r}, H = running sum of the histogram (bin b = rank b + 1), found by every
// workgroup for itself.  Tiles of 2048 bins (eight consecutive bins per thread), sixteen tiles per round so that their
// loads are in flight together and their wave scans interleave; two barriers per round.  The histogram is padded
// with zero bins to whole rounds.
// rmax: H never exceeds the number of genes with a finite rank, so no r above that count can qualify: only the
// tiles that reach up to rank min(G, rmax) are read (a few thousand bins instead of G).
template <bool COH>
__device__ __forceinline__ int bh_cut(const int32_t *hist, int G, int rmax)
{
    __shared__ __attribute__((aligned(16))) int wsum[16][4];
    __shared__ int wbest[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ntile = (min(G, max(rmax, 1)) + 2047) / 2048;
    int best = 0, carry = 0;
    for (int q0 = 0; q0 < ntile; q0 += 16) {
        int hv[16][8];
        int s[16], inc[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const size_t o = (static_cast<size_t>(q0 + e) * 256 + threadIdx.x) * 8;
            if (q0 + e >= ntile) {  // workgroup-uniform
#pragma unroll
                for (int u = 0; u < 8; ++u) hv[e][u] = 0;
            } else if (COH) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned long long w = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(hist + o) + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    hv[e][2 * u] = static_cast<int>(w & 0xFFFFFFFFull); hv[e][2 * u + 1] = static_cast<int>(w >> 32);
                }
            } else {
                const int4 h0 = reinterpret_cast<const int4 *>(hist + o)[0], h1 = reinterpret_cast<const int4 *>(hist + o)[1];
                hv[e][0] = h0.x; hv[e][1] = h0.y; hv[e][2] = h0.z; hv[e][3] = h0.w; hv[e][4] = h1.x; hv[e][5] = h1.y; hv[e][6] = h1.z; hv[e][7] = h1.w;
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            s[e] = ((hv[e][0] + hv[e][1]) + (hv[e][2] + hv[e][3])) + ((hv[e][4] + hv[e][5]) + (hv[e][6] + hv[e][7]));
            inc[e] = s[e];
        }
        const int nt = min(16, ntile - q0);  // workgroup-uniform: tiles of this round that hold anything
#pragma unroll
        for (int o = 1; o < 64; o <<= 1)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (e < nt) { const int u = __shfl_up(inc[e], o, 64); if (lane >= o) inc[e] += u; }
        if (lane == 63)
#pragma unroll
            for (int e = 0; e < 16; ++e) wsum[e][wave] = inc[e];
        __syncthreads();
        int4 ws[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) ws[e] = *reinterpret_cast<const int4 *>(wsum[e]);  // sixteen independent 16-byte reads
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            int run = carry + inc[e] - s[e] + (wave > 0 ? ws[e].x : 0) + (wave > 1 ? ws[e].y : 0) + (wave > 2 ? ws[e].z : 0);
            const int r0 = ((q0 + e) * 256 + threadIdx.x) * 8 + 1;  // rank of the first of this thread's eight bins
#pragma unroll
            for (int u = 0; u < 8; ++u) { run += hv[e][u]; if (r0 + u <= G && run >= r0 + u) best = max(best, r0 + u); }
            carry += ws[e].x + ws[e].y + ws[e].z + ws[e].w;
        }
        __syncthreads();  // wsum is read by everyone before the next round overwrites it
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int u = __shfl_xor(best, o, 64); best = u > best ? u : best; }
    if (lane == 0) wbest[wave] = best;
    __syncthreads();
    const int k = max(max(wbest[0], wbest[1]), max(wbest[2], wbest[3]));
    __syncthreads();
    return k;
}

// m = the smallest rank r at which the step-up rule's p (n/r) is within padj_deg (n + 1: none)
__device__ __forceinline__ int bh_rank(double p, int G, double al)
{
    const double Gd = static_cast<double>(G);
    auto within = [&](int r) { return p * (Gd / static_cast<double>(r)) <= al; };
    if (!within(G)) return G + 1;
    const double est = ceil(p * Gd / al);
    int m = est < 1.0 ? 1 : (est > Gd ? G : static_cast<int>(est));
    while (m > 1 && within(m - 1)) --m;
    while (!within(m)) ++m;  // within(G) holds
    return m;
}

// The two order statistics and the slice's moments from the window members and the block partials, by every workgroup
// for itself (wave 0 sorts window A, wave 1 window B, <= 64 values each).  Returns false when a window lost its order
// statistic.  sel: LDS scratch [2][4].
// x: the window member this lane holds (wave 0: window A, wave 1: window B; lanes past the count: anything);
// pn, pm, pq: the block partial this thread holds (threads past the number of partials: zeros).
__device__ __forceinline__ bool slice_std_vals(const IterArgs &a, double x, const double (&pn)[kPartPer], const double (&pm)[kPartPer], const double (&pq)[kPartPer],
                                               int below_a, int below_b, int cnt_a, int cnt_b, double (*sel)[4], double *red, double &se, double &va, double &vb)
{
    const int ia = a.a0 - below_a, ib = a.b0 - below_b;  // positions of the order statistics inside the sorted windows
    if (!(cnt_a <= kCandMax && cnt_b <= kCandMax && ia >= 0 && ia < cnt_a && ib >= 0 && ib < cnt_b)) return false;  // workgroup-uniform
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave < 2) {
        const int cnt = wave ? cnt_b : cnt_a, pos = wave ? ib : ia;
        x = sort64(lane < cnt ? x : INFINITY);
        const bool in = wave ? lane <= pos : (lane >= pos && lane < cnt);
        const double n = static_cast<double>(wave ? pos + 1 : cnt - pos);
        const double mean = wave_sum(in ? x : 0.0) / n;
        const double q = wave_sum(in ? (x - mean) * (x - mean) : 0.0);
        const double stat = __shfl(x, pos, 64);
        if (lane == 0) { sel[wave][0] = stat; sel[wave][1] = n; sel[wave][2] = mean; sel[wave][3] = q; }
    }
    // Chan's combination of the block partials (the same bits in every thread of every workgroup; one partial per
    // thread); its barriers also publish sel
    // (a thread holds the partials of workgroups t, t + 256, ...: one below 65 537 genes -- the others are zeros and change no bit)
    double n0, mean0, sn = 0.0, sm = 0.0;
#pragma unroll
    for (int u = 0; u < kPartPer; ++u) { sn += pn[u]; sm += pn[u] * pm[u]; }
    block_sum2_256(sn, sm, red, n0, mean0);
    mean0 /= n0;
    double sq = 0.0;
#pragma unroll
    for (int u = 0; u < kPartPer; ++u) sq += pq[u] + pn[u] * (pm[u] - mean0) * (pm[u] - mean0);
    double q0 = block_sum_256(sq, red);
    va = sel[0][0]; vb = sel[1][0];
    const double n1 = sel[0][1], mean1 = sel[0][2], q1 = sel[0][3], n2 = sel[1][1], mean2 = sel[1][2], q2 = sel[1][3];
    lds_barrier();  // sel may be rewritten by the next pass
    if (!(n0 > 0.0)) { mean0 = 0.0; q0 = 0.0; }  // no value between the windows: the mean above divided by zero
    const double n = n0 + n1 + n2;
    const double mean = (n0 * mean0 + n1 * mean1 + n2 * mean2) / n;
    const double m2 = (q0 + n0 * (mean0 - mean) * (mean0 - mean)) + (q1 + n1 * (mean1 - mean) * (mean1 - mean)) +
                      (q2 + n2 * (mean2 - mean) * (mean2 - mean));
    se = sqrt(m2 / (n - 1.0));
    return static_cast<int>(n) == a.b0 - a.a0 + 1;
}

// The same, with the three jobs on three waves and ONE barrier (kl_rank): wave 0 sorts window A, wave 1 window B, wave 2 combines
// the block partials by itself -- lane l holds the partials of workgroups l, l + 64, ... (kPartLane of them) and the sums are
// wave sums -- instead of three workgroup-wide sums with two barriers each behind the sorts.  Every workgroup does the same
// operations in the same order, so all of them get the same bits (not the bits of slice_std_vals: another summation order).
constexpr int kPartLane = kListWgs / 64;
__device__ __forceinline__ bool slice_std_split(const IterArgs &a, double x, const double (&pn)[kPartLane], const double (&pm)[kPartLane], const double (&pq)[kPartLane],
                                                int below_a, int below_b, int cnt_a, int cnt_b, double (*sel)[4], double &se, double &va, double &vb)
{
    const int ia = a.a0 - below_a, ib = a.b0 - below_b;  // positions of the order statistics inside the sorted windows
    if (!(cnt_a <= kCandMax && cnt_b <= kCandMax && ia >= 0 && ia < cnt_a && ib >= 0 && ib < cnt_b)) return false;  // workgroup-uniform
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave < 2) {
        const int cnt = wave ? cnt_b : cnt_a, pos = wave ? ib : ia;
        x = sort64(lane < cnt ? x : INFINITY);
        const bool in = wave ? lane <= pos : (lane >= pos && lane < cnt);
        const double n = static_cast<double>(wave ? pos + 1 : cnt - pos);
        const double mean = wave_sum(in ? x : 0.0) / n;
        const double q = wave_sum(in ? (x - mean) * (x - mean) : 0.0);
        const double stat = __shfl(x, pos, 64);
        if (lane == 0) { sel[wave][0] = stat; sel[wave][1] = n; sel[wave][2] = mean; sel[wave][3] = q; }
    } else if (wave == 2) {
        double sn = 0.0, sm = 0.0;
#pragma unroll
        for (int u = 0; u < kPartLane; ++u) { sn += pn[u]; sm += pn[u] * pm[u]; }
        const double n0 = wave_sum(sn);
        const double mean0 = wave_sum(sm) / n0;
        double sq = 0.0;
#pragma unroll
        for (int u = 0; u < kPartLane; ++u) sq += pq[u] + pn[u] * (pm[u] - mean0) * (pm[u] - mean0);
        const double q0 = wave_sum(sq);
        if (lane == 0) { sel[2][0] = 0.0; sel[2][1] = n0; sel[2][2] = mean0; sel[2][3] = q0; }
    }
    lds_barrier();
    va = sel[0][0]; vb = sel[1][0];
    const double n1 = sel[0][1], mean1 = sel[0][2], q1 = sel[0][3], n2 = sel[1][1], mean2 = sel[1][2], q2 = sel[1][3];
    double n0 = sel[2][1], mean0 = sel[2][2], q0 = sel[2][3];
    if (!(n0 > 0.0)) { mean0 = 0.0; q0 = 0.0; }  // no value between the windows: the mean above divided by zero
    const double n = n0 + n1 + n2;
    const double mean = (n0 * mean0 + n1 * mean1 + n2 * mean2) / n;
    const double m2 = (q0 + n0 * (mean0 - mean) * (mean0 - mean)) + (q1 + n1 * (mean1 - mean) * (mean1 - mean)) +
                      (q2 + n2 * (mean2 - mean) * (mean2 - mean));
    se = sqrt(m2 / (n - 1.0));
    return static_cast<int>(n) == a.b0 - a.a0 + 1;
}

template <bool COH>
__device__ __forceinline__ bool slice_std(const IterArgs &a, const double *cand, int npart, int below_a, int below_b, int cnt_a, int cnt_b,
                                          double (*sel)[4], double *red, double &se, double &va, double &vb)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double x = 0.0, pn[kPartPer] = {0.0}, pm[kPartPer] = {0.0}, pq[kPartPer] = {0.0};
    if (wave < 2 && lane < (wave ? cnt_b : cnt_a) && lane < kCandMax) x = ldc<COH>(cand + wave * kCandMax + lane);
#pragma unroll
    for (int u = 0; u < kPartPer; ++u) {
        const int w = threadIdx.x + 256 * u;
        if (w < npart) { pn[u] = ldc<COH>(a.part + 3 * w); pm[u] = ldc<COH>(a.part + 3 * w + 1); pq[u] = ldc<COH>(a.part + 3 * w + 2); }
    }
    return slice_std_vals(a, x, pn, pm, pq, below_a, below_b, cnt_a, cnt_b, sel, red, se, va, vb);
}

// ---------------------------------------------------------------------------
// The light passes as TWO launches per pass.  A pass has two grid-wide dependencies (all delta1 -> se; all BH ranks ->
// the cut), so two launches is the least without a grid barrier: the mask step of pass t (cut, inds, change list, loop
// control) moves to the front of the launch that derives pass t + 1, and every workgroup does it for ALL genes by
// itself -- the cut from the histogram (a few thousand bins), the new mask from the G ranks (4 G bytes from L2), the
// change list (<= 128 entries) into LDS -- so nothing of it crosses workgroups.  The loop state is a log (LightState,
// reo_internal.h): launch b reads slot b - 1 and the outputs of the launches before it, and only its workgroup 0 writes
// the record of slot b; every pass has its own counters (zeroed with the log), the histogram alternates by launch parity.  The order of the change
// list differs between workgroups (LDS atomics); it only feeds integer sums.  A batch ends with the TAIL form, which
// does the last mask step and writes the state back to IterState for the sorting path and the host.
// The BH cut from the first four 2048-bin tiles of the histogram, already in registers (hv[e]: bins (e * 256 + thread) * 8
// ...+7): bh_cut for a histogram whose finite ranks number at most 8192.  Bins past the bound may hold counts; they
// cannot qualify (H never exceeds the number of finite ranks), so no masking is needed.
// tile0, carry: the four tiles are tiles tile0 .. tile0 + 3 of the histogram and `carry` ranks lie in the tiles before them
// (updated to include these four); a trailing barrier lets the call be repeated.
// The two-launch light passes keep their histogram RELATIVE to the last cut: bin 0 is rank hist_first(kstar) + 1, and the genes
// whose rank is not above hist_first are only counted (LightCnt::nlow).  The cut moves by a handful of ranks per pass, so the
// bins that matter -- from a little below the last cut to the number of finite ranks -- fit ONE tile of 2 048 bins: kl_head
// asks for 64 KB of partial histograms at its start instead of 128, and its scan has a quarter of the work.  A cut that has
// dropped below hist_first cannot be found this way: the pass then goes to the sorting path (like a lost quantile window).
// below: ranks under the last cut that still get bins (IterArgs::hist_below; 256 unless a test asks for less)
__device__ __forceinline__ int hist_first(int kstar, int below) { return kstar > below ? kstar - below : 0; }

template <int NT = 4>   // NT: the tiles of hv that are in use
__device__ __forceinline__ int bh_cut4(const int (&hv)[4][8], int G, int tile0, int &carry_io, int roff = 0)
{
    __shared__ __attribute__((aligned(16))) int wsum[4][4];
    __shared__ int wbest[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int s[4], inc[4];
#pragma unroll
    for (int e = 0; e < NT; ++e) {
        s[e] = ((hv[e][0] + hv[e][1]) + (hv[e][2] + hv[e][3])) + ((hv[e][4] + hv[e][5]) + (hv[e][6] + hv[e][7]));
        inc[e] = s[e];
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1)
#pragma unroll
        for (int e = 0; e < NT; ++e)
        { const int u = __shfl_up(inc[e], o, 64); if (lane >= o) inc[e] += u; }
    if (lane == 63)
#pragma unroll
        for (int e = 0; e < NT; ++e) wsum[e][wave] = inc[e];
    lds_barrier();
    int best = 0, carry = carry_io;
#pragma unroll
    for (int e = 0; e < NT; ++e) {
        const int4 ws = *reinterpret_cast<const int4 *>(wsum[e]);
        int run = carry + inc[e] - s[e] + (wave > 0 ? ws.x : 0) + (wave > 1 ? ws.y : 0) + (wave > 2 ? ws.z : 0);
        const int r0 = roff + ((tile0 + e) * 256 + threadIdx.x) * 8 + 1;   // rank of the first of this thread's eight bins
#pragma unroll
        for (int u = 0; u < 8; ++u) { run += hv[e][u]; if (r0 + u <= G && run >= r0 + u) best = max(best, r0 + u); }
        carry += ws.x + ws.y + ws.z + ws.w;
    }
    carry_io = carry;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int u = __shfl_xor(best, o, 64); best = u > best ? u : best; }
    if (lane == 0) wbest[wave] = best;
    lds_barrier();
    const int k = max(max(wbest[0], wbest[1]), max(wbest[2], wbest[3]));
    lds_barrier();  // wsum / wbest may be rewritten by the next call
    return k;
}

// Touch every 64-byte line of the kernel's argument block at once.  The compiler loads arguments where they are first
// used, one scalar-cache miss after the other (measured: ~5 us from kernel start to the first vector load of kl_head);
// after this they all hit.
template <int BYTES>
__device__ __forceinline__ void warm_kernargs()
{
    const uint32_t *ka = (const uint32_t *)__builtin_amdgcn_kernarg_segment_ptr();  // (address-space cast: C style)
    uint32_t t[(BYTES + 63) / 64];
#pragma unroll
    for (int q = 0; q < (BYTES + 63) / 64; ++q) t[q] = ka[16 * q];
#pragma unroll
    for (int q = 0; q < (BYTES + 63) / 64; ++q) asm volatile("" ::"s"(t[q]));
}

constexpr int kListPre = 5;   // rounds of 256 list entries requested at kernel start (80 workgroups' parts)
constexpr int kHeadPre = 20;  // rows of 256 genes per wave whose BH ranks are requested at kernel start (20 000 genes: all of them)

// mrank of the two-launch form.  Layout: the word of gene 256 R + 64 c + l (row R, wave c, lane l of kl_rank) is stored at
// 256 R + 4 l + c, so that lane l of any wave gets the words of genes (c = 0..3, l) of a row with one coalesced 16-byte
// load.  Word: the BH rank (G + 2: the gene fails the p-value criterion) with the gene's CURRENT mask bit in bit 31, so
// that the mask step compares per lane, with no mask words and no cross-lane traffic; padding slots of the last row
// hold 0 (rank 0 is always inside the cut: new bit 0 = old bit).
__device__ __forceinline__ int mrank_slot(int i) { return (i & ~255) + 4 * (i & 63) + ((i >> 6) & 3); }

// the XCD this wave runs on (0..7)
__device__ __forceinline__ unsigned xcc_id()
{
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
    return x & (kHistParts - 1);
}

// add to one of the kSpread parts of a counter that its readers sum: by XCD with an atomic that stays in the XCD's L2 when
// that has been checked (reo_create), else by workgroup with a device-coherent atomic
__device__ __forceinline__ void spread_add(int32_t (*parts)[32], int v, bool xcc_local)
{
    if (xcc_local) __hip_atomic_fetch_add(&parts[xcc_id()][0], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else atomicAdd(&parts[blockIdx.x & (kSpread - 1)][0], v);
}

// reo_create's check of what the per-XCD histograms rest on: every wave adds 1 to 64 counters of its XCD's partial, 16
// times, with atomics that need not be coherent beyond the XCD's L2, and notes its XCD.  Afterwards every partial must
// hold exactly 16 x (waves that named it) in each counter.
// (the production pattern is two launches: kl_head zeroes all eight partials with plain stores from whatever XCD its
//  workgroups run on, kl_rank then adds with L2-local atomics in the next launch -- the self-test does the same)
__global__ __launch_bounds__(256) void k_xcc_selftest_zero(int32_t *part, int32_t *waves_of)
{
    // every workgroup stores into all partials (values from different XCDs race benignly: all zeros), after dirtying its
    // own XCD's copy of the lines so that a stale line would be seen
    for (int i = threadIdx.x; i < kHistParts * 64; i += 256) part[i] = 0;
    if (threadIdx.x < kHistParts) waves_of[threadIdx.x] = 0;
}

__global__ __launch_bounds__(256) void k_xcc_selftest(int32_t *part, int32_t *waves_of)
{
    const unsigned x = xcc_id();
    const int lane = threadIdx.x & 63;
    for (int k = 0; k < 16; ++k) __hip_atomic_fetch_add(&part[x * 64 + lane], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (lane == 0) atomicAdd(&waves_of[x], 1);
}

// lane l's four words of row R of mrank (COH: coherent 8-byte loads, for the persistent kernel)
template <bool COH>
__device__ __forceinline__ int4 load_rank_row(const int32_t *mrank, int R, int lane)
{
    if (!COH) return reinterpret_cast<const int4 *>(mrank)[R * 64 + lane];
    const unsigned long long *p = reinterpret_cast<const unsigned long long *>(mrank) + (static_cast<size_t>(R) * 64 + lane) * 2;
    const unsigned long long lo = ldc<true>(p), hi = ldc<true>(p + 1);
    return make_int4(static_cast<int>(lo), static_cast<int>(lo >> 32), static_cast<int>(hi), static_cast<int>(hi >> 32));
}

// The mask step's look at the BH ranks: which genes change their mask bit under the cut kstar.  mv: this lane's words of
// the wave's first kHeadPre rows (row wave + 4 q; zeros past the last row); further rows, and all rows again in the
// crowded case, are read from mrank.  Appends (gene << 1 | new bit) to dl (at most kDeltaMax entries kept), counts the
// changes in *s_n and added - removed in *s_nn (both zeroed by the caller, behind a barrier).  Whole workgroup.
// One subtract per gene: with u = its word (rank | current bit << 31) the bit changes iff kstar < u <= kstar + 2^31,
// i.e. iff u - (kstar + 1), as a signed number, is >= 0 (rank > kstar with the bit clear, or rank <= kstar with it set).
// Changes are rare (a handful per pass): every lane reduces its rows to "how many of my rows hold a change, and the
// last such row" without a branch -- a taken branch over a block of cold code costs an instruction-cache miss, and a
// chain of twenty of them was two microseconds -- and the changed genes of that one row are then noted by the wave
// together.  sum(inds) follows from the old count and the changes.
template <bool COH>
__device__ __forceinline__ void mask_scan(const int4 (&mv)[kHeadPre], int kstar, int nrow, const int32_t *mrank, uint32_t *dl, int *s_n, int *s_nn, bool reload_all = false)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t base = static_cast<uint32_t>(kstar) + 1u;
    int4 sel = make_int4(0, 0, 0, 0);
    int selq = 0, nflag = 0;
#pragma unroll
    for (int q = 0; q < kHeadPre; ++q) {  // (rows past the last one hold zeros: never a change)
        const int d0 = static_cast<int>(static_cast<uint32_t>(mv[q].x) - base), d1 = static_cast<int>(static_cast<uint32_t>(mv[q].y) - base);
        const int d2 = static_cast<int>(static_cast<uint32_t>(mv[q].z) - base), d3 = static_cast<int>(static_cast<uint32_t>(mv[q].w) - base);
        const bool f = max(max(d0, d1), max(d2, d3)) >= 0;
        sel.x = f ? mv[q].x : sel.x; sel.y = f ? mv[q].y : sel.y; sel.z = f ? mv[q].z : sel.z; sel.w = f ? mv[q].w : sel.w;
        selq = f ? q : selq;
        nflag += f ? 1 : 0;
    }
    auto note_changes = [&](bool on, int R, const int4 &m) {  // whole wave; lane l (if on): genes 256 R + 64 c + l, c = 0..3
        const uint32_t e[4] = {static_cast<uint32_t>(m.x), static_cast<uint32_t>(m.y), static_cast<uint32_t>(m.z), static_cast<uint32_t>(m.w)};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool chd = on && static_cast<int>(e[c] - base) >= 0;
            const unsigned long long cm = __ballot(chd);
            if (cm) {  // wave-uniform
                const uint32_t nb = (e[c] >> 31) ^ 1u;  // the new bit: inds = .!(pval <= pval_deg .& padj <= padj_deg), :417
                const unsigned long long am = __ballot(chd && nb);
                int pos = 0;
                if (lane == 0) { pos = atomicAdd(s_n, __popcll(cm)); atomicAdd(s_nn, 2 * __popcll(am) - __popcll(cm)); }
                pos = __shfl(pos, 0, 64);
                const int at = pos + __popcll(cm & ((1ULL << lane) - 1ULL));
                if (chd && at < kDeltaMax) dl[at] = (static_cast<uint32_t>(R * 256 + c * 64 + lane) << 1) | nb;
            }
        }
    };
    const bool crowded = reload_all || __ballot(nflag > 1) != 0;  // a lane with changes in two rows (about one pass in fifty): the wave reads its rows again
    if (!crowded && __ballot(nflag == 1)) note_changes(nflag == 1, wave + 4 * selq, sel);
#pragma unroll 1
    for (int R = crowded ? wave : wave + 4 * kHeadPre; R < nrow; R += 4) note_changes(true, R, load_rank_row<COH>(mrank, R, lane));
}

template <bool TAIL, bool LIST>
__global__ __launch_bounds__(256) void kl_head(IterArgs a, LightState *ls, int b)
{
    const int G = a.G, Gp = a.Gp;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ uint32_t dl[kDeltaMax];
    __shared__ int s_n, s_nn, s_fb;
    __shared__ uint4 dbuf[kDeltaMax * 8];
    __shared__ double red[256];
    warm_kernargs<sizeof(IterArgs) + 16>();
    if (!TAIL) STAMP(a, 8);
    const int pb = (b - 1) & 1;
    // ---- everything whose address does not depend on loaded data is requested first: the launch is a chain of
    // dependent round trips otherwise (measured: 22 us with the loads where they are used, most of it waiting)
    LightRec r;
    int bfail = 0, sig = 0, nlow = 0, cdiff = 0, cfound = 0;
    int hv[4][8];
    int4 hx[kHistParts][2];
    int4 mv[kHeadPre];
    int2 le[kListPre];
    int lcnt[kListPre];
    int own_m = 0;
    int4 r0 = make_int4(0, 0, 0, 0), r1 = make_int4(0, 0, 0, 0);
    double win[4];
    uint32_t cyc_w[kDeltaMax / 64];   // cycle watch (workgroup 0, wave 0): changed genes and their snapshot bytes
    uint8_t cyc_s[kDeltaMax / 64];
    int cyc_open = 0;                 // > 0: a comparison with the snapshot is under way, the period it would prove; -1: it cannot be made
    bool cyc_renew = false;
    const int nrow = (G + 255) >> 8;
    if (b > 0) {
        // the histogram of the BH ranks is kept as one partial histogram per XCD (kl_rank), relative to the last cut
        // (hist_first); the first tile of each (2 048 bins: enough unless the cut has more than 1 792 finite ranks above
        // it) is requested now and summed below
        const int32_t *hist = a.hist + static_cast<size_t>(pb) * kHistParts * a.hist_stride;
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int u = 0; u < 8; ++u) hv[e][u] = 0;
#pragma unroll
        for (int x = 0; x < kHistParts; ++x) {   // (the first tile of every partial: the histogram is relative to the last cut, see hist_first)
            const int4 *hp = reinterpret_cast<const int4 *>(hist + static_cast<size_t>(x) * a.hist_stride + threadIdx.x * 8);
            hx[x][0] = hp[0]; hx[x][1] = hp[1];
        }
#pragma unroll
        for (int q = 0; q < kHeadPre; ++q) {
            const int R = wave + 4 * q;
            mv[q] = make_int4(0, 0, 0, 0);
            if (!LIST && R < nrow) mv[q] = reinterpret_cast<const int4 *>(a.mrank)[R * 64 + lane];  // wave-uniform
        }
        if (LIST) {  // the genes near the cut, as listed by the launch before (kl_rank): entry e of workgroup e / kListCap
            const int32_t *cl = a.clist + static_cast<size_t>(pb) * kListStride;
#pragma unroll
            for (int q = 0; q < kListPre; ++q) {
                const int e = threadIdx.x + 256 * q, blk = e / kListCap;
                le[q] = make_int2(0, 0); lcnt[q] = 0;
                if (blk < nrow) { lcnt[q] = cl[blk]; le[q] = reinterpret_cast<const int2 *>(cl + kListWgs)[e]; }
            }
        }
        if (i < G) own_m = a.mrank[mrank_slot(i)];
    }
    if (!TAIL && i < G) {
        const int4 *o = reinterpret_cast<const int4 *>(a.raw + static_cast<size_t>(i) * kRaw);
        r0 = o[0]; r1 = o[1];
    }
    if (b > 0) {  // (after the vector loads: its values are needed in scalar registers, which waits for them)
        const LightSlot *ps = &ls->slot[b - 1];
        r = ps->rec; bfail = ps->bfail; cdiff = ps->cdiff; cfound = ps->cfound;
#pragma unroll
        for (int q = 0; q < kSpread; ++q) { sig += ps->lc.sig[q][0]; nlow += ps->lc.nlow[q][0]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) win[q] = ps->wnext[q];
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) win[q] = a.scal[1 + q];
    }
    if (!TAIL) STAMP(a, 0);
    int n = 0;           // entries of dl: the genes whose mask bit changes in front of the pass derived here
    bool inref = false;  // this thread's gene is in the reference set of that pass
    bool stepped = false;
    if (b == 0) {
        const IterState *st = a.st;  // no light launch writes IterState except the TAIL
        r.t = st->passes; r.nref = st->nref; r.nref_prev = st->nref_prev; r.done = st->done; r.need_full = st->need_full;
        r.raw_pass = st->raw_pass; r.ran = 0; r.dcnt = st->delta_cnt[r.t & 1]; r.kstar = -1;
        r.cper = st->cyc_period; r.cpow = st->cyc_pow; r.clam = st->cyc_lam; r.csel = st->cyc_sel; cdiff = st->cyc_ndiff;
        r.active = (!r.done && r.t < a.n_iter && !r.need_full) ? 1 : 0;
        if (r.active) {
            n = r.raw_pass == r.t ? 0 : min(r.dcnt, kDeltaMax);  // (need_full == 0 implies dcnt <= kDeltaMax)
            if (static_cast<int>(threadIdx.x) < n) dl[threadIdx.x] = a.delta_list[static_cast<size_t>(r.t & 1) * Gp + threadIdx.x];
            inref = i < G && a.refbytes[r.t & 1][i] != 0;
            lds_barrier();
        }
    } else if (r.active && bfail) {
        r.active = 0; r.need_full = 1;  // pass r.t lost a quantile window: the sorting path redoes it (its tallies are in place)
    } else if (r.active && cfound) {
        r.active = 0; r.cper = cfound;  // the mask in front of pass r.t equals the one cfound passes earlier: the host takes it from here (cycle watch)
    } else if (r.active) {
        // ---- the mask step of pass r.t (:413-424)
        const int t = r.t, cur = t & 1, nxt = cur ^ 1;
        const int off = hist_first(r.kstar, a.hist_below), srel = sig - off;   // bins are ranks off + 1 ..; finite ranks above off: at most srel
        {
            const int32_t *hist = a.hist + static_cast<size_t>(pb) * kHistParts * a.hist_stride;
#pragma unroll
            for (int x = 0; x < kHistParts; ++x) {
                hv[0][0] += hx[x][0].x; hv[0][1] += hx[x][0].y; hv[0][2] += hx[x][0].z; hv[0][3] += hx[x][0].w;
                hv[0][4] += hx[x][1].x; hv[0][5] += hx[x][1].y; hv[0][6] += hx[x][1].z; hv[0][7] += hx[x][1].w;
            }
            if (srel > 2048) {  // (workgroup-uniform; not at config 3) the other three tiles of the first four, now
#pragma unroll 1
                for (int x = 0; x < kHistParts; ++x)
#pragma unroll
                    for (int e = 1; e < 4; ++e) {
                        const int4 *hp = reinterpret_cast<const int4 *>(hist + static_cast<size_t>(x) * a.hist_stride + (e * 256 + threadIdx.x) * 8);
                        const int4 h0 = hp[0], h1 = hp[1];
                        hv[e][0] += h0.x; hv[e][1] += h0.y; hv[e][2] += h0.z; hv[e][3] += h0.w; hv[e][4] += h1.x; hv[e][5] += h1.y; hv[e][6] += h1.z; hv[e][7] += h1.w;
                    }
            }
        }
        int carry = nlow;   // the genes whose rank is not above `off`
        int kstar = srel > 2048 ? bh_cut4(hv, G, 0, carry, off) : bh_cut4<1>(hv, G, 0, carry, off);   // (workgroup-uniform)
        if (srel > 8192) {  // (workgroup-uniform; rare: more than 8192 genes inside or near the cut) the tiles after the first four, four at a time
            const int32_t *hist = a.hist + static_cast<size_t>(pb) * kHistParts * a.hist_stride;
            const int ntile = (min(G - off, srel) + 2047) / 2048;
#pragma unroll 1
            for (int tile0 = 4; tile0 < ntile; tile0 += 4) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int u = 0; u < 8; ++u) hv[e][u] = 0;
#pragma unroll 1
                for (int x = 0; x < kHistParts; ++x)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if ((tile0 + e) * 2048 >= a.hist_stride) continue;  // (past the histogram: no such ranks)
                        const int4 *hp = reinterpret_cast<const int4 *>(hist + static_cast<size_t>(x) * a.hist_stride + ((tile0 + e) * 256 + threadIdx.x) * 8);
                        const int4 h0 = hp[0], h1 = hp[1];
                        hv[e][0] += h0.x; hv[e][1] += h0.y; hv[e][2] += h0.z; hv[e][3] += h0.w; hv[e][4] += h1.x; hv[e][5] += h1.y; hv[e][6] += h1.z; hv[e][7] += h1.w;
                    }
                kstar = max(kstar, bh_cut4(hv, G, tile0, carry, off));
            }
        }
        const bool cut_lost = off > 0 && kstar == 0;   // (the same in every workgroup) the cut dropped below the histogram's first bin
        if (!TAIL) STAMP(a, 1);
        if (cut_lost) {
            r.active = 0; r.need_full = 1;  // the sorting path finds the cut of pass r.t (its tallies are in place)
        } else {
        if (threadIdx.x == 0) { s_n = 0; s_nn = 0; s_fb = 0; }
        lds_barrier();
        if (LIST) {
            // every gene whose bit can change is in the list if the cut stayed within `band` ranks of the cut the list was
            // made for (kl_rank) and no workgroup's part overflowed; otherwise all ranks are read (mask_scan)
            const uint32_t base = static_cast<uint32_t>(kstar) + 1u;
            bool bad = r.kstar < 0 || kstar < r.kstar - a.band || kstar > r.kstar + a.band;  // (the same in every workgroup)
            auto note = [&](int cnt, int2 ent, int e) {
                bad = bad || cnt > kListCap;
                const bool chd = (e % kListCap) < cnt && static_cast<int>(static_cast<uint32_t>(ent.x) - base) >= 0;
                const unsigned long long cm = __ballot(chd);
                if (cm) {  // wave-uniform
                    const uint32_t nb = (static_cast<uint32_t>(ent.x) >> 31) ^ 1u;
                    const unsigned long long am = __ballot(chd && nb);
                    int pos = 0;
                    if (lane == 0) { pos = atomicAdd(&s_n, __popcll(cm)); atomicAdd(&s_nn, 2 * __popcll(am) - __popcll(cm)); }
                    pos = __shfl(pos, 0, 64);
                    const int at = pos + __popcll(cm & ((1ULL << lane) - 1ULL));
                    if (chd && at < kDeltaMax) dl[at] = (static_cast<uint32_t>(ent.y) << 1) | nb;
                }
            };
#pragma unroll
            for (int q = 0; q < kListPre; ++q) note(lcnt[q], le[q], threadIdx.x + 256 * q);
            const int32_t *cl = a.clist + static_cast<size_t>(pb) * kListStride;
#pragma unroll 1
            for (int e = threadIdx.x + 256 * kListPre; e < nrow * kListCap; e += 256) note(cl[e / kListCap], reinterpret_cast<const int2 *>(cl + kListWgs)[e], e);
            if (__ballot(bad) && lane == 0) s_fb = 1;
            lds_barrier();
            if (s_fb) {  // workgroup-uniform (and, the inputs being the same, the same in every workgroup)
                lds_barrier();
                if (threadIdx.x == 0) { s_n = 0; s_nn = 0; }
                lds_barrier();
                mask_scan<false>(mv, kstar, nrow, a.mrank, dl, &s_n, &s_nn, true);
            }
        } else {
            mask_scan<false>(mv, kstar, nrow, a.mrank, dl, &s_n, &s_nn);
        }
        const bool ind = i < G && static_cast<int>(static_cast<uint32_t>(own_m) & 0x7FFFFFFFu) > kstar;
        if (i < Gp) a.refbytes[nxt][i] = ind ? 1 : 0;
        const unsigned long long mk = __ballot(ind);
        if (lane == 0 && i < Gp) { a.refbits[nxt][i >> 5] = static_cast<uint32_t>(mk); a.refbits[nxt][(i >> 5) + 1] = static_cast<uint32_t>(mk >> 32); }
        lds_barrier();
        const int chg = s_n, nn = r.nref + s_nn;  // sum(inds), :417-418: the old mask's count (r.nref) + added - removed
        if (!TAIL) STAMP(a, 2);
        if (blockIdx.x == 0) {
            if (threadIdx.x == 0) { a.trace[2 * t] = G - nn; a.trace[2 * t + 1] = nn; }
            if (static_cast<int>(threadIdx.x) < min(chg, kDeltaMax)) a.delta_list[static_cast<size_t>(nxt) * Gp + threadIdx.x] = dl[threadIdx.x];
        }
        r.nref_prev = r.nref;
        const int diff = r.nref - nn;
        if ((diff < 0 ? -diff : diff) < a.n_conv) r.done = 1;  // :419-422
        else r.nref = nn;                                      // :423-424
        r.t = t + 1; r.ran = 1; r.dcnt = chg; r.kstar = kstar;
        r.need_full = chg > kDeltaMax ? 1 : 0;
        r.active = (!r.done && r.t < a.n_iter && !r.need_full) ? 1 : 0;
        inref = ind;
        n = min(chg, kDeltaMax);
        stepped = true;
        // ---- cycle watch.  The mask of the next pass is a function of this pass's mask alone (the table is fixed), so a mask
        // that returns makes everything after it periodic: Brent's search keeps ONE snapshot of the mask, renewed after 1, 2, 4, ..
        // steps, and the number of bits in which the current mask differs from it -- updated from the list of changed genes, which
        // every mask step has anyway (workgroup 0, wave 0; the snapshot bytes are asked for here and looked at when the launch has
        // nothing else left to do).  No differing bit <=> the masks are EQUAL: exact, no hashing.  What follows from a period: api.hip.
        if (r.cper == 0) {
            const bool fresh = r.cpow == 0;   // no snapshot to compare with: this mask becomes the first
            if (!fresh) r.clam += 1;
            if (!fresh && blockIdx.x == 0 && wave == 0) {
                const uint8_t *snap = a.snap + static_cast<size_t>(r.csel) * Gp;
#pragma unroll
                for (int q = 0; q < kDeltaMax / 64; ++q) {
                    const int e = lane + 64 * q;
                    cyc_w[q] = e < n ? dl[e] : 0xFFFFFFFFu;
                    cyc_s[q] = e < n ? snap[cyc_w[q] >> 1] : 0;
                }
                cyc_open = chg <= kDeltaMax ? r.clam : -1;   // (more changes than the list holds: the pass after this one sorts, which drops the snapshot)
            }
            if (fresh || r.clam == r.cpow) {  // (workgroup-uniform) this mask is the new snapshot -- in the OTHER buffer: workgroup 0 may still be reading the old one
                r.cpow = fresh ? 1 : 2 * r.cpow; r.clam = 0; r.csel ^= 1;
                if (i < Gp) a.snap[static_cast<size_t>(r.csel) * Gp + i] = ind ? 1 : 0;
                cyc_renew = true;
            }
        }
        }
    }
    // cycle watch, the end of it (workgroup 0, wave 0): the differing bits after this launch's mask step, and the period if there are none
    auto cyc_close = [&]() {
        if (cyc_open > 0) {
            int d = 0;
#pragma unroll
            for (int q = 0; q < kDeltaMax / 64; ++q)   // a changed gene agreed with the snapshot before the step iff its new bit differs from the snapshot's
                if (cyc_w[q] != 0xFFFFFFFFu) d += ((cyc_s[q] != 0) == ((cyc_w[q] & 1u) != 0)) ? -1 : 1;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
            cdiff += d;
            cfound = cdiff == 0 ? cyc_open : 0;
        } else if (cyc_open < 0) {
            cdiff = 0x40000000; cfound = 0;   // unknown (never zero again before the snapshot is renewed)
        } else {
            cfound = stepped ? 0 : cfound;
        }
        if (cyc_renew) cdiff = 0;
    };
    if (TAIL) {
        if (blockIdx.x == 0 && wave == 0) {
            cyc_close();
            if (cfound) r.cper = cfound;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            IterState *st = a.st;
            st->cyc_period = r.cper; st->cyc_pow = r.cpow; st->cyc_lam = r.clam; st->cyc_sel = r.csel; st->cyc_ndiff = cdiff;
            st->passes = r.t; st->nref = r.nref; st->nref_prev = r.nref_prev; st->done = r.done; st->need_full = r.need_full;
            st->i_iter = r.t - (r.done ? 1 : 0);
            st->raw_pass = r.raw_pass;
            st->delta_cnt[r.t & 1] = r.dcnt;
            if (r.ran) st->last_full = 0;
            if (stepped) { a.scal[1] = win[0]; a.scal[2] = win[1]; a.scal[3] = win[2]; a.scal[4] = win[3]; }
            if (a.host_st) {
                IterState *h = a.host_st;
                h->passes = r.t; h->done = r.done; h->need_full = r.need_full; h->delta_cnt[r.t & 1] = r.dcnt;
                h->cyc_period = r.cper;
                if (r.ran) h->last_full = 0;
            }
        }
        return;
    }
    if (r.active) r.raw_pass = r.t;  // the tallies of pass r.t are made below
    if (blockIdx.x == 0 && threadIdx.x == 0) ls->slot[b].rec = r;
    if (!r.active || static_cast<int>(blockIdx.x) * 256 >= G) {
        if (blockIdx.x == 0 && wave == 0) { cyc_close(); if (lane == 0) { ls->slot[b].cdiff = cdiff; ls->slot[b].cfound = cfound; } }
        return;
    }
    // ---- pass r.t: tallies from the changed rows, delta1, window bookkeeping
    const int pbuf = b & 1;
    LightCnt *lc = &ls->slot[b].lc;
    const double wa_lo = win[0], wa_hi = win[1], wb_lo = win[2], wb_hi = win[3];
    double v = 0.0;
    bool inner = false, inA = false, inB = false, belowA = false, belowB = false;
    if (n) {  // (workgroup-uniform)
        int d[kRaw];
        delta_counts_block(a.table, a.Wp, dl, n, d, dbuf);
        if (!TAIL) STAMP(a, 3);
        if (i < G) {
            r0.x += d[0]; r0.y += d[1]; r0.z += d[2]; r0.w += d[3];
            r1.x += d[4]; r1.y += d[5]; r1.z += d[6]; r1.w += d[7];
            int4 *o = reinterpret_cast<int4 *>(a.raw + static_cast<size_t>(i) * kRaw);
            o[0] = r0; o[1] = r1;
        }
    }
    if (i < G) {
        int32_t c[9];
        const bool tok = tallies_from(r0, r1, r.nref - (inref ? 1 : 0), c);  // the diagonal is never set (:363,385)
        double out[5] = {1.0, 0.0, 0.0, 0.0, 0.0};
        if (tok) mccullagh3<false>(c, out);
        v = out[1];
        if (!tok || !(fabs(v) < INFINITY)) { v = 0.0; raise_fault(a, kFaultTallies); }
        if (!TAIL) STAMP(a, 4);
        a.result[11 * static_cast<size_t>(G) + i] = v;
#pragma unroll
        for (int x = 0; x < kHistParts; ++x) a.hist[(static_cast<size_t>(pbuf) * kHistParts + x) * a.hist_stride + i] = 0;  // this launch parity's histograms: last read two launches ago
        belowA = v < wa_lo; inA = !belowA && v <= wa_hi;
        belowB = v < wb_lo; inB = !belowB && v <= wb_hi;
        inner = v > wa_hi && v < wb_lo;
    }
    if (!TAIL) STAMP(a, 5);
    if (G <= 65535) {
        // the moments of the values between the windows, per WAVE (wave sums only: no barrier; kl_rank combines 4 x as many
        // partials, which its third wave holds sixteen to a lane anyway)
        const double nb = static_cast<double>(__popcll(__ballot(inner))), sum = wave_sum(inner ? v : 0.0);
        const double mean = nb > 0.0 ? sum / nb : 0.0;
        const double m2 = wave_sum(inner ? (v - mean) * (v - mean) : 0.0);
        if (lane == 0) { double *pp = a.part + 3 * (4 * blockIdx.x + wave); pp[0] = nb; pp[1] = mean; pp[2] = m2; }
    } else {   // (above 65 535 genes four partials per workgroup would be more than kl_rank's third wave can hold: one per workgroup)
        double nb, sum;
        block_sum2_256(inner ? 1.0 : 0.0, inner ? v : 0.0, red, nb, sum);
        const double mean = nb > 0.0 ? sum / nb : 0.0;
        const double m2 = block_sum_256(inner ? (v - mean) * (v - mean) : 0.0, red);
        if (threadIdx.x == 0) { a.part[3 * blockIdx.x] = nb; a.part[3 * blockIdx.x + 1] = mean; a.part[3 * blockIdx.x + 2] = m2; }
    }
    if (!TAIL) STAMP(a, 6);
    const unsigned long long mA = __ballot(inA), mB = __ballot(inB), bA = __ballot(belowA), bB = __ballot(belowB);
    int baseA = 0, baseB = 0;
    if (lane == 0) {
        if (mA) baseA = atomicAdd(&lc->cnt_a, __popcll(mA));
        if (mB) baseB = atomicAdd(&lc->cnt_b, __popcll(mB));
        // the counts below the windows, per wave (no workgroup sum, no barrier at the end of the launch: the adds stay in the XCD's L2)
        if (bA) spread_add(lc->below_a, __popcll(bA), a.xcc_local);
        if (bB) spread_add(lc->below_b, __popcll(bB), a.xcc_local);
    }
    baseA = __shfl(baseA, 0, 64); baseB = __shfl(baseB, 0, 64);
    const unsigned long long lt = (1ULL << lane) - 1ULL;
    if (inA) { const int at = baseA + __popcll(mA & lt); if (at < kCandMax) a.cand[at] = v; }
    if (inB) { const int at = baseB + __popcll(mB & lt); if (at < kCandMax) a.cand[kCandMax + at] = v; }
    if (!TAIL) STAMP(a, 7);
    if (blockIdx.x == 0 && wave == 0) { cyc_close(); if (lane == 0) { ls->slot[b].cdiff = cdiff; ls->slot[b].cfound = cfound; } }
}

// second launch of a two-launch light pass: every workgroup finishes the selection for itself (slice_std), then p-values
// (:412), BH ranks and their histogram for its genes.  A rank is stored as "never" (G + 2) when the gene fails the
// p-value criterion of :417, so that the next launch's mask step needs the ranks only.  Like kl_head it asks for all its
// inputs before it looks at any of them.
__global__ __launch_bounds__(256) void kl_rank(IterArgs a, LightState *ls, int b)
{
    warm_kernargs<sizeof(IterArgs) + 16>();
    STAMP(a, 19);
    const int G = a.G, pbuf = b & 1;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool live = i < G;
    LightSlot *sl = &ls->slot[b];
    LightCnt *lc = &sl->lc;
    const int active = sl->rec.active, rec_t = sl->rec.t, rec_kstar = sl->rec.kstar;
    const int cnt_a = lc->cnt_a, cnt_b = lc->cnt_b;
    int below_a = 0, below_b = 0;
#pragma unroll
    for (int q = 0; q < kSpread; ++q) { below_a += lc->below_a[q][0]; below_b += lc->below_b[q][0]; }
    const double x = wave < 2 ? a.cand[wave * kCandMax + lane] : 0.0;  // (every slot of cand exists; slots past the count are ignored)
    double pn[kPartLane], pm[kPartLane], pq[kPartLane];   // wave 2: lane l holds the moments of workgroups l, l + 64, ... (slice_std_split)
#pragma unroll
    for (int u = 0; u < kPartLane; ++u) {
        const int w = lane + 64 * u;
        pn[u] = pm[u] = pq[u] = 0.0;
        if (wave == 2 && w < (G <= 65535 ? 4 : 1) * ((G + 255) / 256)) { pn[u] = a.part[3 * w]; pm[u] = a.part[3 * w + 1]; pq[u] = a.part[3 * w + 2]; }   // (kl_head: a partial per wave up to 65 535 genes, per workgroup above)
    }
    const double wd0 = a.scal[5], wd1 = a.scal[6], wd2 = a.scal[7], wd3 = a.scal[8];
    const double d1 = live ? a.result[11 * static_cast<size_t>(G) + i] : 0.0;
    // this gene's mask bit, both parities (which one counts is in the record, not known yet): asked for here because a load
    // issued behind the stores further down would wait for them to be acknowledged
    const uint8_t ob_e = live ? a.refbytes[0][i] : 0, ob_o = live ? a.refbytes[1][i] : 0;
    if (!active) return;
    STAMP(a, 20);
    __shared__ double sel[3][4];
    double se = 0.0, va = 0.0, vb = 0.0;
    bool ok = slice_std_split(a, x, pn, pm, pq, below_a, below_b, cnt_a, cnt_b, sel, se, va, vb);
    ok = ok && (va + wd1 < vb - wd2);
    STAMP(a, 21);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (ok) {
            a.scal[0] = se;
            sl->wnext[0] = va - wd0; sl->wnext[1] = va + wd1; sl->wnext[2] = vb - wd2; sl->wnext[3] = vb + wd3;
        } else {
            sl->bfail = 1;
        }
    }
    if (!ok) return;
    const double p = live ? normal_p(d1, se) : 1.0;
    if (live) a.result[i] = p;
    const int m = live ? bh_rank(p, G, a.padj_deg) : G + 1;
    STAMP(a, 22);
    {   // the histogram first: these atomics are performed at the memory side (they must be coherent across the XCDs) and
        // the launch cannot end before the last of them has been -- about 4 us when they were the kernel's last
        // instructions; issued here, the rest of the kernel runs in their shadow
        // one partial histogram per XCD, updated with atomics that need not be coherent beyond the XCD's own L2 (only
        // workgroups running on this XCD touch this partial): an atomic that has to be coherent across the XCDs is
        // performed at the memory side, and 3 500 of those kept the launch alive for 4.5 us
        // the bins are relative to the last cut (hist_first): ranks up to `off` are only counted
        const int off = hist_first(rec_kstar, a.hist_below);
        const unsigned long long first = __ballot(m == 1 && off == 0), finite = __ballot(m <= G), low = __ballot(m <= off);
        const bool lead1 = m == 1 && off == 0 && lane == __ffsll(static_cast<long long>(first)) - 1;  // the hot bin: one add per wave
        const bool binned = m > off && m <= G && !(m == 1 && off == 0);
        if (a.xcc_local) {
            int32_t *hist = a.hist + (static_cast<size_t>(pbuf) * kHistParts + xcc_id()) * a.hist_stride;
            if (lead1) __hip_atomic_fetch_add(&hist[0], __popcll(first), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (binned) __hip_atomic_fetch_add(&hist[m - 1 - off], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {  // (self-test failed or switched off: device-coherent atomics into partial 0)
            int32_t *hist = a.hist + static_cast<size_t>(pbuf) * kHistParts * a.hist_stride;
            if (lead1) atomicAdd(&hist[0], __popcll(first));
            else if (binned) atomicAdd(&hist[m - 1 - off], 1);
        }
        if (lane == 0 && finite) spread_add(lc->sig, __popcll(finite), a.xcc_local);
        if (lane == 0 && low) spread_add(lc->nlow, __popcll(low), a.xcc_local);
    }
    {   // every slot of the row is written (the mask step reads whole rows), transposed through LDS so that each wave
        // stores whole cache lines: rows written as scattered 4-byte pieces came back slowly in the next launch
        __shared__ int32_t row[256];
        const uint32_t oldbit = ((rec_t & 1) ? ob_o : ob_e) != 0 ? 0x80000000u : 0u;
        row[4 * lane + wave] = live ? static_cast<int32_t>(static_cast<uint32_t>(p <= a.pval_deg ? m : G + 2) | oldbit) : 0;
        lds_barrier();
        a.mrank[blockIdx.x * 256 + threadIdx.x] = row[threadIdx.x];
    }
    if (rec_kstar >= 0) {
        // the genes whose mask bit can change at the next mask step if its cut lands within `band` ranks of the last one:
        // ranks inside the band, and bits that are wrong for every cut in it.  By workgroup, no global atomics.
        __shared__ int s_c;
        if (threadIdx.x == 0) s_c = 0;
        lds_barrier();
        const int kp = rec_kstar, rk = p <= a.pval_deg ? m : G + 2;
        const bool ob = ((rec_t & 1) ? ob_o : ob_e) != 0;
        const bool want = live && ((rk > kp - a.band && rk <= kp + a.band) || (rk > kp + a.band && !ob) || (rk <= kp - a.band && ob));
        const unsigned long long cm = __ballot(want);
        int pos = 0;
        if (cm && lane == 0) pos = atomicAdd(&s_c, __popcll(cm));
        pos = __shfl(pos, 0, 64) + __popcll(cm & ((1ULL << lane) - 1ULL));
        int32_t *cl = a.clist + static_cast<size_t>(pbuf) * kListStride;
        if (want && pos < kListCap)
            reinterpret_cast<int2 *>(cl + kListWgs)[blockIdx.x * kListCap + pos] = make_int2(static_cast<int>(static_cast<uint32_t>(rk) | (ob ? 0x80000000u : 0u)), i);
        lds_barrier();
        if (threadIdx.x == 0) cl[blockIdx.x] = min(s_c, kListCap + 1);
    }
    STAMP(a, 23);
}

// ---------------------------------------------------------------------------
// The light passes as ONE launch per pass (round 4; opt-in, REO_LIGHT=3: measured 24.0 us per pass against the 23.1 of the
// two-launch form -- the launch it saves is paid back by the exact p-values of the listed genes, which now sit on the
// critical path between se and the mask step; DESIGN.md has the marks).  A pass has two grid-wide dependencies
// (all delta1 -> se; all BH ranks -> the cut), and the two-launch form above pays a launch for each.  Here the second
// one is carried across the launch boundary TOGETHER with the first: launch b derives pass t (tallies, delta1, window
// bookkeeping for se(t)) and, without knowing se(t), files every gene's p-value between two brackets -- the p-values under
// se_lo = se(t-1) (1 - eta) and se_hi = se(t-1) (1 + eta); p is monotone in se, so whatever se(t) turns out to be inside
// that interval, p_lo <= p <= p_hi and, the step-up rank m being monotone in p, m_lo <= m <= m_hi.  With the cut of
// the pass before, k', and the band [k' - band, k' + band] a gene is then
//   * surely inside the cut's reach (m_hi <= k' - band): counted, by workgroup, into n_sure;
//   * surely outside (m_lo > k' + band);
//   * or LISTED with its delta1 (a few dozen genes: those the bracket or the band cannot decide, and those whose mask bit
//     contradicts the sure verdict),
// and the histogram of m_lo (an upper bound of the true H(r) = #{m <= r}) is built as before.  Launch b + 1 then, in every
// workgroup: se(t) from the windows (exact, as before); the check that it lies inside the bracket; the cut of the m_lo
// histogram, which bounds the true cut from above and must not leave the band; the EXACT p and m of the listed genes
// under se(t); H(r) = n_sure + #{listed: m <= r} for every r of the band -- exact there -- and with it the true cut
// k* = max{r : H(r) >= r}; the new mask bits of the listed genes (nobody else's can change) and the change list; then
// pass t + 1.  Any check that fails (se left the bracket, the cut left the band, a list overflowed) hands pass t to the
// sorting path, exactly like a lost quantile window.  eta follows the drift of se: four times the last relative change,
// between 2^-11 and 2^-6 (measured: the iteration settles into a short cycle with changes of 2e-4 .. 1.4e-3 per pass).
// Buffers that one launch both reads (the other workgroups' results of the launch before) and writes (this pass's) are
// double-buffered by launch parity (window members, block moments, lists) or come in three (the histogram: read, filled, cleared).
constexpr double kEtaMin = 1.0 / 2048.0, kEtaMax = 1.0 / 64.0, kEta0 = 1.0 / 128.0;

template <bool TAIL>
__global__ __launch_bounds__(256) void kl_one(IterArgs a, LightState *ls, int b)
{
    const int G = a.G, Gp = a.Gp;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool live = i < G;
    __shared__ uint32_t dl[kDeltaMax];
    __shared__ int s_n, s_nn, s_nl, s_bad, s_c;
    __shared__ uint4 dbuf[kDeltaMax * 8];
    __shared__ double red[256];
    __shared__ double sel[2][4];
    __shared__ int wcnt[4][2];
    __shared__ double lm_d[kOneListMax];     // listed genes of the mask step: delta1 ...
    __shared__ int lm_m[kOneListMax];        // ... exact BH rank (G + 1: none) ...
    __shared__ uint32_t lm_g[kOneListMax];   // ... gene | old bit << 31 | passes the p-value criterion << 30
    __shared__ int wbest[4];
    warm_kernargs<sizeof(IterArgs) + 16>();
    if (!TAIL) STAMP(a, 8);
    const int pb = (b + 1) & 1, pbuf = b & 1;              // parity of the launch before / of this one (lists, window members, block moments)
    const int hb_prev = (b + 2) % 3, hb = b % 3, hb_next = (b + 1) % 3;   // histograms: read (filled by the launch before), filled here, cleared here
    const int nrow = (G + 255) >> 8;
    // ---- everything whose address does not depend on loaded data is requested first
    LightRec r;
    int hv[4][8];
    int4 hx[kHistParts][2][2];
    int4 le[kListPre];
    int lcnt[kListPre];
    int own_state = 0;
    double d1prev = 0.0;
    int4 r0 = make_int4(0, 0, 0, 0), r1 = make_int4(0, 0, 0, 0);
    double win[4];
    int cnt_a = 0, cnt_b = 0, below_a = 0, below_b = 0, sig = 0, nsure = 0;
    double xw = 0.0, pn[kPartPer] = {0.0}, pm[kPartPer] = {0.0}, pq[kPartPer] = {0.0}, se_base_prev = 0.0, eta_prev = 0.0;   // (this form: at most 65 535 genes, one partial per thread)
    const double wd0 = a.scal[5], wd1 = a.scal[6], wd2 = a.scal[7], wd3 = a.scal[8];
    if (b > 0) {
        const int32_t *hist = a.hist + static_cast<size_t>(hb_prev) * kHistParts * a.hist_stride;
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int u = 0; u < 8; ++u) hv[e][u] = 0;
#pragma unroll
        for (int x = 0; x < kHistParts; ++x)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int4 *hp = reinterpret_cast<const int4 *>(hist + static_cast<size_t>(x) * a.hist_stride + (e * 256 + threadIdx.x) * 8);
                hx[x][e][0] = hp[0]; hx[x][e][1] = hp[1];
            }
        const int32_t *ol = a.olist + static_cast<size_t>(pb) * kOneStride;
#pragma unroll
        for (int q = 0; q < kListPre; ++q) {
            const int e = threadIdx.x + 256 * q, blk = e / kOneListCap;
            le[q] = make_int4(0, 0, 0, 0); lcnt[q] = 0;
            if (blk < nrow) { lcnt[q] = ol[blk]; le[q] = reinterpret_cast<const int4 *>(ol + 256)[e]; }
        }
        if (live) { own_state = a.mrank[i]; d1prev = a.result[11 * static_cast<size_t>(G) + i]; }
        if (wave < 2) xw = a.cand[pb * 2 * kCandMax + wave * kCandMax + lane];
        if (static_cast<int>(threadIdx.x) < nrow) { const double *pp = a.part + static_cast<size_t>(pb) * 768 + 3 * threadIdx.x; pn[0] = pp[0]; pm[0] = pp[1]; pq[0] = pp[2]; }
    }
    if (!TAIL && live) {
        const int4 *o = reinterpret_cast<const int4 *>(a.raw + static_cast<size_t>(i) * kRaw);
        r0 = o[0]; r1 = o[1];
    }
    if (b > 0) {
        const LightSlot *ps = &ls->slot[b - 1];
        r = ps->rec;
        cnt_a = ps->lc.cnt_a; cnt_b = ps->lc.cnt_b;
#pragma unroll
        for (int q = 0; q < kSpread; ++q) { below_a += ps->lc.below_a[q][0]; below_b += ps->lc.below_b[q][0]; sig += ps->lc.sig[q][0]; nsure += ps->lc.nsure[q][0]; }
        se_base_prev = ps->se_base; eta_prev = ps->eta;
    }
    if (!TAIL) STAMP(a, 0);
    int n = 0;           // entries of dl: the genes whose mask bit changes in front of the pass derived here
    bool inref = false;  // this thread's gene is in the reference set of that pass
    bool stepped = false;
    double se = 0.0, se_base = 0.0, eta = kEta0;
    if (b == 0) {
        const IterState *st = a.st;  // no light launch writes IterState except the TAIL
        r.t = st->passes; r.nref = st->nref; r.nref_prev = st->nref_prev; r.done = st->done; r.need_full = st->need_full;
        r.raw_pass = st->raw_pass; r.ran = 0; r.dcnt = st->delta_cnt[r.t & 1]; r.kstar = st->kstar;
        r.active = (!r.done && r.t < a.n_iter && !r.need_full && r.kstar >= 0) ? 1 : 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) win[q] = a.scal[1 + q];
        se_base = a.scal[0];
        if (r.active) {
            n = r.raw_pass == r.t ? 0 : min(r.dcnt, kDeltaMax);  // (need_full == 0 implies dcnt <= kDeltaMax)
            if (static_cast<int>(threadIdx.x) < n) dl[threadIdx.x] = a.delta_list[static_cast<size_t>(r.t & 1) * Gp + threadIdx.x];
            inref = live && a.refbytes[r.t & 1][i] != 0;
            lds_barrier();
        }
    } else if (r.active) {
        // ---- the end of pass r.t: se, the cut, the mask step (:409-424)
        const int t = r.t, cur = t & 1, nxt = cur ^ 1;
        double va = 0.0, vb = 0.0;
        int why = 0;   // which check sent the pass to the sorting path (diagnostics: REO_DEBUG_PASSES)
        bool ok = slice_std_vals(a, xw, pn, pm, pq, below_a, below_b, cnt_a, cnt_b, sel, red, se, va, vb);
        ok = ok && (va + wd1 < vb - wd2);
        if (!ok) why |= 1;
        if (!(fabs(se - se_base_prev) <= 0.5 * eta_prev * se_base_prev)) { ok = false; why |= 2; }   // se left the bracket of the launch before
        win[0] = va - wd0; win[1] = va + wd1; win[2] = vb - wd2; win[3] = vb + wd3;
        se_base = se;
        {   // the next bracket: eight times the last relative change of se, and never less than half the last bracket
            const double drift = se_base_prev > 0.0 ? fabs(se / se_base_prev - 1.0) * 8.0 : kEta0;
            eta = drift > 0.5 * eta_prev ? drift : 0.5 * eta_prev;
            eta = eta < kEtaMin ? kEtaMin : (eta > kEtaMax ? kEtaMax : eta);
        }
        if (!TAIL) STAMP(a, 1);
        // the cut of the m_lo histogram: an upper bound of the true cut
        const int32_t *hist = a.hist + static_cast<size_t>(hb_prev) * kHistParts * a.hist_stride;
#pragma unroll
        for (int x = 0; x < kHistParts; ++x)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                hv[e][0] += hx[x][e][0].x; hv[e][1] += hx[x][e][0].y; hv[e][2] += hx[x][e][0].z; hv[e][3] += hx[x][e][0].w;
                hv[e][4] += hx[x][e][1].x; hv[e][5] += hx[x][e][1].y; hv[e][6] += hx[x][e][1].z; hv[e][7] += hx[x][e][1].w;
            }
        if (sig > 4096) {  // (workgroup-uniform) the other two tiles, now
#pragma unroll 1
            for (int x = 0; x < kHistParts; ++x)
#pragma unroll
                for (int e = 2; e < 4; ++e) {
                    const int4 *hp = reinterpret_cast<const int4 *>(hist + static_cast<size_t>(x) * a.hist_stride + (e * 256 + threadIdx.x) * 8);
                    const int4 h0 = hp[0], h1 = hp[1];
                    hv[e][0] += h0.x; hv[e][1] += h0.y; hv[e][2] += h0.z; hv[e][3] += h0.w; hv[e][4] += h1.x; hv[e][5] += h1.y; hv[e][6] += h1.z; hv[e][7] += h1.w;
                }
        }
        int carry = 0;
        int k_hi = bh_cut4(hv, G, 0, carry);
        if (sig > 8192) {  // (workgroup-uniform; rare) the tiles after the first four, four at a time
            const int ntile = (min(G, sig) + 2047) / 2048;
#pragma unroll 1
            for (int tile0 = 4; tile0 < ntile; tile0 += 4) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int u = 0; u < 8; ++u) hv[e][u] = 0;
#pragma unroll 1
                for (int x = 0; x < kHistParts; ++x)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if ((tile0 + e) * 2048 >= a.hist_stride) continue;  // (past the histogram: no such ranks)
                        const int4 *hp = reinterpret_cast<const int4 *>(hist + static_cast<size_t>(x) * a.hist_stride + ((tile0 + e) * 256 + threadIdx.x) * 8);
                        const int4 h0 = hp[0], h1 = hp[1];
                        hv[e][0] += h0.x; hv[e][1] += h0.y; hv[e][2] += h0.z; hv[e][3] += h0.w; hv[e][4] += h1.x; hv[e][5] += h1.y; hv[e][6] += h1.z; hv[e][7] += h1.w;
                    }
                k_hi = max(k_hi, bh_cut4(hv, G, tile0, carry));
            }
        }
        const int K_in = r.kstar - a.band, K_out = min(r.kstar + a.band, G);   // the band the launch before sorted its genes by
        if (k_hi > K_out) { ok = false; why |= 4; }
        if (!TAIL) STAMP(a, 2);
        // the listed genes: gathered into LDS first (a few dozen entries scattered over the workgroups' parts), then ONE round of
        // exact p and m under se, an entry per thread
        if (threadIdx.x == 0) { s_n = 0; s_nn = 0; s_nl = 0; s_bad = 0; }
        lds_barrier();
        {
            bool bad = false;
            auto take = [&](int cnt, const int4 &ent, int e) {
                bad = bad || cnt > kOneListCap;
                const bool have = (e % kOneListCap) < cnt;
                const unsigned long long hm = __ballot(have);
                if (hm) {  // wave-uniform
                    int pos = 0;
                    if (lane == 0) pos = atomicAdd(&s_nl, __popcll(hm));
                    pos = __shfl(pos, 0, 64) + __popcll(hm & ((1ULL << lane) - 1ULL));
                    if (have && pos < kOneListMax) { lm_g[pos] = static_cast<uint32_t>(ent.x); lm_d[pos] = __hiloint2double(ent.w, ent.z); }
                }
            };
#pragma unroll
            for (int q = 0; q < kListPre; ++q) take(lcnt[q], le[q], threadIdx.x + 256 * q);
            const int32_t *ol = a.olist + static_cast<size_t>(pb) * kOneStride;
#pragma unroll 1
            for (int e = threadIdx.x + 256 * kListPre; e < nrow * kOneListCap; e += 256) take(ol[e / kOneListCap], reinterpret_cast<const int4 *>(ol + 256)[e], e);
            if (__ballot(bad) && lane == 0) s_bad = 1;
        }
        lds_barrier();
        for (int j = threadIdx.x; j < min(s_nl, kOneListMax); j += 256) {
            const double p = normal_p(lm_d[j], se);
            lm_m[j] = bh_rank(p, G, a.padj_deg);
            if (p <= a.pval_deg) lm_g[j] |= 0x40000000u;
        }
        lds_barrier();
        const int nl = s_nl;
        if (s_bad) { ok = false; why |= 8; }
        if (nl > kOneListMax) { ok = false; why |= 16; }
        // H(r) = n_sure + #{listed: m <= r} for the r of the band; the cut is the largest r with H(r) >= r
        int kstar = 0;
        {
            const int rlo = max(K_in, 1), span = K_out - rlo + 1;   // (<= 2 band + 1 <= 255 candidates, one per thread)
            int best = 0;
            if (ok && static_cast<int>(threadIdx.x) < span) {
                const int rr = rlo + threadIdx.x;
                int h = nsure;
                for (int j = 0; j < nl; ++j) h += lm_m[j] <= rr ? 1 : 0;
                best = h >= rr ? rr : 0;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const int u = __shfl_xor(best, o, 64); best = u > best ? u : best; }
            if (lane == 0) wbest[wave] = best;
            lds_barrier();
            kstar = max(max(wbest[0], wbest[1]), max(wbest[2], wbest[3]));
            if (ok && kstar == 0 && rlo > 1) { ok = false; why |= 32; }   // the cut lies below the band
            if (span > 256) { ok = false; why |= 64; }
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) { int32_t *dg = ls->slot[b].pad0; dg[0] = why; dg[1] = k_hi; dg[2] = nl; dg[3] = nsure; dg[4] = kstar; }
        if (!ok) {
            r.active = 0; r.need_full = 1;  // pass r.t goes to the sorting path (its tallies are in place)
        } else {
            // the mask step: only listed genes can change their bit
            for (int j = threadIdx.x; j < nl; j += 256) {
                const uint32_t g = lm_g[j];
                const uint32_t ob = g >> 31, nb = ((g & 0x40000000u) && lm_m[j] <= kstar) ? 0u : 1u;  // inds = .!(pval <= pval_deg .& padj <= padj_deg), :417
                if (nb != ob) {
                    const int at = atomicAdd(&s_n, 1);
                    atomicAdd(&s_nn, nb ? 1 : -1);
                    if (at < kDeltaMax) dl[at] = ((g & 0x3FFFFFFFu) << 1) | nb;
                }
            }
            bool ind = false;
            if (live) {
                const int sv = own_state & 3;
                if (sv == 2) {
                    const double p = normal_p(d1prev, se);
                    ind = !(p <= a.pval_deg && bh_rank(p, G, a.padj_deg) <= kstar);
                } else ind = sv != 0;
            }
            if (i < Gp) a.refbytes[nxt][i] = ind ? 1 : 0;
            const unsigned long long mk = __ballot(ind);
            if (lane == 0 && i < Gp) { a.refbits[nxt][i >> 5] = static_cast<uint32_t>(mk); a.refbits[nxt][(i >> 5) + 1] = static_cast<uint32_t>(mk >> 32); }
            lds_barrier();
            const int chg = s_n, nn = r.nref + s_nn;  // sum(inds), :417-418: the old mask's count (r.nref) + added - removed
            if (blockIdx.x == 0) {
                if (threadIdx.x == 0) { a.trace[2 * t] = G - nn; a.trace[2 * t + 1] = nn; a.scal[0] = se; }
                if (static_cast<int>(threadIdx.x) < min(chg, kDeltaMax)) a.delta_list[static_cast<size_t>(nxt) * Gp + threadIdx.x] = dl[threadIdx.x];
            }
            r.nref_prev = r.nref;
            const int diff = r.nref - nn;
            if ((diff < 0 ? -diff : diff) < a.n_conv) r.done = 1;  // :419-422
            else r.nref = nn;                                      // :423-424
            r.t = t + 1; r.ran = 1; r.dcnt = chg; r.kstar = kstar;
            r.need_full = chg > kDeltaMax ? 1 : 0;
            r.active = (!r.done && r.t < a.n_iter && !r.need_full) ? 1 : 0;
            inref = ind;
            n = min(chg, kDeltaMax);
            stepped = true;
        }
        if (!TAIL) STAMP(a, 3);
    }
    if (TAIL) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            IterState *st = a.st;
            st->passes = r.t; st->nref = r.nref; st->nref_prev = r.nref_prev; st->done = r.done; st->need_full = r.need_full;
            st->i_iter = r.t - (r.done ? 1 : 0);
            st->raw_pass = r.raw_pass;
            st->delta_cnt[r.t & 1] = r.dcnt;
            st->kstar = r.kstar;
            if (r.ran) st->last_full = 0;
            if (stepped) { a.scal[1] = win[0]; a.scal[2] = win[1]; a.scal[3] = win[2]; a.scal[4] = win[3]; }
            if (a.host_st) {
                IterState *h = a.host_st;
                h->passes = r.t; h->done = r.done; h->need_full = r.need_full; h->delta_cnt[r.t & 1] = r.dcnt;
                if (r.ran) h->last_full = 0;
            }
        }
        return;
    }
    if (r.active) r.raw_pass = r.t;  // the tallies of pass r.t are made below
    LightSlot *sl = &ls->slot[b];
    if (blockIdx.x == 0 && threadIdx.x == 0) { sl->rec = r; sl->se_base = se_base; sl->eta = eta; }
    if (!r.active || static_cast<int>(blockIdx.x) * 256 >= G) return;
    // ---- pass r.t: tallies from the changed rows, delta1, window bookkeeping, brackets, lists
    LightCnt *lc = &sl->lc;
    const double wa_lo = win[0], wa_hi = win[1], wb_lo = win[2], wb_hi = win[3];
    double v = 0.0;
    bool inner = false, inA = false, inB = false, belowA = false, belowB = false;
    if (n) {  // (workgroup-uniform)
        int d[kRaw];
        delta_counts_block(a.table, a.Wp, dl, n, d, dbuf);
        if (live) {
            r0.x += d[0]; r0.y += d[1]; r0.z += d[2]; r0.w += d[3];
            r1.x += d[4]; r1.y += d[5]; r1.z += d[6]; r1.w += d[7];
            int4 *o = reinterpret_cast<int4 *>(a.raw + static_cast<size_t>(i) * kRaw);
            o[0] = r0; o[1] = r1;
        }
    }
    STAMP(a, 4);
    if (live) {
        int32_t c[9];
        const bool tok = tallies_from(r0, r1, r.nref - (inref ? 1 : 0), c);  // the diagonal is never set (:363,385)
        double out[5] = {1.0, 0.0, 0.0, 0.0, 0.0};
        if (tok) mccullagh3<false>(c, out);
        v = out[1];
        if (!tok || !(fabs(v) < INFINITY)) { v = 0.0; raise_fault(a, kFaultTallies); }
        a.result[11 * static_cast<size_t>(G) + i] = v;
#pragma unroll
        for (int x = 0; x < kHistParts; ++x) a.hist[(static_cast<size_t>(hb_next) * kHistParts + x) * a.hist_stride + i] = 0;  // the histogram of the NEXT launch: last read by the launch before this one
        belowA = v < wa_lo; inA = !belowA && v <= wa_hi;
        belowB = v < wb_lo; inB = !belowB && v <= wb_hi;
        inner = v > wa_hi && v < wb_lo;
    }
    STAMP(a, 5);
    {
        double nb, sum;
        block_sum2_256(inner ? 1.0 : 0.0, inner ? v : 0.0, red, nb, sum);
        const double mean = nb > 0.0 ? sum / nb : 0.0;
        const double m2 = block_sum_256(inner ? (v - mean) * (v - mean) : 0.0, red);
        if (threadIdx.x == 0) { double *pp = a.part + static_cast<size_t>(pbuf) * 768 + 3 * blockIdx.x; pp[0] = nb; pp[1] = mean; pp[2] = m2; }
    }
    const unsigned long long mA = __ballot(inA), mB = __ballot(inB), bA = __ballot(belowA), bB = __ballot(belowB);
    int baseA = 0, baseB = 0;
    if (lane == 0) {
        if (mA) baseA = atomicAdd(&lc->cnt_a, __popcll(mA));
        if (mB) baseB = atomicAdd(&lc->cnt_b, __popcll(mB));
        wcnt[wave][0] = __popcll(bA); wcnt[wave][1] = __popcll(bB);
    }
    baseA = __shfl(baseA, 0, 64); baseB = __shfl(baseB, 0, 64);
    const unsigned long long lt = (1ULL << lane) - 1ULL;
    double *cand = a.cand + pbuf * 2 * kCandMax;
    if (inA) { const int at = baseA + __popcll(mA & lt); if (at < kCandMax) cand[at] = v; }
    if (inB) { const int at = baseB + __popcll(mB & lt); if (at < kCandMax) cand[kCandMax + at] = v; }
    if (threadIdx.x == 0) s_c = 0;
    lds_barrier();
    if (threadIdx.x == 0) {
        const int ba = wcnt[0][0] + wcnt[1][0] + wcnt[2][0] + wcnt[3][0], bb = wcnt[0][1] + wcnt[1][1] + wcnt[2][1] + wcnt[3][1];
        if (ba) spread_add(lc->below_a, ba, a.xcc_local);
        if (bb) spread_add(lc->below_b, bb, a.xcc_local);
    }
    STAMP(a, 6);
    // brackets of this gene's p-value and rank around se_base (the se of the pass before; this pass's is not known yet)
    {
        const double Gd = static_cast<double>(G);
        const double p_lo = live ? normal_p(v, se_base * (1.0 - eta)) : 1.0, p_hi = live ? normal_p(v, se_base * (1.0 + eta)) : 1.0;
        const int m_lo = live ? bh_rank(p_lo, G, a.padj_deg) : G + 1;
        const int K_in = r.kstar - a.band, K_out = min(r.kstar + a.band, G);
        const bool bh_in = K_in >= 1 && p_hi * (Gd / static_cast<double>(K_in)) <= a.padj_deg;   // m_hi <= K_in (the step-up rule's own expression)
        const bool bh_out = m_lo > K_out;
        const bool pd_pass = p_hi <= a.pval_deg, pd_fail = p_lo > a.pval_deg;
        const bool bit_sure = bh_out || pd_fail || (bh_in && pd_pass);
        const bool newbit = bh_out || pd_fail;   // (when sure) 1 = not a DEG: stays in / enters the reference set
        const bool listed = live && (!(bh_in || bh_out) || !bit_sure || newbit != inref);
        {   // the histogram of m_lo: one partial per XCD (atomics that stay in that XCD's L2), the hot bin once per wave
            const unsigned long long first = __ballot(m_lo == 1), finite = __ballot(m_lo <= G), sure_in = __ballot(live && bh_in && !listed);
            const bool lead1 = m_lo == 1 && lane == __ffsll(static_cast<long long>(first)) - 1;
            if (a.xcc_local) {
                int32_t *hist = a.hist + (static_cast<size_t>(hb) * kHistParts + xcc_id()) * a.hist_stride;
                if (lead1) __hip_atomic_fetch_add(&hist[0], __popcll(first), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else if (m_lo >= 2 && m_lo <= G) __hip_atomic_fetch_add(&hist[m_lo - 1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                int32_t *hist = a.hist + static_cast<size_t>(hb) * kHistParts * a.hist_stride;
                if (lead1) atomicAdd(&hist[0], __popcll(first));
                else if (m_lo >= 2 && m_lo <= G) atomicAdd(&hist[m_lo - 1], 1);
            }
            if (lane == 0 && finite) spread_add(lc->sig, __popcll(finite), a.xcc_local);
            if (lane == 0 && sure_in) spread_add(lc->nsure, __popcll(sure_in), a.xcc_local);
        }
        if (live) a.mrank[i] = listed ? 2 : (newbit ? 1 : 0);
        // the list of this workgroup: genes that the next launch decides exactly
        const unsigned long long cm = __ballot(listed);
        int pos = 0;
        if (cm && lane == 0) pos = atomicAdd(&s_c, __popcll(cm));
        pos = __shfl(pos, 0, 64) + __popcll(cm & ((1ULL << lane) - 1ULL));
        int32_t *ol = a.olist + static_cast<size_t>(pbuf) * kOneStride;
        if (listed && pos < kOneListCap)
            reinterpret_cast<int4 *>(ol + 256)[blockIdx.x * kOneListCap + pos] =
                make_int4(static_cast<int>(static_cast<uint32_t>(i) | (inref ? 0x80000000u : 0u)), 0, __double2loint(v), __double2hiint(v));
        lds_barrier();
        if (threadIdx.x == 0) ol[blockIdx.x] = min(s_c, kOneListCap + 1);
    }
    STAMP(a, 7);
}

// ---------------------------------------------------------------------------
// The light passes as ONE persistent launch: the two launches of a pass (kl_head, kl_rank) become the two phases of a
// loop, and their boundaries two grid barriers.  Thread i owns gene i for the whole launch: its tally counters, mask bit,
// delta1, p-value and BH rank stay in registers; what crosses workgroups (histogram, rank rows, window members, block
// moments, counters) goes through coherent loads / stores (ldc / stc: sc1) or agent-scope atomics.  The loop state is
// computed by every workgroup for itself, identically.  Gp / 256 <= 256 workgroups of 256 threads: all resident.
//
// Grid barrier: one monotonic counter (zeroed by the host before the launch); every wave drains its stores
// (s_waitcnt vmcnt(0)), the workgroup meets, lane 0 adds one and polls with sc1 loads until every workgroup of the round
// has arrived, the workgroup meets again.  No fences: measured 2.3 us per round with 80 workgroups against 4.6 us with a
// release / acquire fence pair (tools/microbench_gridbar.hip).  Every spin is bounded (about 0.5 s): on expiry the
// workgroup raises st->fault and leaves, and so do the others at their next barrier, so the grid always drains.
__device__ __forceinline__ bool grid_barrier(unsigned *bar, unsigned nwg, unsigned &gen, int *fault)
{
    __shared__ int ok_s;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        ++gen;
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = gen * nwg;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
        int ok = 1;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (__builtin_amdgcn_s_memrealtime() - t0 > 50000000ull || __hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(fault, kFaultBarrier, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        ok_s = ok;
    }
    __syncthreads();
    return ok_s != 0;
}

__global__ __launch_bounds__(256) void kl_persist(IterArgs a, LightState *ls, unsigned *bar)
{
    IterState *st = a.st;
    LightRec r;  // the state as the previous launch left it (a kernel boundary: plain loads)
    r.t = st->passes; r.nref = st->nref; r.nref_prev = st->nref_prev; r.done = st->done; r.need_full = st->need_full;
    r.raw_pass = st->raw_pass; r.ran = 0; r.dcnt = st->delta_cnt[r.t & 1]; r.kstar = -1;
    r.active = (!r.done && r.t < a.n_iter && !r.need_full) ? 1 : 0;
    if (!r.active || st->fault) return;  // the same in every workgroup
    const int G = a.G, Gp = a.Gp;
    const unsigned nwg = gridDim.x;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < G, holds = static_cast<int>(blockIdx.x) * 256 < G;  // (workgroups of padding genes only help with the barriers)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nrow = (G + 255) >> 8;
    __shared__ uint32_t dl[kDeltaMax];
    __shared__ int s_n, s_nn;
    __shared__ double red[256];
    __shared__ double sel[2][4];
    __shared__ int wcnt[4][2];
    __shared__ int32_t row[256];
    int4 r0 = make_int4(0, 0, 0, 0), r1 = make_int4(0, 0, 0, 0);
    bool inref = false;
    if (live) {
        const int4 *o = reinterpret_cast<const int4 *>(a.raw + static_cast<size_t>(i) * kRaw);
        r0 = o[0]; r1 = o[1];
        inref = a.refbytes[r.t & 1][i] != 0;
    }
    double win[4] = {a.scal[1], a.scal[2], a.scal[3], a.scal[4]};
    const double wd0 = a.scal[5], wd1 = a.scal[6], wd2 = a.scal[7], wd3 = a.scal[8];
    int n = r.raw_pass == r.t ? 0 : min(r.dcnt, kDeltaMax);
    if (static_cast<int>(threadIdx.x) < n) dl[threadIdx.x] = a.delta_list[static_cast<size_t>(r.t & 1) * Gp + threadIdx.x];
    lds_barrier();
    double d1 = 0.0, p = 1.0, se = 0.0;
    int own_rank = 0;
    unsigned gen = 0;
    bool windows_moved = false, tally_fault = false;
    while (true) {
        const int par = r.t & 1;
        LightCnt *lc = &ls->slot[par].lc;
        int32_t *hist = a.hist + static_cast<size_t>(par) * a.hist_stride;
        STAMP(a, 0);
        // ---------------- phase 1 (kl_head's second half): tallies from the changed rows, delta1, window bookkeeping
        bool inner = false, inA = false, inB = false, belowA = false, belowB = false;
        if (live) {
            if (n) {
                int d[kRaw];
                delta_counts<false, 8>(a.table, a.Wp, dl, n, i, d);  // (the table does not change: plain loads)
                r0.x += d[0]; r0.y += d[1]; r0.z += d[2]; r0.w += d[3];
                r1.x += d[4]; r1.y += d[5]; r1.z += d[6]; r1.w += d[7];
            }
            int32_t c[9];
            const bool tok = tallies_from(r0, r1, r.nref - (inref ? 1 : 0), c);  // the diagonal is never set (:363,385)
            double out[5] = {1.0, 0.0, 0.0, 0.0, 0.0};
            if (tok) mccullagh3<false>(c, out);
            d1 = out[1];
            if (!tok || !(fabs(d1) < INFINITY)) { d1 = 0.0; tally_fault = true; }
            stc<true>(hist + i, 0);  // this parity's histogram: last read two passes ago
            belowA = d1 < win[0]; inA = !belowA && d1 <= win[1];
            belowB = d1 < win[2]; inB = !belowB && d1 <= win[3];
            inner = d1 > win[1] && d1 < win[2];
        }
        r.raw_pass = r.t;
        {
            double nb, sum;
            block_sum2_256(inner ? 1.0 : 0.0, inner ? d1 : 0.0, red, nb, sum);
            const double mean = nb > 0.0 ? sum / nb : 0.0;
            const double m2 = block_sum_256(inner ? (d1 - mean) * (d1 - mean) : 0.0, red);
            if (threadIdx.x == 0 && holds) { stc<true>(a.part + 3 * blockIdx.x, nb); stc<true>(a.part + 3 * blockIdx.x + 1, mean); stc<true>(a.part + 3 * blockIdx.x + 2, m2); }
        }
        {
            const unsigned long long mA = __ballot(inA), mB = __ballot(inB), bA = __ballot(belowA), bB = __ballot(belowB);
            int baseA = 0, baseB = 0;
            if (lane == 0) {
                if (mA) baseA = atomicAdd(&lc->cnt_a, __popcll(mA));
                if (mB) baseB = atomicAdd(&lc->cnt_b, __popcll(mB));
                wcnt[wave][0] = __popcll(bA); wcnt[wave][1] = __popcll(bB);
            }
            baseA = __shfl(baseA, 0, 64); baseB = __shfl(baseB, 0, 64);
            const unsigned long long lt = (1ULL << lane) - 1ULL;
            if (inA) { const int at = baseA + __popcll(mA & lt); if (at < kCandMax) stc<true>(a.cand + at, d1); }
            if (inB) { const int at = baseB + __popcll(mB & lt); if (at < kCandMax) stc<true>(a.cand + kCandMax + at, d1); }
            lds_barrier();
            if (threadIdx.x == 0) {
                const int ba = wcnt[0][0] + wcnt[1][0] + wcnt[2][0] + wcnt[3][0], bb = wcnt[0][1] + wcnt[1][1] + wcnt[2][1] + wcnt[3][1];
                if (ba) atomicAdd(&lc->below_a[blockIdx.x & (kSpread - 1)][0], ba);
                if (bb) atomicAdd(&lc->below_b[blockIdx.x & (kSpread - 1)][0], bb);
            }
        }
        STAMP(a, 1);
        if (!grid_barrier(bar, nwg, gen, &st->fault)) break;
        STAMP(a, 2);
        // ---------------- phase 2 (kl_rank): the selection, p-values, BH ranks and their histogram
        if (blockIdx.x == 0 && threadIdx.x < kSpread) {
            // the other parity's counters: every workgroup is past its last look at them (the mask step in front of phase 1)
            LightCnt *z = &ls->slot[par ^ 1].lc;
            if (threadIdx.x == 0) { stc<true>(&z->cnt_a, 0); stc<true>(&z->cnt_b, 0); }
            stc<true>(&z->below_a[threadIdx.x][0], 0); stc<true>(&z->below_b[threadIdx.x][0], 0); stc<true>(&z->sig[threadIdx.x][0], 0);
        }
        {
            const int cnt_a = ldc<true>(&lc->cnt_a), cnt_b = ldc<true>(&lc->cnt_b);
            int below_a = 0, below_b = 0;
#pragma unroll
            for (int q = 0; q < kSpread; ++q) { below_a += ldc<true>(&lc->below_a[q][0]); below_b += ldc<true>(&lc->below_b[q][0]); }
            const double x = wave < 2 ? ldc<true>(a.cand + wave * kCandMax + lane) : 0.0;
            double pn[kPartPer] = {0.0}, pm[kPartPer] = {0.0}, pq[kPartPer] = {0.0};   // (this form: at most 65 535 genes)
            if (static_cast<int>(threadIdx.x) < nrow) { pn[0] = ldc<true>(a.part + 3 * threadIdx.x); pm[0] = ldc<true>(a.part + 3 * threadIdx.x + 1); pq[0] = ldc<true>(a.part + 3 * threadIdx.x + 2); }
            double va = 0.0, vb = 0.0;
            bool ok = slice_std_vals(a, x, pn, pm, pq, below_a, below_b, cnt_a, cnt_b, sel, red, se, va, vb);
            ok = ok && (va + wd1 < vb - wd2);
            if (!ok) { r.active = 0; r.need_full = 1; break; }  // the same in every workgroup: this pass runs on the sorting path (its tallies are in place)
            win[0] = va - wd0; win[1] = va + wd1; win[2] = vb - wd2; win[3] = vb + wd3;  // the next pass's windows
            windows_moved = true;
        }
        int m = G + 1;
        if (live) {
            p = normal_p(d1, se);
            m = bh_rank(p, G, a.padj_deg);
            own_rank = p <= a.pval_deg ? m : G + 2;
        }
        row[4 * lane + wave] = live ? static_cast<int32_t>(static_cast<uint32_t>(own_rank) | (inref ? 0x80000000u : 0u)) : 0;
        lds_barrier();
        stc<true>(a.mrank + blockIdx.x * 256 + threadIdx.x, row[threadIdx.x]);
        {
            const unsigned long long first = __ballot(m == 1), finite = __ballot(m <= G);
            if (m == 1) { if (lane == __ffsll(static_cast<long long>(first)) - 1) atomicAdd(&hist[0], __popcll(first)); }
            else if (m <= G) atomicAdd(&hist[m - 1], 1);
            if (lane == 0 && finite) atomicAdd(&lc->sig[blockIdx.x & (kSpread - 1)][0], __popcll(finite));
        }
        STAMP(a, 3);
        if (!grid_barrier(bar, nwg, gen, &st->fault)) break;
        STAMP(a, 4);
        // ---------------- the mask step of pass r.t (kl_head's first half, :413-424), by every workgroup for itself
        int hv[4][8];
        int4 mv[kHeadPre];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned long long *hp = reinterpret_cast<const unsigned long long *>(hist + (e * 256 + threadIdx.x) * 8);
#pragma unroll
            for (int u = 0; u < 4; ++u) { const unsigned long long w = ldc<true>(hp + u); hv[e][2 * u] = static_cast<int>(w); hv[e][2 * u + 1] = static_cast<int>(w >> 32); }
        }
#pragma unroll
        for (int q = 0; q < kHeadPre; ++q) {
            const int R = wave + 4 * q;
            mv[q] = make_int4(0, 0, 0, 0);
            if (R < nrow) mv[q] = load_rank_row<true>(a.mrank, R, lane);
        }
        int sig = 0;
#pragma unroll
        for (int q = 0; q < kSpread; ++q) sig += ldc<true>(&lc->sig[q][0]);
        STAMP(a, 5);
        const int t = r.t, nxt = par ^ 1;
        int carry0 = 0;
        const int kstar = sig <= 8192 ? bh_cut4(hv, G, 0, carry0) : bh_cut<true>(hist, G, sig);
        STAMP(a, 6);
        if (threadIdx.x == 0) { s_n = 0; s_nn = 0; }
        lds_barrier();
        mask_scan<true>(mv, kstar, nrow, a.mrank, dl, &s_n, &s_nn);
        const bool ind = live && own_rank > kstar;
        if (i < Gp) a.refbytes[nxt][i] = ind ? 1 : 0;  // (read again by later launches only)
        const unsigned long long mk = __ballot(ind);
        if (lane == 0 && i < Gp) { a.refbits[nxt][i >> 5] = static_cast<uint32_t>(mk); a.refbits[nxt][(i >> 5) + 1] = static_cast<uint32_t>(mk >> 32); }
        lds_barrier();
        const int chg = s_n, nn = r.nref + s_nn;  // sum(inds), :417-418
        if (blockIdx.x == 0) {
            if (threadIdx.x == 0) { a.trace[2 * t] = G - nn; a.trace[2 * t + 1] = nn; }
            if (static_cast<int>(threadIdx.x) < min(chg, kDeltaMax)) a.delta_list[static_cast<size_t>(nxt) * Gp + threadIdx.x] = dl[threadIdx.x];
        }
        STAMP(a, 7);
        r.nref_prev = r.nref;
        const int diff = r.nref - nn;
        if ((diff < 0 ? -diff : diff) < a.n_conv) r.done = 1;  // :419-422
        else r.nref = nn;                                      // :423-424
        r.t = t + 1; r.ran = 1; r.dcnt = chg;
        r.need_full = chg > kDeltaMax ? 1 : 0;
        r.active = (!r.done && r.t < a.n_iter && !r.need_full) ? 1 : 0;
        inref = ind;
        n = min(chg, kDeltaMax);
        if (!r.active) break;
    }
    // ---------------- hand the state back to the launches that follow
    if (tally_fault) raise_fault(a, kFaultTallies);  // (after the last barrier: the barriers poll the same word to leave early)
    if (live) {
        int4 *o = reinterpret_cast<int4 *>(a.raw + static_cast<size_t>(i) * kRaw);
        o[0] = r0; o[1] = r1;
        a.result[11 * static_cast<size_t>(G) + i] = d1;
        a.result[i] = p;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->passes = r.t; st->nref = r.nref; st->nref_prev = r.nref_prev; st->done = r.done; st->need_full = r.need_full;
        st->i_iter = r.t - (r.done ? 1 : 0);
        st->raw_pass = r.raw_pass;
        st->delta_cnt[r.t & 1] = r.dcnt;
        if (r.ran) st->last_full = 0;
        a.scal[0] = se;
        if (windows_moved) { a.scal[1] = win[0]; a.scal[2] = win[1]; a.scal[3] = win[2]; a.scal[4] = win[3]; }
    }
}

// ---------------------------------------------------------------------------
// Exchange of the class table between shards, gather form (api.hip, exchange_table).  A work unit of the pair kernel
// = kUnitH i-tiles (1024 gene rows) x one panel of W = 32 Wc gene columns; its FORWARD words are the rectangle rows x
// (words of the panel) of the table, its MIRROR words (the same pairs seen from the other gene, :386) the transposed
// rectangle with the low / high planes swapped.  A shard packs the forward rectangles of its own units as they stand
// in its table; after the all-gather every shard ORs the others' rectangles into its table (x_expand_fwd), then the
// transposes (x_expand_mirror).  Rectangles of different units are disjoint, and so are transposed rectangles; a
// rectangle can overlap a transposed one next to the diagonal, which is why the two steps are separate launches and
// everything is OR-ed (bits that arrive twice -- a sender's own mirror bits inside one of its rectangles -- are the
// same bits).  Pack layout of a unit: [1024 rows][4 planes][Wc words]; rows and words past the table are zeros.
struct XArgs {
    uint32_t *table;
    const uint32_t *units;  // every unit of the build: panel << 16 | i-range; owner = index % world
    int G, Wp, Wc, total, world, rank, maxu;
    int m0, mcnt;           // the slots (units per shard: unit = shard + slot * world) of this exchange: [m0, m0 + mcnt) -- all of them, or one wave's
};
constexpr int kUnitRows = kUnitH * kTileI;

__global__ __launch_bounds__(256) void x_pack(XArgs a, uint32_t *__restrict__ send)
{
    const int m = blockIdx.y, gu = a.rank + (a.m0 + m) * a.world;
    const size_t uw = static_cast<size_t>(kUnitRows) * kPlanes * a.Wc;
    const size_t e = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
    if (e >= uw) return;
    uint32_t v = 0;
    if (gu < a.total) {
        const uint32_t um = a.units[gu];
        const int p = static_cast<int>(um >> 16), r = static_cast<int>(um & 0xFFFFu);
        const int w = static_cast<int>(e % a.Wc), pl = static_cast<int>((e / a.Wc) % kPlanes), rr = static_cast<int>(e / (static_cast<size_t>(a.Wc) * kPlanes));
        const int row = r * kUnitRows + rr, cw = p * a.Wc + w;
        if (row < a.G && cw < a.Wp) v = a.table[(static_cast<size_t>(row) * kPlanes + pl) * a.Wp + cw];
    }
    send[static_cast<size_t>(m) * uw + e] = v;
}

__global__ __launch_bounds__(256) void x_expand_fwd(XArgs a, const uint32_t *__restrict__ recv)
{
    const int s = blockIdx.y / a.mcnt, m = blockIdx.y % a.mcnt, gu = s + (a.m0 + m) * a.world;
    if (s == a.rank || gu >= a.total) return;
    const size_t uw = static_cast<size_t>(kUnitRows) * kPlanes * a.Wc;
    const size_t e = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
    if (e >= uw) return;
    const uint32_t v = recv[(static_cast<size_t>(s) * a.mcnt + m) * uw + e];
    if (!v) return;   // (only words that carry bits are touched: see the note on concurrent pair kernels below)
    const uint32_t um = a.units[gu];
    const int p = static_cast<int>(um >> 16), r = static_cast<int>(um & 0xFFFFu);
    const int w = static_cast<int>(e % a.Wc), pl = static_cast<int>((e / a.Wc) % kPlanes), rr = static_cast<int>(e / (static_cast<size_t>(a.Wc) * kPlanes));
    const int row = r * kUnitRows + rr, cw = p * a.Wc + w;
    if (row < a.G && cw < a.Wp) a.table[(static_cast<size_t>(row) * kPlanes + pl) * a.Wp + cw] |= v;
}

// One workgroup per (unit, column word jb of the panel, plane); each wave transposes pairs of 32 x 32 bit blocks: lane l
// holds the word of source row 32 ib + (l & 31), ib = 2 pair + (l >> 5); the ballot of bit c over the wave is, for output
// row 32 jb + c, the two words that cover source rows 64 pair .. 64 pair + 63 -- kept by lane c and OR-ed into the table.
__global__ __launch_bounds__(256) void x_expand_mirror(XArgs a, const uint32_t *__restrict__ recv)
{
    const int s = blockIdx.z / a.mcnt, m = blockIdx.z % a.mcnt, gu = s + (a.m0 + m) * a.world;
    if (s == a.rank || gu >= a.total) return;
    const int jb = blockIdx.x, pl = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t uw = static_cast<size_t>(kUnitRows) * kPlanes * a.Wc;
    const uint32_t *src = recv + (static_cast<size_t>(s) * a.mcnt + m) * uw;
    const uint32_t um = a.units[gu];
    const int p = static_cast<int>(um >> 16), r = static_cast<int>(um & 0xFFFFu);
    const int j = (p * a.Wc + jb) * 32 + (lane & 31);  // the output row of lanes 0..31
    for (int pair = wave; pair < kUnitRows / 64; pair += 4) {
        const int rr = 64 * pair + lane;  // source row inside the unit
        const uint32_t word = src[(static_cast<size_t>(rr) * kPlanes + pl) * a.Wc + jb];
        if (!__ballot(word != 0)) continue;  // wave-uniform: nothing in these 64 x 32 bits
        unsigned long long mine = 0;
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            const unsigned long long b = __ballot((word >> c) & 1u);
            mine = lane == c ? b : mine;
        }
        const int ow = (r * kUnitRows) / 32 + 2 * pair;  // two consecutive words of the output row
        if (lane < 32 && j < a.G && mine != 0 && ow + 1 < a.Wp) {
            // word by word, and only words that carry bits: every word of the table belongs to ONE tile (all its bits come from
            // one unit, hence one shard), so a word with bits of this sender is touched by nobody else -- but its neighbour may
            // be a word that this shard's own pair kernel is writing right now (the pipelined exchange runs beside the pair
            // kernel's later waves), and a read-modify-write of a pair of words would put a stale value back
            uint32_t *o = a.table + (static_cast<size_t>(j) * kPlanes + (pl ^ 1)) * a.Wp + ow;  // L <-> H
            const uint32_t lo32 = static_cast<uint32_t>(mine), hi32 = static_cast<uint32_t>(mine >> 32);
            if (lo32) o[0] |= lo32;
            if (hi32) o[1] |= hi32;
        }
    }
}

}  // namespace

// ------------------------------------------------------------------ launchers

// bits needed for every number the pair kernel compares: positions 0..G-1 and band ends up to G
static int plane_bits(int64_t G) { return G <= 4095 ? 12 : (G <= 32767 ? 15 : (G <= 65535 ? 16 : (G <= 131071 ? 17 : 18))); }

// the wave form with more than 16 planes (more than 65 535 genes; at most 65 535 samples): only the wave kernels have a loop for them
template <int NB>
static void launch_big_pairs(reo_ctx *c, const K1Args &a, bool shared, bool multi, size_t plane_elems, bool wide)
{
    const unsigned gridw = static_cast<unsigned>(c->k1_items_n);
    if (wide) {  // (two groups: launch_k1 refuses the rest)
        if (gridw == 0) return;
        if (c->has_ties) k1w_pairs_wide<NB, true><<<gridw, 64, 0, c->stream>>>(a);
        else k1w_pairs_wide<NB, false><<<gridw, 64, 0, c->stream>>>(a);
        return;
    }
    if (shared) {
        if (!c->gc_valid) {
            if (gridw > 0) {
                if (c->has_ties) k1w_group_counts<NB, true><<<gridw, 64, 0, c->stream>>>(a, c->gcounts.p, plane_elems);
                else k1w_group_counts<NB, false><<<gridw, 64, 0, c->stream>>>(a, c->gcounts.p, plane_elems);
            }
            c->gc_valid = true;
        }
        const unsigned grid = static_cast<unsigned>((a.n_units + 7) / 8 * 8 * kUnitH * a.Q);
        k1_classify<kRJ><<<grid, 256, 0, c->stream>>>(a, c->gcounts.p, plane_elems);
        return;
    }
    if (gridw == 0) return;
    if (multi) {
        if (c->has_ties) k1w_pairs_multi<NB, true><<<gridw, 64, 0, c->stream>>>(a);
        else k1w_pairs_multi<NB, false><<<gridw, 64, 0, c->stream>>>(a);
    } else if (c->has_ties) k1w_pairs<NB, true><<<gridw, 64, 0, c->stream>>>(a);
    else k1w_pairs<NB, false><<<gridw, 64, 0, c->stream>>>(a);
}

template <int NB>
static void launch_pair_kernels(reo_ctx *c, const K1Args &a, unsigned grid, bool shared, bool multi, size_t plane_elems, bool wide)
{
    if (wide) {  // more than 65 535 samples: 32-bit counts (never the shared per-group planes, which are 16-bit)
        if (multi) {
            if (c->has_ties) k1_pairs_wide<NB, true, true><<<grid, 256, 0, c->stream>>>(a);
            else k1_pairs_wide<NB, false, true><<<grid, 256, 0, c->stream>>>(a);
        } else if (c->k1_wave) {  // wave form: the count loop in runs of 2 047 blocks, 32-bit totals
            const unsigned gridw = static_cast<unsigned>(c->k1_items_n);
            if (gridw == 0) return;
            if (c->has_ties) k1w_pairs_wide<NB, true><<<gridw, 64, 0, c->stream>>>(a);
            else k1w_pairs_wide<NB, false><<<gridw, 64, 0, c->stream>>>(a);
        } else {
            if (c->has_ties) k1_pairs_wide<NB, true, false><<<grid, 256, 0, c->stream>>>(a);
            else k1_pairs_wide<NB, false, false><<<grid, 256, 0, c->stream>>>(a);
        }
    } else if (shared) {
        if (!c->gc_valid) {
            if (c->k1_wave) {  // one wave per workgroup, the generated count loop, one item per (tile, chunk)
                const unsigned gridw = static_cast<unsigned>(c->k1_items_n);
                if (gridw > 0) {
                    if (c->has_ties) k1w_group_counts<NB, true><<<gridw, 64, 0, c->stream>>>(a, c->gcounts.p, plane_elems);
                    else k1w_group_counts<NB, false><<<gridw, 64, 0, c->stream>>>(a, c->gcounts.p, plane_elems);
                }
            } else if (c->has_ties) k1_group_counts<NB, true><<<grid, 256, 0, c->stream>>>(a, c->gcounts.p, plane_elems);
            else k1_group_counts<NB, false><<<grid, 256, 0, c->stream>>>(a, c->gcounts.p, plane_elems);
            c->gc_valid = true;
        }
        if (c->has_ties && !c->k1_wave) k1_classify<kRJTies><<<grid, 256, 0, c->stream>>>(a, c->gcounts.p, plane_elems);
        else k1_classify<kRJ><<<grid, 256, 0, c->stream>>>(a, c->gcounts.p, plane_elems);
    } else if (multi) {
        if (c->has_ties) k1_pairs<NB, true, true><<<grid, 256, 0, c->stream>>>(a);
        else k1_pairs<NB, false, true><<<grid, 256, 0, c->stream>>>(a);
    } else if (c->k1_wave) {  // one wave per workgroup, generated count loop (two groups)
        const unsigned gridw = static_cast<unsigned>(c->k1_items_n);
        if (gridw == 0) return;
        if (a.gate) k1w_pairs_gated<NB><<<gridw, 64, 0, c->stream>>>(a);   // the tie form is chosen on the device (K1Args::gate)
        else if (c->has_ties) k1w_pairs<NB, true><<<gridw, 64, 0, c->stream>>>(a);
        else k1w_pairs<NB, false><<<gridw, 64, 0, c->stream>>>(a);
    } else {
        if (c->has_ties) k1_pairs<NB, true, false><<<grid, 256, 0, c->stream>>>(a);
        else k1_pairs<NB, false, false><<<grid, 256, 0, c->stream>>>(a);
    }
}

static int32_t exchange_args(reo_ctx *c, XArgs &a, int m0, int mcnt);

// sides (wave form, two groups): which sides' items are launched -- bit 0 the comparison's own group, bit 1 the rest; 3 = the whole
// table.  keep_table: the class table has been cleared by the caller and holds other sides' planes already (the pipelined upload,
// transform.hip eager_upload, launches a side as soon as its group's samples are ranked).
int32_t launch_k1(reo_ctx *c, int k, int sides, bool keep_table, const int32_t *gate, const K1Range *range, bool prepare)
{
    K1Args a;
    a.P = c->pos.p; a.AL = c->lo.p; a.AH = c->hi.p; a.table = c->table.p;
    a.G = static_cast<int>(c->G); a.Gp = c->Gp; a.Wp = c->Wp;
    const bool multi = c->ngroups > 2;
    const int other = multi ? k : 1 - k;  // two groups: the treat side is the other group
    a.cb = c->goff32[k] / 32; a.ce = c->goff32[k + 1] / 32;
    a.tb = c->goff32[other] / 32; a.te = c->goff32[other + 1] / 32;
    a.gc = k; a.gt = other;
    a.nc = c->goff[k + 1] - c->goff[k]; a.nt = static_cast<int>(c->S) - a.nc;  // gsi1, gsi2 (:358-359)
    a.goff = c->goff_dev.p; a.ngroups = c->ngroups;
    a.m1 = c->thr[2 * k]; a.m2 = c->thr[2 * k + 1];
    a.seed = c->seed;

    // work units: panel p = Q consecutive j-chunks, cut into i-ranges of kUnitH tiles.  Q keeps the
    // panel's pos planes (Q x 256 RJ genes x nblk blocks x 64 B) within about 2 MiB of the 4 MiB L2 of an XCD.
    const bool wide = c->S > 65535;  // a count may not fit 16 bits: the unpacked form of the pair loop
    const bool big = c->G > 65535;    // more than 16 position planes: only the wave form has a loop for them
    if (big && wide && multi) {
        set_error("more than 65535 genes and more than 65535 samples: two groups only (the one-vs-rest wide kernel reads the 16-plane layout)");
        return REO_EINVAL;
    }
    const bool wave = (c->k1_wave || big) && !multi;  // the wave form (two groups; also more than 65 535 samples: k1w_pairs_wide): kRJ genes per lane
    int32_t rc;
    if (sides != 3 && (!wave || wide || c->world > 1)) { set_error("a single side of the pair kernel: wave form, one shard, at most 65535 samples"); return REO_EINVAL; }
    // > 2 groups: count every group once, then classify per comparison -- if the planes fit
    const size_t plane_elems = static_cast<size_t>(c->Gp) * c->Gp;
    bool shared = multi && c->share_counts && !wide;
    if (shared && !c->gc_valid) {
        size_t free_b = 0, total_b = 0;
        REO_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
        const size_t need = plane_elems * (c->ngroups + 1) * sizeof(uint16_t);
        const size_t have = c->gcounts.n * sizeof(uint16_t);
        if (need > have && need - have + (size_t(4) << 30) > free_b) shared = false;  // keep 4 GiB for everything else
        if (shared && (rc = c->gcounts.ensure(plane_elems * (c->ngroups + 1)))) return rc;
    }
    const bool wcounts = shared && (c->k1_wave || big);  // the per-group counts by the wave form's loop (k1w_group_counts): kRJ genes per lane too
    const bool wmulti = multi && !shared && !wide && big;  // a comparison recounted by the wave form (k1w_pairs_multi): above 65 535 genes only --
                                                           // below, the workgroup form recounts 10 % faster (ten groups: 3.7 - 4.3 against 4.1 - 4.8 ms)
    const int RJ = (wave || wcounts || wmulti) ? kRJ : (wide ? (c->has_ties ? kRJWideTies : kRJWide) : (c->has_ties ? kRJTies : kRJ));  // genes j per lane
    const int CJ = kTileJ * RJ;
    const int NJ = (c->Gp + CJ - 1) / CJ, NIT = c->Gp / kTileI;
    const size_t chunk_bytes = static_cast<size_t>(CJ) * (c->goff32[c->ngroups] / 32) * 64;
    const int Q = chunk_bytes * 4 <= (2u << 20) ? 4 : (chunk_bytes * 2 <= (2u << 20) ? 2 : 1);
    const int NP = (NJ + Q - 1) / Q;
    std::vector<uint32_t> units;
    c->units_all_host.clear();
    int64_t owned = 0, total = 0;
    uint32_t gu = 0;
    for (int p = 0; p < NP; ++p) {
        const int ni = std::min(NIT, (CJ / kTileI) * Q * (p + 1));  // i-tiles that reach this panel's columns
        for (int r = 0; r * kUnitH < ni; ++r, ++gu) {
            const bool mine = c->world == 1 || static_cast<int>(gu % c->world) == c->rank;
            c->units_all_host.push_back(static_cast<uint32_t>(p) << 16 | static_cast<uint32_t>(r));
            if (mine) units.push_back(static_cast<uint32_t>(p) << 16 | static_cast<uint32_t>(r));
            for (int t = r * kUnitH; t < std::min(ni, (r + 1) * kUnitH); ++t)
                for (int jc = p * Q; jc < std::min(NJ, (p + 1) * Q); ++jc) {
                    if ((jc * CJ + CJ - 1) / 64 < (t * kTileI) / 64) continue;
                    ++total;
                    if (mine) ++owned;
                }
        }
    }
    c->tiles_owned = owned; c->tiles_total = total;
    c->k1_cj = CJ; c->k1_q = Q;
    a.n_units = static_cast<int>(units.size()); a.Q = Q;
    if ((rc = c->unit_map.ensure(std::max<size_t>(units.size(), 1)))) return rc;
    if (!units.empty() && (c->unit_map.p != c->unit_map_uploaded || units != c->unit_map_host)) {  // (the same geometry as last time: already there)
        REO_HIP_CHECK(hipMemcpyAsync(c->unit_map.p, units.data(), units.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        REO_HIP_CHECK(hipStreamSynchronize(c->stream));  // `units` is a local: the copy must have read it before any return below
        c->unit_map_host = units;
        c->unit_map_uploaded = c->unit_map.p;
    }
    a.unit_map = c->unit_map.p;
    a.items = nullptr; a.stamps = nullptr; a.gate = gate;
    a.park = nullptr; a.park_ge = nullptr; a.park_mode = 0;
    if ((range || prepare) && (sides != 1 && sides != 2)) { set_error("a range of sample blocks / a prepared launch: one side of the pair kernel"); return REO_EINVAL; }
    if (gate && (!wave || wide || big)) { set_error("a gated launch of the pair kernel: wave form, at most 65535 genes and samples"); return REO_EINVAL; }
    const unsigned grid = static_cast<unsigned>((units.size() + 7) / 8 * 8 * kUnitH * Q);   // (workgroup forms)
    // item list of the wave form for a set of units: the units in order, side-major, i-tile-major, wave chunks fastest; kept
    // until the geometry changes.  (Group counts: one item per tile and chunk, all groups' blocks.)  part: which wave of how
    // many the units are (pipelined exchange; 0 of 1: all owned units)
    auto item_list = [&](const std::vector<uint32_t> &units, reo_ctx::ItemList &cache, int part, int nparts) -> int32_t {
        const int CW = 64 * RJ, QW = Q * (CJ / CW);
        const uint32_t nsides = wave ? 2u : 1u;
        const bool halves = wave && !wide && c->k1_half;  // k1w_pairs only: its last round's items are dealt as two halves each
        const uint64_t key[5] = {static_cast<uint64_t>(c->G) << 32 | static_cast<uint32_t>(c->Gp), static_cast<uint64_t>(halves ? 1 : 0) << 48 | static_cast<uint64_t>(RJ) << 32 | static_cast<uint32_t>(Q),
                                 static_cast<uint64_t>(c->world) << 32 | static_cast<uint32_t>(c->rank),
                                 static_cast<uint64_t>(nsides) << 56 | static_cast<uint64_t>(units.size()) << 24 |
                                     static_cast<uint64_t>(wave ? std::max(a.ce - a.cb, a.te - a.tb) : c->goff32[c->ngroups] / 32),
                                 static_cast<uint64_t>(sides) << 48 | static_cast<uint64_t>(part) << 32 | static_cast<uint32_t>(nparts)};
        if (!cache.buf.p || std::memcmp(key, cache.key, sizeof key) != 0) {
            // The list is a function of the geometry alone: a process-wide store keeps the last few as host vectors, so that a NEW context
            // on the same problem shape (the drop-in call makes one per identify_degs) does not build and sort it again (0.9 ms per
            // side at config 3, on the critical path of the pipelined upload).
            struct Stored { uint64_t key[7]; std::vector<uint32_t> units, items; };
            static std::mutex store_mu;
            static std::vector<Stored> *store = new std::vector<Stored>();
            const uint64_t skey[7] = {key[0], key[1], key[2], key[3], key[4], static_cast<uint64_t>(c->n_cus),
                                      static_cast<uint64_t>(c->k1_order) << 2 | static_cast<uint64_t>(big ? 1 : 0) << 1 | static_cast<uint64_t>(wide ? 1 : 0)};
            std::vector<uint32_t> items;
            bool have = false;
            {
                std::lock_guard<std::mutex> lk(store_mu);
                for (const Stored &st : *store)
                    if (std::memcmp(st.key, skey, sizeof skey) == 0 && st.units == units) { items = st.items; have = true; break; }
            }
            if (!have) {
            // Workgroup b runs on XCD b & 7.  An item goes to the list of XCD (wave chunk & 7), and an XCD walks its list
            // group by group of its chunks (as many pos chunks of one side as fit about 2.5 MB of its 4 MiB L2), inside a
            // group side-major, then i-tile-major, chunks fastest: the group's pos planes stay in that L2 while each tile
            // operand (32 rows x the side's blocks) streams through it once per group.  (Dealing the items of the
            // unit-by-unit order one at a time made every XCD touch every chunk and re-read every tile operand per unit:
            // 2.0 GB of L2 fills per launch at config 3, against 0.24 GB algorithmic.)  The lists are then levelled by
            // moving the surplus of the long ones -- their last items -- to the short ones, and interleaved.
            std::vector<uint32_t> lists[8];
            const int G = static_cast<int>(c->G);
            size_t total_items = 0;
            const int side_blocks = wave ? std::max(a.ce - a.cb, a.te - a.tb) : c->goff32[c->ngroups] / 32;
            const size_t chunk_side_bytes = static_cast<size_t>(CW) * std::max(side_blocks, 1) * 64;
            const int per_group = static_cast<int>(std::max<size_t>(1, (size_t(5) << 19) / chunk_side_bytes));  // chunks of one XCD per group
            uint32_t tile_block = 32;   // (order 2) i-tiles per block: their tile operands together about 1 MB, a power of two from 4 to 32
            while (tile_block > 4 && static_cast<size_t>(tile_block) * kTileI * std::max(side_blocks, 1) * (big ? 128 : 64) > (size_t(1) << 20)) tile_block >>= 1;
            for (uint32_t um : units)
                for (uint32_t side = 0; side < nsides; ++side) {
                    if (wave && !((sides >> side) & 1)) continue;
                    for (int t = 0; t < kUnitH; ++t)
                        for (int w = 0; w < QW; ++w) {
                            const int it = static_cast<int>(um & 0xFFFFu) * kUnitH + t, cw = static_cast<int>(um >> 16) * QW + w;
                            const int i0 = it * kTileI, jw = cw * CW;
                            if (i0 >= G || jw >= G || ((jw + CW - 1) >> 6) < (i0 >> 6)) continue;  // no pair i < j < G in it
                            lists[cw & 7].push_back(side << 31 | static_cast<uint32_t>(cw) << 16 | static_cast<uint32_t>(it));
                            ++total_items;
                        }
                }
            // order inside an XCD's list: chunk group, side, i-tile, chunk -- sorted as one 64-bit key per item (the comparator form,
            // with its two divisions per comparison, took 0.9 ms per side at config 3)
            std::vector<uint64_t> keys;
            for (auto &l : lists) {
                keys.resize(l.size());
                for (size_t q = 0; q < l.size(); ++q) {
                    const uint32_t x = l[q], cx = (x >> 16) & 0x7FFFu;
                    if (c->k1_order == 0)
                        keys[q] = static_cast<uint64_t>((cx >> 3) / static_cast<uint32_t>(per_group)) << 32 | static_cast<uint64_t>(x >> 31) << 31 |
                                  static_cast<uint64_t>(x & 0xFFFFu) << 15 | cx;
                    else if (c->k1_order == 1)   // i-tiles fastest inside a chunk: the mirror words of a chunk's genes (one 32-bit word per i-tile,
                                                 // neighbours in their table rows) reach L2 one after the other -- but every chunk re-reads every tile operand
                        keys[q] = static_cast<uint64_t>((cx >> 3) / static_cast<uint32_t>(per_group)) << 48 | static_cast<uint64_t>(x >> 31) << 47 |
                                  static_cast<uint64_t>(cx) << 16 | (x & 0xFFFFu);
                    else   // 2: blocks of TB consecutive i-tiles; inside a block chunk by chunk, the block's tiles fastest: TB mirror words in a row
                           // (TB x 4 bytes of a table line) while the block's tile operands (TB x 32 rows x the side's blocks x 64 B: about 1 MB) stay in L2
                        keys[q] = static_cast<uint64_t>((cx >> 3) / static_cast<uint32_t>(per_group)) << 48 | static_cast<uint64_t>(x >> 31) << 47 |
                                  static_cast<uint64_t>((x & 0xFFFFu) / tile_block) << 31 | static_cast<uint64_t>(cx) << 16 | (x & 0xFFFFu);
                }
                std::sort(keys.begin(), keys.end());
                for (size_t q = 0; q < l.size(); ++q) {
                    const uint64_t k = keys[q];
                    if (c->k1_order == 0) l[q] = static_cast<uint32_t>((k >> 31) & 1u) << 31 | static_cast<uint32_t>(k & 0x7FFFu) << 16 | static_cast<uint32_t>((k >> 15) & 0xFFFFu);
                    else l[q] = static_cast<uint32_t>((k >> 47) & 1u) << 31 | static_cast<uint32_t>((k >> 16) & 0x7FFFu) << 16 | static_cast<uint32_t>(k & 0xFFFFu);
                }
            }
            const size_t per = (total_items + 7) / 8;
            std::vector<uint32_t> surplus;
            for (auto &l : lists)
                while (l.size() > per) { surplus.push_back(l.back()); l.pop_back(); }
            for (auto &l : lists)
                while (l.size() < per && !surplus.empty()) { l.push_back(surplus.back()); surplus.pop_back(); }
            items.reserve(total_items);
            for (size_t k = 0; k < per; ++k)
                for (auto &l : lists)
                    if (k < l.size()) items.push_back(l[k]);
            // All items take the same time, so the resident waves (slots) work through the list in rounds; when the last
            // round fills at most half of the slots, its items are dealt as two half-height items each (rows 0-15 and 16-31
            // of the tile: bit 15 set, bit 14 = which half) and the launch ends half an item's time earlier -- 0.45 of a round
            // out of 16.45 at config 3; a shard of one eighth of the tiles has 2.06 rounds.  Both halves of an item stay
            // on the item's XCD (the tail is a multiple of 8 items).
            if (halves && !items.empty()) {
                const size_t slots = static_cast<size_t>(c->n_cus) * 4 * (big ? 2 : 3);
                const size_t left = items.size() % slots;
                if (left > 0 && left <= slots / 2) {
                    const size_t n = std::min(items.size(), (left + 7) / 8 * 8);
                    const std::vector<uint32_t> tail(items.end() - static_cast<ptrdiff_t>(n), items.end());
                    items.resize(items.size() - n);
                    for (uint32_t half = 0; half < 2; ++half)
                        for (uint32_t x : tail) items.push_back(x | 0x8000u | (half ? 0x4000u : 0u));
                }
            }
                std::lock_guard<std::mutex> lk(store_mu);
                if (store->size() >= 16) store->erase(store->begin());
                Stored st;
                std::memcpy(st.key, skey, sizeof skey);
                st.units = units; st.items = items;
                store->push_back(std::move(st));
            }   // (!have)
            int32_t rc2;
            if ((rc2 = cache.buf.ensure(std::max<size_t>(items.size(), 1)))) return rc2;
            if (!items.empty()) {
                REO_HIP_CHECK(hipMemcpyAsync(cache.buf.p, items.data(), items.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
                REO_HIP_CHECK(hipStreamSynchronize(c->stream));  // `items` is a local
            }
            cache.n = items.size();
            std::memcpy(cache.key, key, sizeof key);
        }
        return REO_OK;
    };
    auto dispatch = [&]() {   // the pair kernel(s) for a.items / c->k1_items_n on c->stream
        switch (plane_bits(c->G)) {
        case 12: launch_pair_kernels<12>(c, a, grid, shared, multi, plane_elems, wide); break;
        case 15: launch_pair_kernels<15>(c, a, grid, shared, multi, plane_elems, wide); break;
        case 16: launch_pair_kernels<16>(c, a, grid, shared, multi, plane_elems, wide); break;
        case 17: launch_big_pairs<17>(c, a, shared, multi, plane_elems, wide); break;
        default: launch_big_pairs<18>(c, a, shared, multi, plane_elems, wide); break;
        }
    };
    c->x_pipelined = false;
    // Several shards and a gather exchange: this shard's units are counted in WAVES on two alternating streams (the tail of
    // one wave's launch overlaps the head of the next), and as soon as wave w is done a third stream packs its forward
    // words, all-gathers the shards' packs of that wave and unpacks them, while wave w + 1 is being counted.  Every word of
    // the table belongs to one unit, the unpack kernels touch only words that carry another shard's bits, and the pair
    // kernel only writes words of its own units: the three streams never meet in a word.  (Wave form for two groups only;
    // everything else exchanges behind the pair kernel, as in round 3.)
    const int maxu_x = std::max(1, (static_cast<int>(c->units_all_host.size()) + std::max(c->world, 1) - 1) / std::max(c->world, 1));
    const int nwaves = (wave && sides == 3 && c->world > 1 && (c->comm || c->ag) && !c->in_multi && !c->k1_stamps) ? std::min({c->x_waves, 8, maxu_x}) : 1;   // (a reo_create_multi context hands its packs to the leader itself: comm.hip)
    if (nwaves > 1) {
        const int mw = (maxu_x + nwaves - 1) / nwaves;   // slots per wave: the same on every shard
        const size_t uw = static_cast<size_t>(kUnitRows) * kPlanes * (static_cast<size_t>(Q) * CJ / 32);
        if ((rc = c->xsend.ensure(uw * mw * nwaves)) || (rc = c->xrecv.ensure(uw * mw * nwaves * c->world))) return rc;
        if (!c->xs) {
            for (int q = 0; q < 2; ++q) REO_HIP_CHECK(handle_stream(&c->k1s[q], 0));
            REO_HIP_CHECK(handle_stream(&c->xs, 0));
            REO_HIP_CHECK(handle_event(&c->ev_fork, 0));
            REO_HIP_CHECK(handle_event(&c->ev_x, 0));
            for (int q = 0; q < 2; ++q) REO_HIP_CHECK(handle_event(&c->ev_k1_join[q], 0));
            for (int q = 0; q < 8; ++q) REO_HIP_CHECK(handle_event(&c->ev_k1[q], 0));
        }
        // the waves' item lists (uploads synchronise c->stream: before anything is forked)
        std::vector<std::vector<uint32_t>> wunits(nwaves);
        for (size_t m = 0; m < units.size(); ++m) wunits[std::min<size_t>(m / mw, nwaves - 1)].push_back(units[m]);
        for (int w = 0; w < nwaves; ++w)
            if ((rc = item_list(wunits[w], c->k1_wave_items[w], w, nwaves))) return rc;
        // the exchange kernels read the unit list and the geometry of THIS build
        c->k1_cj = CJ; c->k1_q = Q;
        XArgs xa;
        if ((rc = exchange_args(c, xa, 0, mw))) return rc;
        if (!c->table_prezeroed) REO_HIP_CHECK(hipMemsetAsync(c->table.p, 0, static_cast<size_t>(c->G) * kPlanes * c->Wp * sizeof(uint32_t), c->stream));
        c->table_prezeroed = false;
        c->last_k1_shared = 0;
        tic(c, 1);
        // From the fork to the join nothing returns: whatever fails in between, the two pair-kernel streams and the exchange stream
        // are joined into c->stream afterwards, so that everything queued on them is ordered before whatever the caller does next
        // on the context's stream -- its waits, and reo_destroy's frees (an early return here used to leave pair and exchange
        // kernels, or an RCCL all-gather, queued on streams that nothing waited for).  The first failure is kept in xrc.
        int32_t xrc = REO_OK;
        auto hip_ok = [&](hipError_t e, const char *what) {
            if (e == hipSuccess || xrc) return e == hipSuccess;
            set_error("%s failed: %s (pipelined exchange)", what, hipGetErrorString(e));
            xrc = e == hipErrorOutOfMemory ? REO_ENOMEM : REO_EHIP;
            return false;
        };
        hipStream_t main_stream = c->stream;
        bool forked = hip_ok(hipEventRecord(c->ev_fork, c->stream), "hipEventRecord(fork)");
        for (int q = 0; q < 2 && forked; ++q) forked = hip_ok(hipStreamWaitEvent(c->k1s[q], c->ev_fork, 0), "hipStreamWaitEvent(fork)");
        if (forked) forked = hip_ok(hipStreamWaitEvent(c->xs, c->ev_fork, 0), "hipStreamWaitEvent(fork)");
        for (int w = 0; w < nwaves && !xrc; ++w) {
            hipStream_t ks = c->k1s[w & 1];
            c->stream = ks;   // (the launchers below enqueue on the context's stream)
            a.items = c->k1_wave_items[w].buf.p; c->k1_items_n = c->k1_wave_items[w].n;
            if (c->k1_items_n) dispatch();
            c->stream = main_stream;
            if (!hip_ok(hipGetLastError(), "pair kernel launch")) break;
            if (!hip_ok(hipEventRecord(c->ev_k1[w], ks), "hipEventRecord(wave)") || !hip_ok(hipStreamWaitEvent(c->xs, c->ev_k1[w], 0), "hipStreamWaitEvent(wave)")) break;
            const int m0 = w * mw, mc = std::max(0, std::min(mw, maxu_x - m0));
            if (mc == 0) continue;
            uint32_t *send = c->xsend.p + uw * mw * w, *recv = c->xrecv.p + uw * mw * w * c->world;
            const int64_t bytes = static_cast<int64_t>(uw) * mc * sizeof(uint32_t);
            if ((xrc = launch_pack_units(c, m0, mc, send, c->xs))) break;
            if (c->comm) { if ((xrc = comm_allgather(c, send, recv, bytes, c->xs)) < 0) break; xrc = REO_OK; }
            else {
                const int hrc = c->ag(send, recv, bytes, c->xs, c->ag_user);  // stream-ordered on the exchange stream
                if (hrc) { set_error("all-gather hook failed with %d", hrc); xrc = REO_ECOMM; break; }
            }
            if ((xrc = launch_expand_units(c, m0, mc, recv, c->xs))) break;
        }
        // join: the pair kernels first (their makespan is the K1 stage time), then the exchange stream (what is left of it: the
        // exposed part of the exchange).  A join that cannot even be queued falls back to host waits for the side streams.
        bool joined = true;
        for (int q = 0; q < 2; ++q)
            if (hipEventRecord(c->ev_k1_join[q], c->k1s[q]) != hipSuccess || hipStreamWaitEvent(c->stream, c->ev_k1_join[q], 0) != hipSuccess) joined = false;
        toc(c);
        tic(c, 6);
        if (hipEventRecord(c->ev_x, c->xs) != hipSuccess || hipStreamWaitEvent(c->stream, c->ev_x, 0) != hipSuccess) joined = false;
        toc(c);
        if (!joined) {
            for (int q = 0; q < 2; ++q) (void)hipStreamSynchronize(c->k1s[q]);
            (void)hipStreamSynchronize(c->xs);
            if (!xrc) { set_error("joining the streams of the pipelined exchange failed"); xrc = REO_EHIP; }
        }
        if (xrc) return xrc;
        c->x_pipelined = true;
        return REO_OK;
    }
    if (wave || (wcounts && !c->gc_valid) || wmulti) {
        reo_ctx::ItemList &il = c->k1_wave_items[sides == 3 ? 0 : sides];   // (a side's list keeps its own slot: the two sides of a pipelined upload alternate)
        const auto w0 = std::chrono::steady_clock::now();
        if ((rc = item_list(units, il, 0, 1))) return rc;
        if (c->debug_passes) fprintf(stderr, "  launch_k1 sides %d: item list (%zu items) ready after %.0f us\n", sides, il.n, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count());
        a.items = il.buf.p; c->k1_items_n = il.n;
        if (prepare) return REO_OK;   // (the unit map and this side's item list are on the device; nothing is launched)
        if (range) {
            // the item list above is the side's (its key holds the side's WHOLE block count): every range of the side runs the same
            // items, so blockIdx.x names the same (tile, chunk) in each and the park slots line up
            const int side = sides == 1 ? 0 : 1, sb = side ? a.tb : a.cb, se = side ? a.te : a.ce;
            if (range->b0 < 0 || range->b1 <= range->b0 || sb + range->b1 > se || (range->first != (range->b0 == 0)) || (range->last != (sb + range->b1 == se))) {
                set_error("a range of sample blocks outside its side"); return REO_EINVAL;
            }
            if (!(range->first && range->last)) {
                if ((rc = c->k1_park[side].ensure(std::max<size_t>(il.n, 1) * kParkSlot)) || (rc = c->k1_park_ge[side].ensure(std::max<size_t>(il.n, 1)))) return rc;
                a.park = c->k1_park[side].p; a.park_ge = c->k1_park_ge[side].p;
                a.park_mode = (range->first ? 0 : 1) | (range->last ? 0 : 2);
            }
            (side ? a.tb : a.cb) = sb + range->b0;
            (side ? a.te : a.ce) = sb + range->b1;
        }
        if (c->k1_stamps) REO_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&a.stamps), (std::max<size_t>(c->k1_items_n, 1) * 4 + 2) * sizeof(unsigned long long)));
    }
    // (every reader of the table -- the passes, the pack, a sum hook's element count, the scan of a hook's table -- works on
    //  G * kPlanes * Wp words, which is also what the transform's early clear covers; a grow-only buffer may be larger)
    if (!keep_table) {
        if (!c->table_prezeroed) REO_HIP_CHECK(hipMemsetAsync(c->table.p, 0, static_cast<size_t>(c->G) * kPlanes * c->Wp * sizeof(uint32_t), c->stream));
        c->table_prezeroed = false;
    }
    if (units.empty()) return REO_OK;
    c->last_k1_shared = shared ? 1 : 0;
    if (a.stamps) k_time_mark<<<1, 64, 0, c->stream>>>(a.stamps + c->k1_items_n * 4);
    tic(c, 1);
    dispatch();
    toc(c);
    if (a.stamps) k_time_mark<<<1, 64, 0, c->stream>>>(a.stamps + c->k1_items_n * 4 + 1);
    REO_HIP_CHECK(hipGetLastError());
    if (a.stamps) {  // diagnostic: where an item's time goes (100 MHz marks)
        std::vector<unsigned long long> h(c->k1_items_n * 4 + 2);
        REO_HIP_CHECK(hipMemcpyAsync(h.data(), a.stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        REO_HIP_CHECK(hipStreamSynchronize(c->stream));
        double pro = 0, loop = 0, emit = 0;
        unsigned long long t0 = ~0ull, t1 = 0;
        for (size_t i = 0; i < c->k1_items_n; ++i) {
            pro += static_cast<double>(h[4 * i + 1] - h[4 * i]); loop += static_cast<double>(h[4 * i + 2] - h[4 * i + 1]);
            emit += static_cast<double>(h[4 * i + 3] - h[4 * i + 2]);
            t0 = std::min(t0, h[4 * i]); t1 = std::max(t1, h[4 * i + 3]);
        }
        const double n = static_cast<double>(c->k1_items_n) * 100.0;  // marks per microsecond
        fprintf(stderr, "[reo] K1 wave items %zu: prologue %.2f us, count loop %.2f us, classification %.2f us per item; first start to last end %.3f ms\n",
                c->k1_items_n, pro / n, loop / n, emit / n, static_cast<double>(t1 - t0) / 1e5);
        fprintf(stderr, "[reo]   a launch before it ended %.1f us before the first item began; one after it began %.1f us after the last item ended\n",
                static_cast<double>(static_cast<long long>(t0 - h[c->k1_items_n * 4])) / 100.0, static_cast<double>(static_cast<long long>(h[c->k1_items_n * 4 + 1] - t1)) / 100.0);
        // the launch in 24 slices of time: items in flight (average) and the count loop's duration of the items that started in the slice
        constexpr int kSl = 24;
        const double span = static_cast<double>(t1 - t0) + 1.0;
        double busy[kSl] = {}, dur[kSl] = {};
        size_t started[kSl] = {};
        for (size_t i = 0; i < c->k1_items_n; ++i) {
            const double b = static_cast<double>(h[4 * i] - t0), e = static_cast<double>(h[4 * i + 3] - t0);
            const int s0 = static_cast<int>(b / span * kSl);
            started[s0]++; dur[s0] += static_cast<double>(h[4 * i + 2] - h[4 * i + 1]);
            for (int s = s0; s < kSl; ++s) {
                const double lo = span * s / kSl, hi = span * (s + 1) / kSl;
                if (e <= lo) break;
                busy[s] += (std::min(e, hi) - std::max(b, lo)) / (hi - lo);
            }
        }
        fprintf(stderr, "[reo]   items in flight by slice:");
        for (int s = 0; s < kSl; ++s) fprintf(stderr, " %.0f", busy[s]);
        fprintf(stderr, "\n[reo]   loop us of items started in slice:");
        for (int s = 0; s < kSl; ++s) fprintf(stderr, " %.0f", started[s] ? dur[s] / static_cast<double>(started[s]) / 100.0 : 0.0);
        fprintf(stderr, "\n");
        (void)hipFree(a.stamps);
    }
    return REO_OK;
}


int64_t exchange_unit_words(const reo_ctx *c) { return static_cast<int64_t>(kUnitRows) * kPlanes * (c->k1_q * c->k1_cj / 32); }
int32_t exchange_units_per_rank(const reo_ctx *c)
{
    const int total = static_cast<int>(c->units_all_host.size()), world = std::max(c->world, 1);
    return std::max(1, (total + world - 1) / world);
}

// the unit list of the last launch_k1 on the device (uploaded when it changes)
static int32_t upload_units_all(reo_ctx *c)
{
    int32_t rc;
    const size_t total = c->units_all_host.size();
    if ((rc = c->units_all.ensure(std::max<size_t>(total, 1)))) return rc;
    if (total && (c->units_all.p != c->units_all_uploaded || c->units_all_host != c->units_all_dev)) {
        REO_HIP_CHECK(hipMemcpyAsync(c->units_all.p, c->units_all_host.data(), total * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        REO_HIP_CHECK(hipStreamSynchronize(c->stream));  // (the vector may be rebuilt by the next launch_k1 while the copy is in flight)
        c->units_all_dev = c->units_all_host;
        c->units_all_uploaded = c->units_all.p;
    }
    return REO_OK;
}

static int32_t exchange_args(reo_ctx *c, XArgs &a, int m0, int mcnt)
{
    const int32_t rc = upload_units_all(c);
    if (rc) return rc;
    a.table = c->table.p; a.units = c->units_all.p;
    a.G = static_cast<int>(c->G); a.Wp = c->Wp; a.Wc = c->k1_q * c->k1_cj / 32;
    a.total = static_cast<int>(c->units_all_host.size()); a.world = std::max(c->world, 1); a.rank = c->rank; a.maxu = exchange_units_per_rank(c);
    a.m0 = m0; a.mcnt = mcnt < 0 ? a.maxu : mcnt;
    return REO_OK;
}

// slots [m0, m0 + mcnt) of this shard -> send (mcnt < 0: all slots -> c->xsend, on the context's stream)
int32_t launch_pack_units(reo_ctx *c, int m0, int mcnt, uint32_t *send, hipStream_t st)
{
    XArgs a;
    int32_t rc = exchange_args(c, a, m0, mcnt);
    if (rc) return rc;
    const size_t uw = static_cast<size_t>(exchange_unit_words(c));
    x_pack<<<dim3(static_cast<unsigned>((uw + 255) / 256), a.mcnt), 256, 0, st ? st : c->stream>>>(a, send ? send : c->xsend.p);
    REO_HIP_CHECK(hipGetLastError());
    return REO_OK;
}

// recv = every shard's pack of slots [m0, m0 + mcnt) -> the table: the others' words and their mirrors
int32_t launch_expand_units(reo_ctx *c, int m0, int mcnt, const uint32_t *recv, hipStream_t st)
{
    XArgs a;
    int32_t rc = exchange_args(c, a, m0, mcnt);
    if (rc) return rc;
    if (a.world < 2) return REO_OK;  // (a communicator of one rank: its own pack came back)
    const size_t uw = static_cast<size_t>(exchange_unit_words(c));
    if (!st) st = c->stream;
    if (!recv) recv = c->xrecv.p;
    x_expand_fwd<<<dim3(static_cast<unsigned>((uw + 255) / 256), a.mcnt * a.world), 256, 0, st>>>(a, recv);
    x_expand_mirror<<<dim3(a.Wc, kPlanes, a.mcnt * a.world), 256, 0, st>>>(a, recv);
    REO_HIP_CHECK(hipGetLastError());
    return REO_OK;
}

int32_t launch_counts(reo_ctx *c, int64_t i0, int64_t i1, int64_t j0, int64_t j1, uint16_t *d_gt, uint16_t *d_eq)
{
    dim3 grid(static_cast<unsigned>((j1 - j0 + 255) / 256), static_cast<unsigned>(i1 - i0));
    if (c->G > 65535)
        k1_counts_big<<<grid, 256, 0, c->stream>>>(c->pos.p, c->lo.p, c->hi.p, c->Gp, plane_bits(c->G), c->goff_dev.p, c->ngroups,
                                                   static_cast<int>(i0), static_cast<int>(i1), static_cast<int>(j0), static_cast<int>(j1), d_gt, d_eq);
    else
        k1_counts<<<grid, 256, 0, c->stream>>>(c->pos.p, c->lo.p, c->hi.p, c->Gp, plane_bits(c->G), c->goff_dev.p, c->ngroups,
                                               static_cast<int>(i0), static_cast<int>(i1), static_cast<int>(j0), static_cast<int>(j1), d_gt, d_eq);
    REO_HIP_CHECK(hipGetLastError());
    return REO_OK;
}

// 0 = consistent, 1 = not (synchronises the stream)
int32_t launch_check_table(reo_ctx *c, int *bad)
{
    *bad = 0;
    int32_t rc = c->check_flag.ensure(1);  // (kept in the context: an allocation per build would put a device-wide hipFree on the exchange path)
    if (rc) return rc;
    REO_HIP_CHECK(hipMemsetAsync(c->check_flag.p, 0, sizeof(int32_t), c->stream));
    const int G = static_cast<int>(c->G);
    k_check_table<<<(G + 3) / 4, 256, 0, c->stream>>>(c->table.p, G, c->Wp, c->check_flag.p);
    int32_t h = 0;
    REO_HIP_CHECK(hipMemcpyAsync(&h, c->check_flag.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
    REO_HIP_CHECK(hipStreamSynchronize(c->stream));
    *bad = h;
    return REO_OK;
}

int32_t launch_decode(reo_ctx *c, int64_t i0, int64_t i1, int64_t j0, int64_t j1, uint8_t *d_code)
{
    dim3 grid(static_cast<unsigned>((j1 - j0 + 255) / 256), static_cast<unsigned>(i1 - i0));
    k_decode<<<grid, 256, 0, c->stream>>>(c->table.p, c->Wp, static_cast<int>(i0), static_cast<int>(i1),
                                          static_cast<int>(j0), static_cast<int>(j1), d_code);
    REO_HIP_CHECK(hipGetLastError());
    return REO_OK;
}

int32_t launch_pack_ref(reo_ctx *c, const uint8_t *d_bytes, uint32_t *d_bits)
{
    k_pack_ref<<<c->Gp / 256, 256, 0, c->stream>>>(d_bytes, c->Gp, d_bits);
    REO_HIP_CHECK(hipGetLastError());
    return REO_OK;
}

// cycle watch (kl_head found the reference set of pass T equal to that of pass T - period; api.hip decides how much to skip): the
// state of pass T is the state of pass T + skip when skip is a multiple of the period -- and even, for the buffers indexed by
// the parity of the pass.  The skipped passes' trace entries (:418) are the last period's.
__global__ void k_cycle_skip(IterState *st, IterState *host_st, int32_t *trace, int skip, int period)
{
    const int T = st->passes;
    for (int u = T + static_cast<int>(threadIdx.x); u < T + skip; u += static_cast<int>(blockDim.x)) {
        const int src = T - period + (u - T) % period;   // (>= 0: the snapshot was taken at pass T - period)
        trace[2 * u] = trace[2 * src]; trace[2 * u + 1] = trace[2 * src + 1];
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    st->passes = T + skip;
    if (st->raw_pass >= 0) st->raw_pass += skip;
    st->i_iter += skip;
    st->cyc_period = -1;
    if (host_st) { host_st->passes = T + skip; host_st->cyc_period = -1; }
}

int32_t launch_cycle_skip(reo_ctx *c, int skip, int period)
{
    if (skip < 0 || period < 1 || skip % period != 0 || (skip & 1)) { set_error("launch_cycle_skip: skip %d, period %d", skip, period); return REO_EINVAL; }
    k_cycle_skip<<<1, 256, 0, c->stream>>>(c->state.p, c->state_mirror ? c->host_state : nullptr, c->trace.p, skip, period);
    REO_HIP_CHECK(hipGetLastError());
    return REO_OK;
}

int32_t launch_iter_init(reo_ctx *c, const uint8_t *host_ref, const IterState *host_state)
{
    IterInitArgs a;
    a.host_ref = host_ref; a.host_state = reinterpret_cast<const uint32_t *>(host_state);
    a.refbytes = c->refbytes[0].p; a.refbits = c->refbits[0].p; a.state = reinterpret_cast<uint32_t *>(c->state.p);
    a.zero[0] = reinterpret_cast<uint32_t *>(c->result.p); a.zero_n[0] = static_cast<size_t>(c->G) * 15 * 2;
    a.zero[1] = reinterpret_cast<uint32_t *>(c->modes.p); a.zero_n[1] = c->modes.n;
    a.zero[2] = reinterpret_cast<uint32_t *>(c->hist.p); a.zero_n[2] = c->hist.n;
    a.G = static_cast<int>(c->G); a.Gp = c->Gp; a.state_words = static_cast<int>(sizeof(IterState) / 4);
    k_iter_init<<<c->Gp / 256, 256, 0, c->stream>>>(a);
    REO_HIP_CHECK(hipGetLastError());
    return REO_OK;
}

// reo_tally: whole-table scan with the mask of parity 0, then the nine tallies per gene
int32_t launch_tally(reo_ctx *c, int nref)
{
    const int G = static_cast<int>(c->G);
    tic(c, 2);
    k2_scan<<<(G + 3) / 4, 256, 0, c->stream>>>(c->table.p, reinterpret_cast<const uint4 *>(c->refbits[0].p), G, c->Wp, c->raw.p);
    toc(c);
    c->t_ms[4] += 1.0;
    k3_tallies<<<(G + 255) / 256, 256, 0, c->stream>>>(c->raw.p, c->refbytes[0].p, nref, G, c->cont.p);
    REO_HIP_CHECK(hipGetLastError());
    return REO_OK;
}

static IterArgs iter_args(reo_ctx *c, int replay)
{
    IterArgs a;
    a.st = c->state.p; a.table = c->table.p;
    a.G = static_cast<int>(c->G); a.Gp = c->Gp; a.Wp = c->Wp;
    a.n_iter = c->it_n_iter; a.n_conv = c->it_n_conv; a.a0 = c->it_a0; a.b0 = c->it_b0;
    a.pval_deg = c->it_pval_deg; a.padj_deg = c->it_padj_deg;
    for (int t = 0; t < 2; ++t) { a.refbits[t] = c->refbits[t].p; a.refbytes[t] = c->refbytes[t].p; }
    a.raw = c->raw.p; a.delta_list = c->delta_list.p; a.result = c->result.p;
    const int nchunk = (a.G + kSortChunk - 1) / kSortChunk;
    a.chunk_v = c->chunk_v.p; a.chunk_i = c->chunk_i.p;
    a.chunk_spl = c->chunk_v.p + static_cast<size_t>(nchunk) * kSortChunk;  // [nchunk][kSortChunk / 32]
    a.sorted_d = c->sorted_d.p; a.sorted_spl = c->sorted_d.p + ((c->G + 63) / 64) * 64;  // [ceil(G / 64)]
    a.sorted_p = c->sorted_p.p; a.rank_s = c->rank_s.p; a.rank_a = c->rank_a.p;
    a.part = c->part.p; a.blockmin = c->blockmin.p; a.scal = c->scal.p;
    a.trace = c->trace.p; a.modes = nullptr;
    a.cand = c->cand.p; a.hist = c->hist.p; a.hist_stride = static_cast<int>(c->hist.n / (3 * kHistParts)); a.mrank = c->mrank.p;
    a.replay = replay; a.k2_idx = 0;
    a.clist = c->clist.p; a.olist = c->olist.p; a.band = c->light_band; a.hist_below = c->hist_below; a.snap = c->snap.p; a.xcc_local = c->xcc_local;
    // (no light passes -- switched off, or given up by the running call: the sorting path then leaves need_full set, else its
    //  launches would wait for light passes that nobody enqueues)
    a.window = c->light_window; a.light_min_g = (c->it_no_light || c->light_mode == 0) ? 0x7FFFFFFF : c->light_min_g;
    a.stamps = reinterpret_cast<unsigned long long *>(c->scal.p + 32);
    a.host_st = c->state_mirror ? c->host_state : nullptr;  // (hipHostMalloc memory: the same address on the device)
    return a;
}

// One pass on the sorting path (K2 + six K3 kernels); every kernel returns at once unless the device-side state
// asks for such a pass.  replay: recompute the output columns of the last executed pass from its tallies.
int32_t launch_full_pass(reo_ctx *c, bool replay)
{
    IterArgs a = iter_args(c, replay ? 1 : 0);
    const int G = a.G, nb = (G + 255) / 256;
    const int nchunk = (G + kSortChunk - 1) / kSortChunk;
    const int nmerge = (nchunk * kSortChunk + kMergeThreads / kMergeLanes - 1) / (kMergeThreads / kMergeLanes);
    if (!replay) {
        if (c->k2_idx < static_cast<int>(c->modes.n)) { a.modes = c->modes.p; a.k2_idx = c->k2_idx; }
        ++c->k2_idx;
        tic(c, 2);
        k2_tally<<<(G + 3) / 4, 256, 0, c->stream>>>(a);
        toc(c);
        c->t_ms[4] += 1.0;
    }
    k3_derive<<<nb, 256, 0, c->stream>>>(a);
    k3_sort_chunks<<<nchunk, kSortChunk, 0, c->stream>>>(a);
    const size_t spl_bytes = static_cast<size_t>(nchunk) * kSplit * sizeof(double);
    if (spl_bytes > (size_t(48) << 10))  // (near the top of the gene range the splitter table passes what a launch gets by default)
        REO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k3_merge_rank), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(spl_bytes)));
    k3_merge_rank<<<nmerge, kMergeThreads, spl_bytes, c->stream>>>(a, nchunk);
    k3_abs_rank<<<nb, 256, static_cast<size_t>((G + 63) / 64) * sizeof(double), c->stream>>>(a, nmerge);
    k3_bh_local<<<(G + 1023) / 1024, 1024, 0, c->stream>>>(a);
    k3_finalize<<<c->Gp / 256, 256, 0, c->stream>>>(a);
    REO_HIP_CHECK(hipGetLastError());
    return REO_OK;
}

// Light passes until the state stops them (convergence, n_iter, a pass that needs the sorting path): one persistent launch.
int32_t launch_light_persistent(reo_ctx *c)
{
    const IterArgs a = iter_args(c, 0);
    const unsigned nwg = static_cast<unsigned>(a.Gp / 256);  // every mask byte, padding included, has its thread
    REO_HIP_CHECK(hipMemsetAsync(c->gridbar.p, 0, sizeof(unsigned), c->stream));
    REO_HIP_CHECK(hipMemsetAsync(c->lstate.p, 0, 2 * sizeof(LightSlot), c->stream));  // the counters of both pass parities
    kl_persist<<<nwg, 256, 0, c->stream>>>(a, c->lstate.p, c->gridbar.p);
    REO_HIP_CHECK(hipGetLastError());
    return REO_OK;
}

// A batch of light passes in the two-launch form: nlight x (kl_head, kl_rank) and the tail that ends the last pass (the default),
// or nlight launches of kl_one and its tail (REO_LIGHT=3).
int32_t launch_light_batch(reo_ctx *c, int nlight)
{
    const IterArgs a = iter_args(c, 0);
    const int nb = (a.G + 255) / 256, nbp = a.Gp / 256;  // (the head writes every mask byte, padding included)
    if (nlight < 1 || nlight > kLightBatch) { set_error("light batch of %d passes", nlight); return REO_EINVAL; }
    REO_HIP_CHECK(hipMemsetAsync(c->lstate.p, 0, static_cast<size_t>(nlight + 1) * sizeof(LightSlot), c->stream));  // the slots this batch writes
    if (c->it_light_form == 3) {
        // launch b fills histogram b % 3 and clears (b + 1) % 3: the first one of a batch is cleared here
        REO_HIP_CHECK(hipMemsetAsync(c->hist.p, 0, static_cast<size_t>(kHistParts) * a.hist_stride * sizeof(int32_t), c->stream));
        for (int b = 0; b < nlight; ++b) kl_one<false><<<nbp, 256, 0, c->stream>>>(a, c->lstate.p, b);
        kl_one<true><<<nbp, 256, 0, c->stream>>>(a, c->lstate.p, nlight);
        REO_HIP_CHECK(hipGetLastError());
        return REO_OK;
    }
    // the first mask step of a batch reads all BH ranks (the launch before it had no cut to make a list around); later
    // ones read the list of genes near the cut
    for (int b = 0; b < nlight; ++b) {
        if (b < 2) kl_head<false, false><<<nbp, 256, 0, c->stream>>>(a, c->lstate.p, b);
        else kl_head<false, true><<<nbp, 256, 0, c->stream>>>(a, c->lstate.p, b);
        kl_rank<<<nb, 256, 0, c->stream>>>(a, c->lstate.p, b);
    }
    if (nlight < 2) kl_head<true, false><<<nbp, 256, 0, c->stream>>>(a, c->lstate.p, nlight);
    else kl_head<true, true><<<nbp, 256, 0, c->stream>>>(a, c->lstate.p, nlight);
    REO_HIP_CHECK(hipGetLastError());
    return REO_OK;
}

int32_t xcc_selftest(reo_ctx *c, int *ok)
{
    *ok = 0;
    DevBuf<int32_t> buf;
    // the verdict is a property of the device, not of the context: checked once per device and process
    static std::mutex mu;
    static int verdict[64];   // 0 unknown, 1 failed, 2 passed
    {
        std::lock_guard<std::mutex> lk(mu);
        if (c->device >= 0 && c->device < 64 && verdict[c->device]) { *ok = verdict[c->device] == 2; return REO_OK; }
    }
    int32_t rc = buf.ensure(kHistParts * 64 + kHistParts);
    if (rc) return rc;
    REO_HIP_CHECK(hipMemsetAsync(buf.p, 0xFF, buf.n * sizeof(int32_t), c->stream));  // garbage first: the zeroing launch has to win
    k_xcc_selftest<<<1024, 256, 0, c->stream>>>(buf.p, buf.p + kHistParts * 64);      // dirty every XCD's L2 with atomics on the lines
    k_xcc_selftest_zero<<<1024, 256, 0, c->stream>>>(buf.p, buf.p + kHistParts * 64);  // plain stores from arbitrary XCDs (kl_head's zeroing)
    k_xcc_selftest<<<1024, 256, 0, c->stream>>>(buf.p, buf.p + kHistParts * 64);      // L2-local atomics in the next launch (kl_rank)
    std::vector<int32_t> h(buf.n);
    REO_HIP_CHECK(hipMemcpyAsync(h.data(), buf.p, buf.n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    REO_HIP_CHECK(hipStreamSynchronize(c->stream));
    int64_t waves = 0;
    bool good = true;
    for (int x = 0; x < kHistParts; ++x) {
        const int32_t w = h[kHistParts * 64 + x];
        waves += w;
        for (int l = 0; l < 64; ++l) good = good && h[x * 64 + l] == 16 * w;
    }
    *ok = good && waves == 1024 * 4 ? 1 : 0;
    {
        std::lock_guard<std::mutex> lk(mu);
        if (c->device >= 0 && c->device < 64) verdict[c->device] = *ok ? 2 : 1;
    }
    return REO_OK;
}

int32_t light_min_genes() { return kLightMinG; }
int32_t light_window() { return kWindow; }

int32_t launch_mccullagh(reo_ctx *c, const int32_t *d_cont, int64_t n, double *d_out)
{
    k_mccullagh<<<static_cast<unsigned>((n + 255) / 256), 256, 0, c->stream>>>(d_cont, n, d_out);
    REO_HIP_CHECK(hipGetLastError());
    return REO_OK;
}

}  // namespace reo
