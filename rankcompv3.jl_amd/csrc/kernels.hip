// Hand-written gfx950 kernels of the REO hot path.
//
//   K1  k1_pairs    pair compare -> per-group counts -> stable-REO class ->
//                   4 bit planes per ordered pair      (src/RankCompV3.jl:363-392)
//       k1_group_counts + k1_classify: the same for one-vs-rest over more than two groups
//                   (:375-390) -- every group counted once, counts kept in HBM, one cheap
//                   classification per comparison
//   K2  k2_tally    class table x reference mask -> per-gene tallies   (:403)
//       (delta form: the same launch updates the counters from the rows of the genes whose mask bit changed)
//   K3  k3_*        McCullagh test, trimmed std, normal p, BH, new mask, loop control (:404-425,225-259)
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
//   table u32 [G][4][Wp]  bit planes cL cH tL tH of row i: bit j of plane cL is
//                   set iff pair (i,j) is "i<j stable" in ctrl (ic==1), cH iff
//                   ic==3, tL/tH likewise for treat.  4 bits per ORDERED pair,
//                   the diagonal is all-zero (like the reference's R, :363).
//
// Wave = 64 lanes everywhere; no warp-32 idiom is used.
#include <algorithm>

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
// MI355X (tools/microbench_bitop.hip, tools/k1b_proto.hip): v_bitop3_b32 with three VGPR sources issues at full
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
};

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
template <int RI, int RJ, typename F>
__device__ __forceinline__ void emit_side(const K1Args &a, int i0, int jl, int bi, int lane, int pl, int hi_thr, int lo_thr, F val)
{
#pragma unroll
    for (int r = 0; r < RJ; ++r) {
        const int j = jl + 64 * r;
        const int bj = __builtin_amdgcn_readfirstlane(j >> 6);  // wave-uniform, and the compiler should know it
        if ((bj << 6) >= a.Gp || bj < bi) continue;
        const int d = j - i0;                          // rows i0+ii with ii < d are above the diagonal
        const bool near = (bj << 6) - i0 < RI;         // wave-uniform: some lane has d < 32
        const unsigned long long lanes_ok = __ballot(j < a.G);
        uint32_t wL = 0, wH = 0;
        uint32_t fLlo = 0, fLhi = 0, fHlo = 0, fHhi = 0;  // lane ii: forward words of row i0+ii
#pragma unroll
        for (int ii = RI - 1; ii >= 0; --ii) {
            const int n = val(r, ii);
            const unsigned long long ok = near ? (__ballot(d > ii) & lanes_ok) : lanes_ok;
            const unsigned long long mH = __ballot(n >= hi_thr) & ok;
            const unsigned long long mL = __ballot(n <= lo_thr) & ok & ~mH;
            shift_in(wH, mH);
            shift_in(wL, mL);
            write_lane(fHlo, static_cast<uint32_t>(mH), ii);
            write_lane(fHhi, static_cast<uint32_t>(mH >> 32), ii);
            write_lane(fLlo, static_cast<uint32_t>(mL), ii);
            write_lane(fLhi, static_cast<uint32_t>(mL >> 32), ii);
        }
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
            if (!diag) {
                row[0] = wH; row[a.Wp] = wL;
            } else {
                if (wH) atomicOr(row, wH);
                if (wL) atomicOr(row + a.Wp, wL);
            }
        }
    }
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
__device__ __forceinline__ void delta_genes(const uint32_t *__restrict__ table, int G, int Wp, const uint32_t *__restrict__ list,
                                            int n, int32_t *__restrict__ raw)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= G) return;
    const int w = i >> 5, sh = i & 31;
    int d[kRaw] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int e = 0; e < n; ++e) {
        const uint32_t ent = list[e];  // wave-uniform
        const uint32_t *row = table + static_cast<size_t>(ent >> 1) * kPlanes * Wp + w;
        const int sgn = (ent & 1u) ? 1 : -1;
        // pair (i, j) seen from gene i: cL(i,j) = cH(j,i), cH(i,j) = cL(j,i), likewise for the treat side
        const int cl = (row[Wp] >> sh) & 1, ch = (row[0] >> sh) & 1;
        const int tl = (row[3 * Wp] >> sh) & 1, th = (row[2 * Wp] >> sh) & 1;
        d[0] += sgn * cl; d[1] += sgn * ch; d[2] += sgn * tl; d[3] += sgn * th;
        d[4] += sgn * (cl & tl); d[5] += sgn * (cl & th); d[6] += sgn * (ch & tl); d[7] += sgn * (ch & th);
    }
    int4 *o = reinterpret_cast<int4 *>(raw + static_cast<size_t>(i) * kRaw);
    int4 a = o[0], b = o[1];
    a.x += d[0]; a.y += d[1]; a.z += d[2]; a.w += d[3];
    b.x += d[4]; b.y += d[5]; b.z += d[6]; b.w += d[7];
    o[0] = a; o[1] = b;
}

// The K2 stage of one pass: one launch, the device picks the form.  slot < 0: always a full scan.
__global__ __launch_bounds__(256) void k2_tally(const IterState *__restrict__ st, const uint32_t *__restrict__ table,
                                                const uint4 *__restrict__ refbits, int G, int Wp,
                                                int32_t *__restrict__ raw, int slot,
                                                const uint32_t *__restrict__ list, int32_t *__restrict__ modes)
{
    if (st->done) return;
    const int n = slot >= 0 ? st->delta_cnt[slot] : kDeltaMax + 1;
    const bool full = n > kDeltaMax;
    if (modes && blockIdx.x == 0 && threadIdx.x == 0) modes[st->passes] = full ? 1 : 0;  // for the stage timers
    if (full) tally_rows(reinterpret_cast<const uint4 *>(table), refbits, G, Wp / 4, raw);
    else if (static_cast<int>(blockIdx.x) * 256 < G) delta_genes(table, G, Wp, list, n, raw);
}

// ---------------------------------------------------------------------------
// K3.  McCullagh's test for a 3x3 table, closed form of :225-259.
//   N = [[a b][b d]], n = (a, d), R = (R1, R2); singular iff a*d == b*b (exact
//   integer test, equivalent to abs(det(N)) <= eps() for integer N, :242).
__device__ void mccullagh3(const int32_t *n, double *out)
{
    const long long n12 = n[1], n13 = n[2], n21 = n[3], n23 = n[5], n31 = n[6], n32 = n[7];
    const long long a = n12 + n13 + n21 + n31;  // N11  (:232)
    const long long b = n13 + n31;              // N12
    const long long d = n13 + n23 + n31 + n32;  // N22
    const long long R1 = n12 + n13, R2 = n13 + n23;  // :239
    const long long det = a * d - b * b;
    if (det == 0) { out[0] = 1.0; out[1] = out[2] = out[3] = out[4] = 0.0; return; }
    const double fa = static_cast<double>(a), fd = static_cast<double>(d), fdet = static_cast<double>(det);
    const double w1 = static_cast<double>(d * (a - b)) / fdet;  // omega2 = inv(N) n  (:246)
    const double w2 = static_cast<double>(a * (d - b)) / fdet;
    const double nu = 1.0 / (fa * w1 + fd * w2);                // :247
    const double r1 = static_cast<double>(R1), r2 = static_cast<double>(R2);
    const double d1 = (fa * w1 * nu) * log((r1 + 0.5) / (fa - r1 + 0.5)) +
                      (fd * w2 * nu) * log((r2 + 0.5) / (fd - r2 + 0.5));  // :248-249
    const double A = w1 * r1 + w2 * r2, B = w1 * (fa - r1) + w2 * (fd - r2);
    const double d2 = log((0.5 + A) / (0.5 + B));               // :250
    const double v1 = 4.0 * (1.0 + 0.25 * d1 * d1) * nu;        // :251
    const double v2 = 4.0 * (1.0 + 0.25 * d2 * d2) * nu;        // :252
    const double se = sqrt((v1 + v2) * 0.5);                    // :253
    const double z1 = d1 / se;                                  // :254
    double p = erfc(fabs(z1) * 0.70710678118654752440);         // 2*min(cdf, ccdf), :255
    out[0] = p > 1.0 ? 1.0 : p;
    out[1] = d1; out[2] = d2; out[3] = se; out[4] = z1;
}

__global__ void k_mccullagh(const int32_t *__restrict__ cont, int64_t n, double *__restrict__ out)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t c[9];
    for (int t = 0; t < 9; ++t) c[t] = cont[i * 9 + t];
    double o[5];
    mccullagh3(c, o);
    for (int t = 0; t < 5; ++t) out[i * 5 + t] = o[t];
}

// raw counters -> 9 tallies (:403) [-> McCullagh -> result columns 3..15 (:404-405)]
// nref comes from the
// device-side iteration state so that passes can be enqueued back to back.
// one gene: raw counters -> 9 tallies (:403) [-> McCullagh -> result columns 3..15 (:404-405)]; returns delta1
__device__ __forceinline__ double derive_gene(const int32_t *__restrict__ raw, const uint8_t *__restrict__ refbytes,
                                              int nref, int G, int i, int32_t *__restrict__ cont,
                                              double *__restrict__ result, bool with_stats)
{
    const int4 r0 = reinterpret_cast<const int4 *>(raw)[2 * i], r1 = reinterpret_cast<const int4 *>(raw)[2 * i + 1];
    const int cLt = r0.x, cHt = r0.y, tLt = r0.z, tHt = r0.w, LL = r1.x, LH = r1.y, HL = r1.z, HH = r1.w;
    const int total = nref - (refbytes[i] ? 1 : 0);  // the diagonal is never set (:363,385)
    int32_t c[9];
    c[0] = LL; c[2] = LH; c[6] = HL; c[8] = HH;
    c[1] = cLt - LL - LH;
    c[7] = cHt - HL - HH;
    c[3] = tLt - LL - HL;
    c[5] = tHt - LH - HH;
    c[4] = total - (cLt + cHt + c[3] + c[5]);
    if (cont) {
#pragma unroll
        for (int t = 0; t < 9; ++t) cont[static_cast<size_t>(i) * 9 + t] = c[t];
    }
    if (!with_stats) return 0.0;
    double o[5];
    mccullagh3(c, o);
    const size_t Gs = G;
    result[i] = o[0];
    result[Gs + i] = 1.0;
#pragma unroll
    for (int t = 0; t < 9; ++t) result[(2 + t) * Gs + i] = static_cast<double>(c[t]);
    result[11 * Gs + i] = o[1]; result[12 * Gs + i] = o[2];
    result[13 * Gs + i] = o[3]; result[14 * Gs + i] = o[4];
    return o[1];
}

// stand-alone form for reo_tally (tallies only)
__global__ __launch_bounds__(256) void k3_derive(const IterState *__restrict__ st, const int32_t *__restrict__ raw,
                                                 const uint8_t *__restrict__ refbytes, int G,
                                                 int32_t *__restrict__ cont, double *__restrict__ result,
                                                 int with_stats, IterState *__restrict__ stw, int slot_next)
{
    if (st->done) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0 && stw) stw->delta_cnt[slot_next] = 0;  // k3_finalize of this pass fills that list
    if (i >= G) return;
    derive_gene(raw, refbytes, st->nref, G, i, cont, result, with_stats != 0);
}

// Sum over the 256 threads of a workgroup, the same bits in every thread and every workgroup: an
// xor-butterfly inside each wave (a + b == b + a, so all lanes of a wave agree at every step), then the
// four wave sums in a fixed order.  Two barriers (the second lets `red` be reused at once).
__device__ __forceinline__ double block_sum_256(double v, double *red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const double r = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return r;
}

// ---- ranking: the sort of :409 and the order BH needs (:413) ---------------
// (1) k3_sort_chunks: bitonic sort of 2048-gene chunks in LDS by (delta1, gene);
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
__device__ __forceinline__ void cmpx(double &v, uint32_t &i, double pv, uint32_t pi, bool keep_min)
{
    const bool take = keep_min ? (pv < v) : (pv > v);
    if (take) { v = pv; i = pi; }
}

// Bitonic sort of one kSortChunk-gene chunk, four consecutive elements per thread in registers
// (kSortChunk / 4 threads).  Exchange distances 1 and 2 stay inside the thread, distances 4..128 go
// through wave shuffles, and only the stages with distance >= 256 (partner in another wave) use
// LDS + a barrier.
constexpr int kSortE = 1;  // elements per thread (K3 per 128 passes at 20 000 genes: 4 -> 6.58 ms, 2 -> 6.56, 1 -> 6.49)
constexpr int kSortThreads = kSortChunk / kSortE;

__global__ __launch_bounds__(kSortThreads) void k3_sort_chunks(const IterState *__restrict__ st,
                                                               const double *__restrict__ d1, int G,
                                                               double *__restrict__ cv, uint16_t *__restrict__ ci,
                                                               double *__restrict__ splitters)
{
    if (st->done) return;
    constexpr int E = kSortE;
    __shared__ double sv[kSortChunk];
    __shared__ uint16_t si[kSortChunk];
    const int t = threadIdx.x, base = blockIdx.x * kSortChunk;
    const int x0 = E * t;  // element index of this thread's first slot
    double v[E];
    uint32_t id[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        v[e] = base + x0 + e < G ? d1[base + x0 + e] : INFINITY;  // padding sorts to the end of the last chunk
        id[e] = x0 + e;
    }
    for (int k = 2; k <= kSortChunk; k <<= 1) {
        const bool up = (x0 & k) == 0;  // for k >= E all slots of a thread share the direction
        for (int j = k >> 1; j >= 64 * E; j >>= 1) {
            const int m = j / E;  // partner thread distance (>= 64: another wave)
            __syncthreads();
#pragma unroll
            for (int e = 0; e < E; ++e) { sv[x0 + e] = v[e]; si[x0 + e] = static_cast<uint16_t>(id[e]); }
            __syncthreads();
            const int px = E * (t ^ m);
            const bool keep_min = ((t & m) == 0) == up;
#pragma unroll
            for (int e = 0; e < E; ++e) cmpx(v[e], id[e], sv[px + e], si[px + e], keep_min);
        }
        for (int j = (k >> 1) < 32 * E ? (k >> 1) : 32 * E; j >= E; j >>= 1) {
            const int m = j / E;  // 1..32: inside the wave
            const bool keep_min = ((t & m) == 0) == up;
            double pv[E];
            uint32_t pi[E];
#pragma unroll
            for (int e = 0; e < E; ++e) { pv[e] = __shfl_xor(v[e], m, 64); pi[e] = __shfl_xor(id[e], m, 64); }
#pragma unroll
            for (int e = 0; e < E; ++e) cmpx(v[e], id[e], pv[e], pi[e], keep_min);
        }
#pragma unroll
        for (int j = E >> 1; j >= 1; j >>= 1) {  // partners inside the thread
            if (j >= k) continue;
#pragma unroll
            for (int a = 0; a < E; ++a) {
                if (a & j) continue;
                const int b = a + j;
                const bool up_ = ((x0 + a) & k) == 0;
                if ((v[a] > v[b]) == up_) {
                    const double tv = v[a]; v[a] = v[b]; v[b] = tv;
                    const uint32_t ti = id[a]; id[a] = id[b]; id[b] = ti;
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) { cv[base + x0 + e] = v[e]; ci[base + x0 + e] = static_cast<uint16_t>(id[e]); }
    // every 32nd element once more, packed: k3_merge_rank stages these into LDS with contiguous loads
    if ((x0 & 31) == 0) splitters[blockIdx.x * (kSortChunk / 32) + (x0 >> 5)] = v[0];
}

// A search in a sorted array a[0..n) for the count of elements that sort before v (`<` when strict, `<=`
// otherwise), in two levels: a coarse search over every 2^LOG-th element staged in LDS narrows it to a
// half-open index range [l, h) of fewer than 2^LOG elements, then LOG halving steps in global memory finish.
// The two levels are separate calls so that a thread with several searches can run their global steps in
// lockstep: the loads of one step are independent, so k searches cost one round trip per step, not k.
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

__global__ __launch_bounds__(kMergeThreads) void k3_merge_rank(const IterState *__restrict__ st, const double *__restrict__ cv,
                                                     const uint16_t *__restrict__ ci, int G, int nchunk,
                                                     int a0, int b0, uint32_t *__restrict__ rs,
                                                     double *__restrict__ sorted_d, double *__restrict__ part,
                                                     const double *__restrict__ splitters, double *__restrict__ sorted_spl)
{
    if (st->done) return;
    __shared__ double slice[kMergeThreads / kMergeLanes];
    __shared__ double spl[(65536 / kSortChunk) * kSplit];  // [chunk][kSplit]: 2048 splitters at most (G <= 65535)
    for (int t = threadIdx.x; t < nchunk * kSplit; t += kMergeThreads) spl[t] = splitters[t];  // = cv[chunk][32 m], packed
    __syncthreads();
    const int e = blockIdx.x * (kMergeThreads / kMergeLanes) + threadIdx.x / kMergeLanes;  // element (position in the chunked array)
    const int sub = threadIdx.x % kMergeLanes;
    const int c = e / kSortChunk, p = e % kSortChunk;
    const bool live = e < nchunk * kSortChunk;
    const int gene = live ? c * kSortChunk + ci[e] : G;
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
        const int rank = p + count;
        rs[gene] = rank;
        sorted_d[rank] = v;
        if ((rank & 63) == 0) sorted_spl[rank >> 6] = v;  // packed splitters for k3_abs_rank
        in = rank >= a0 && rank <= b0;  // inside the 5 %-95 % slice of :411
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
        if (threadIdx.x == 0) { part[3 * blockIdx.x] = n; part[3 * blockIdx.x + 1] = mean; part[3 * blockIdx.x + 2] = m2; }
    }
}

// |delta1| ranks (= rank of pval ascending), then se = std of the 5 %-95 % slice of the sorted delta1
// (n-1 estimator, :409-411) from the per-block moments of k3_merge_rank (every workgroup combines them
// the same way: n = sum n_b, mean = sum n_b mean_b / n, M2 = sum [M2_b + n_b (mean_b - mean)^2], fixed-
// order tree sums), then pval = pvalue(Normal(0,se), delta1, tail=:both) (:412) -> column 1, and into
// rank order for the BH step.
__global__ __launch_bounds__(256) void k3_abs_rank(const IterState *__restrict__ st, const double *__restrict__ d1,
                                                   const uint32_t *__restrict__ rs,
                                                   const double *__restrict__ sorted_d,
                                                   const double *__restrict__ sorted_spl, int G,
                                                   const double *__restrict__ part, int npart,
                                                   uint32_t *__restrict__ ra, double *__restrict__ pval,
                                                   double *__restrict__ sorted_p, double *__restrict__ scal)
{
    if (st->done) return;
    __shared__ double red[256];
    __shared__ double spl[1024];  // every 64th element of the sorted vector (G <= 65535)
    const int nspl = (G + 63) >> 6;
    for (int t = threadIdx.x; t < nspl; t += 256) spl[t] = sorted_spl[t];  // = sorted_d[64 t], packed
    double nb = 0.0, sb = 0.0;
    for (int t = threadIdx.x; t < npart; t += 256) { const double n_ = part[3 * t]; nb += n_; sb += n_ * part[3 * t + 1]; }
    const double n = block_sum_256(nb, red);  // (also the barrier that publishes spl)
    const double mean = block_sum_256(sb, red) / n;
    double qb = 0.0;
    for (int t = threadIdx.x; t < npart; t += 256) {
        const double n_ = part[3 * t], d_ = part[3 * t + 1] - mean;
        qb += part[3 * t + 2] + n_ * d_ * d_;
    }
    const double se = sqrt(block_sum_256(qb, red) / (n - 1.0));
    if (blockIdx.x == 0 && threadIdx.x == 0) scal[0] = se;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= G) return;
    const double v = d1[i];
    const int r = rs[i];
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
    ra[i] = rank;
    double p;
    if (se == 0.0) {
        p = 0.0;  // Normal(0,0): cdf/ccdf degenerate to a step, the smaller tail is 0
    } else {
        const double z = fabs(v) / se;
        p = erfc(z * 0.70710678118654752440);
        p = p > 1.0 ? 1.0 : p;
    }
    pval[i] = p;
    sorted_p[rank] = p;
}

// Benjamini-Hochberg step-up (:413), part 1: p_(r) * (n/r) and the reverse
// cumulative minimum inside blocks of 1024 ranks (in place) + the block minima.
__global__ __launch_bounds__(1024) void k3_bh_local(const IterState *__restrict__ st, double *__restrict__ sorted_p,
                                                    int G, double *__restrict__ blockmin)
{
    if (st->done) return;
    __shared__ double wmin[16];
    const int r = blockIdx.x * 1024 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double v = r < G ? sorted_p[r] * (static_cast<double>(G) / static_cast<double>(r + 1)) : INFINITY;
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
    if (r < G) sorted_p[r] = v;
    if (threadIdx.x == 0) blockmin[blockIdx.x] = v;
}

// BH part 2 + mask update.  padj -> column 2 (:416); non-DEG mask inds (:417)
// as bytes and bits for the next pass; the last workgroup to finish applies
// the loop control of :418-424 to the device-side iteration state.
__global__ __launch_bounds__(256) void k3_finalize(IterState *__restrict__ st, const double *__restrict__ pval,
                                                   const double *__restrict__ sufmin,
                                                   const double *__restrict__ blockmin,
                                                   const uint32_t *__restrict__ ra, int G, int Gp,
                                                   double pval_deg, double padj_deg, int n_conv,
                                                   double *__restrict__ padj, uint8_t *__restrict__ nbytes,
                                                   uint32_t *__restrict__ nbits, int32_t *__restrict__ trace,
                                                   const uint8_t *__restrict__ obytes, uint32_t *__restrict__ dlist,
                                                   int slot_next)
{
    if (st->done) return;
    __shared__ double tail[65];
    const int nb = (G + 1023) / 1024;
    __shared__ double bm[64];
    if (static_cast<int>(threadIdx.x) < nb) bm[threadIdx.x] = blockmin[threadIdx.x];
    __syncthreads();
    if (static_cast<int>(threadIdx.x) <= nb) {  // tail[b] = minimum over the blocks after b (one thread per b)
        double run = INFINITY;
        for (int k = threadIdx.x + 1; k < nb; ++k) { const double m = bm[k]; run = m < run ? m : run; }
        tail[threadIdx.x] = run;
    }
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    bool ind = false;
    if (i < G) {
        const uint32_t r = ra[i];
        double q = sufmin[r];
        const double t = tail[r >> 10];
        q = t < q ? t : q;
        q = q < 1.0 ? q : 1.0;
        padj[i] = q;
        ind = !(pval[i] <= pval_deg && q <= padj_deg);
    }
    if (i < Gp) nbytes[i] = ind ? 1 : 0;
    // genes whose mask bit changes: the next pass can update its tallies from their rows alone (delta_genes)
    const bool changed = i < G && ind != (obytes[i] != 0);
    const unsigned long long cm = __ballot(changed);
    if (cm) {
        int basepos = 0;
        if ((threadIdx.x & 63) == 0) basepos = atomicAdd(&st->delta_cnt[slot_next], __popcll(cm));
        basepos = __shfl(basepos, 0, 64);
        const int at = basepos + __popcll(cm & ((1ULL << (threadIdx.x & 63)) - 1ULL));
        if (changed && at < kDeltaMax) dlist[at] = (static_cast<uint32_t>(i) << 1) | (ind ? 1u : 0u);
    }
    const unsigned long long m = __ballot(ind);
    __shared__ int wave_nn[4];
    if ((threadIdx.x & 63) == 0) {
        if (i < Gp) {
            nbits[i >> 5] = static_cast<uint32_t>(m);
            nbits[(i >> 5) + 1] = static_cast<uint32_t>(m >> 32);
        }
        wave_nn[threadIdx.x >> 6] = __popcll(m);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // one atomic per workgroup, not per wave: 313 serialised updates of one word were a third of this kernel
        const int nn_blk = wave_nn[0] + wave_nn[1] + wave_nn[2] + wave_nn[3];
        if (nn_blk) atomicAdd(&st->nn_acc, nn_blk);
        __threadfence();
        const int t = atomicAdd(&st->ticket, 1);
        if (t == static_cast<int>(gridDim.x) - 1) {
            __threadfence();
            const int nn = atomicAdd(&st->nn_acc, 0);  // sum(inds), :417-418
            const int pass = st->passes;
            trace[2 * pass] = G - nn;
            trace[2 * pass + 1] = nn;
            st->passes = pass + 1;
            const int diff = st->nref - nn;
            if ((diff < 0 ? -diff : diff) < n_conv) {
                st->done = 1;       // :419-422
            } else {
                st->i_iter += 1;    // :423
                st->nref = nn;      // ref_gene_vec = inds, :424
            }
            st->nn_acc = 0;
            st->ticket = 0;
        }
    }
}

}  // namespace

// ------------------------------------------------------------------ launchers

// bits needed for every number the pair kernel compares: positions 0..G-1 and band ends up to G
static int plane_bits(int64_t G) { return G <= 4095 ? 12 : (G <= 32767 ? 15 : 16); }

template <int NB>
static void launch_pair_kernels(reo_ctx *c, const K1Args &a, unsigned grid, bool shared, bool multi, size_t plane_elems)
{
    if (shared) {
        if (!c->gc_valid) {
            if (c->has_ties) k1_group_counts<NB, true><<<grid, 256, 0, c->stream>>>(a, c->gcounts.p, plane_elems);
            else k1_group_counts<NB, false><<<grid, 256, 0, c->stream>>>(a, c->gcounts.p, plane_elems);
            c->gc_valid = true;
        }
        if (c->has_ties) k1_classify<kRJTies><<<grid, 256, 0, c->stream>>>(a, c->gcounts.p, plane_elems);
        else k1_classify<kRJ><<<grid, 256, 0, c->stream>>>(a, c->gcounts.p, plane_elems);
    } else if (multi) {
        if (c->has_ties) k1_pairs<NB, true, true><<<grid, 256, 0, c->stream>>>(a);
        else k1_pairs<NB, false, true><<<grid, 256, 0, c->stream>>>(a);
    } else {
        if (c->has_ties) k1_pairs<NB, true, false><<<grid, 256, 0, c->stream>>>(a);
        else k1_pairs<NB, false, false><<<grid, 256, 0, c->stream>>>(a);
    }
}

int32_t launch_k1(reo_ctx *c, int k)
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
    const int RJ = c->has_ties ? kRJTies : kRJ;  // genes j per lane
    const int CJ = kTileJ * RJ;
    const int NJ = (c->Gp + CJ - 1) / CJ, NIT = c->Gp / kTileI;
    const size_t chunk_bytes = static_cast<size_t>(CJ) * (c->goff32[c->ngroups] / 32) * 64;
    const int Q = chunk_bytes * 4 <= (2u << 20) ? 4 : (chunk_bytes * 2 <= (2u << 20) ? 2 : 1);
    const int NP = (NJ + Q - 1) / Q;
    std::vector<uint32_t> units;
    int64_t owned = 0, total = 0;
    uint32_t gu = 0;
    for (int p = 0; p < NP; ++p) {
        const int ni = std::min(NIT, (CJ / kTileI) * Q * (p + 1));  // i-tiles that reach this panel's columns
        for (int r = 0; r * kUnitH < ni; ++r, ++gu) {
            const bool mine = c->world == 1 || static_cast<int>(gu % c->world) == c->rank;
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
    int32_t rc;
    if ((rc = c->unit_map.ensure(std::max<size_t>(units.size(), 1)))) return rc;
    if (!units.empty()) {
        REO_HIP_CHECK(hipMemcpyAsync(c->unit_map.p, units.data(), units.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        REO_HIP_CHECK(hipStreamSynchronize(c->stream));  // `units` is a local: the copy must have read it before any return below
    }
    a.unit_map = c->unit_map.p;
    REO_HIP_CHECK(hipMemsetAsync(c->table.p, 0, c->table.n * sizeof(uint32_t), c->stream));
    if (units.empty()) return REO_OK;
    const unsigned grid = static_cast<unsigned>((units.size() + 7) / 8 * 8 * kUnitH * Q);
    // > 2 groups: count every group once, then classify per comparison -- if the planes fit
    const size_t plane_elems = static_cast<size_t>(c->Gp) * c->Gp;
    bool shared = multi && c->share_counts;
    if (shared && !c->gc_valid) {
        size_t free_b = 0, total_b = 0;
        REO_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
        const size_t need = plane_elems * (c->ngroups + 1) * sizeof(uint16_t);
        const size_t have = c->gcounts.n * sizeof(uint16_t);
        if (need > have && need - have + (size_t(4) << 30) > free_b) shared = false;  // keep 4 GiB for everything else
        if (shared && (rc = c->gcounts.ensure(plane_elems * (c->ngroups + 1)))) return rc;
    }
    c->last_k1_shared = shared ? 1 : 0;
    tic(c, 1);
    switch (plane_bits(c->G)) {
    case 12: launch_pair_kernels<12>(c, a, grid, shared, multi, plane_elems); break;
    case 15: launch_pair_kernels<15>(c, a, grid, shared, multi, plane_elems); break;
    default: launch_pair_kernels<16>(c, a, grid, shared, multi, plane_elems); break;
    }
    toc(c);
    REO_HIP_CHECK(hipGetLastError());
    return REO_OK;
}

int32_t launch_counts(reo_ctx *c, int64_t i0, int64_t i1, int64_t j0, int64_t j1, uint16_t *d_gt, uint16_t *d_eq)
{
    dim3 grid(static_cast<unsigned>((j1 - j0 + 255) / 256), static_cast<unsigned>(i1 - i0));
    k1_counts<<<grid, 256, 0, c->stream>>>(c->pos.p, c->lo.p, c->hi.p, c->Gp, plane_bits(c->G), c->goff_dev.p, c->ngroups,
                                           static_cast<int>(i0), static_cast<int>(i1), static_cast<int>(j0), static_cast<int>(j1), d_gt, d_eq);
    REO_HIP_CHECK(hipGetLastError());
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

int32_t launch_k2(reo_ctx *c, const uint32_t *d_refbits, int slot, bool allow_delta)
{
    const int G = static_cast<int>(c->G);
    tic(c, 2);
    k2_tally<<<(G + 3) / 4, 256, 0, c->stream>>>(c->state.p, c->table.p, reinterpret_cast<const uint4 *>(d_refbits), G, c->Wp,
                                                 c->raw.p, allow_delta ? slot : -1,
                                                 allow_delta ? c->delta_list.p + static_cast<size_t>(slot) * c->Gp : nullptr,
                                                 allow_delta ? c->modes.p : nullptr);
    toc(c);
    c->t_ms[4] += 1.0;
    REO_HIP_CHECK(hipGetLastError());
    return REO_OK;
}

int32_t launch_derive(reo_ctx *c, const uint8_t *d_refbytes, int with_stats)
{
    const int G = static_cast<int>(c->G);
    k3_derive<<<(G + 255) / 256, 256, 0, c->stream>>>(c->state.p, c->raw.p, d_refbytes, G, c->cont.p, c->result.p,
                                                      with_stats, nullptr, 0);
    REO_HIP_CHECK(hipGetLastError());
    return REO_OK;
}

int32_t launch_stats(reo_ctx *c, int cur, double pval_deg, double padj_deg, int n_conv, int64_t a, int64_t b)
{
    const int G = static_cast<int>(c->G);
    const int nb = (G + 255) / 256;
    double *res = c->result.p;
    const double *d1 = res + 11 * c->G;
    const int nchunk = (G + kSortChunk - 1) / kSortChunk;
    const int nmerge = (nchunk * kSortChunk + kMergeThreads / kMergeLanes - 1) / (kMergeThreads / kMergeLanes);
    k3_derive<<<nb, 256, 0, c->stream>>>(c->state.p, c->raw.p, c->refbytes[cur].p, G, nullptr, res, 1, c->state.p, 1 - cur);
    double *chunk_spl = c->chunk_v.p + static_cast<size_t>(nchunk) * kSortChunk;  // [nchunk][kSortChunk / 32]
    double *sorted_spl = c->sorted_d.p + ((c->G + 63) / 64) * 64;               // [ceil(G / 64)]
    k3_sort_chunks<<<nchunk, kSortThreads, 0, c->stream>>>(c->state.p, d1, G, c->chunk_v.p, c->chunk_i.p, chunk_spl);
    k3_merge_rank<<<nmerge, kMergeThreads, 0, c->stream>>>(c->state.p, c->chunk_v.p, c->chunk_i.p, G, nchunk,
                                                           static_cast<int>(a - 1), static_cast<int>(b - 1),
                                                           c->rank_s.p, c->sorted_d.p, c->part.p, chunk_spl, sorted_spl);
    k3_abs_rank<<<nb, 256, 0, c->stream>>>(c->state.p, d1, c->rank_s.p, c->sorted_d.p, sorted_spl, G, c->part.p, nmerge,
                                           c->rank_a.p, res, c->sorted_p.p, c->scal.p);
    k3_bh_local<<<(G + 1023) / 1024, 1024, 0, c->stream>>>(c->state.p, c->sorted_p.p, G, c->blockmin.p);
    k3_finalize<<<c->Gp / 256, 256, 0, c->stream>>>(c->state.p, res, c->sorted_p.p, c->blockmin.p, c->rank_a.p, G,
                                                    c->Gp, pval_deg, padj_deg, n_conv, res + c->G,
                                                    c->refbytes[1 - cur].p, c->refbits[1 - cur].p, c->trace.p,
                                                    c->refbytes[cur].p,
                                                    c->delta_list.p + static_cast<size_t>(1 - cur) * c->Gp, 1 - cur);
    REO_HIP_CHECK(hipGetLastError());
    return REO_OK;
}

int32_t launch_mccullagh(reo_ctx *c, const int32_t *d_cont, int64_t n, double *d_out)
{
    k_mccullagh<<<static_cast<unsigned>((n + 255) / 256), 256, 0, c->stream>>>(d_cont, n, d_out);
    REO_HIP_CHECK(hipGetLastError());
    return REO_OK;
}

}  // namespace reo
