// Per-sample rank/band transform.
//
// The reference compares raw values pair by pair with is_greater
// (/root/reference/src/RankCompV3.jl:71-77): tie iff abs(x - y) < 0.1, else
// x > y.  For one sample, sort the genes by value; fl(x - y) is monotone in y,
// so the genes tied with a gene form a contiguous band [lo, hi] around its
// position p in that order, and every gene outside the band is ordered by
// position.  Hence for genes i, j of one sample
//     tie(i, j)            <=>  lo_i <= pos_j <= hi_i
//     x_i > x_j, not tied  <=>  pos_j <  lo_i
// which turns the float (0.1-band) and the integer (equality) comparators
// into the same two 16-bit integer compares for the pair kernel.  The band
// edges are found with the reference's own predicate evaluated in the input's
// arithmetic, so the result is exact, not approximate.
//
// Output for the pair kernel (kernels.hip): the three 16-bit numbers of every (gene, sample) leave as BIT
// PLANES over blocks of 32 samples (t_slice), because [pos_j < lo_i] for 32 samples at once is a borrow chain
// of one v_bitop3_b32 per bit.  Every group is padded to whole 32-sample blocks; padding samples have
// lo = hi = 0, which no position is below.
#include <sched.h>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#ifdef REO_WITH_ROCPRIM   // tools only (A/B against the library's segmented sort): the shipped library has no library call on this path
#include <rocprim/rocprim.hpp>
#endif

#include <algorithm>
#include <chrono>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <limits>
#include <mutex>
#include <numeric>
#include <thread>
#include <type_traits>

#include "reo_internal.h"

namespace reo {

namespace {

// NaN in the expression matrix (the only value refused; +-Inf are ranked as the reference compares them, see Codec<double>)
constexpr const char *kNaNMessage =
    "expression matrix contains NaN: is_greater (src/RankCompV3.jl:71-77) answers false for every comparison with a NaN, so a NaN gene "
    "would be below every later gene and above every earlier one -- an outcome of the row order, not an ordering; drop or impute such "
    "rows first (+-Inf are accepted)";

template <class T>
struct Codec;

template <>
struct Codec<double> {
    __device__ static uint64_t enc(double x)
    {
        uint64_t u = __double_as_longlong(x);
        return (u >> 63) ? ~u : (u | 0x8000000000000000ULL);
    }
    __device__ static double dec(uint64_t k)
    {
        uint64_t u = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFULL) : ~k;
        return __longlong_as_double(u);
    }
    __device__ static bool tie(double x, double y) { return fabs(x - y) < 0.1; }
    // is_greater (/root/reference/src/RankCompV3.jl:71-77) on infinities: abs(Inf - Inf) = NaN is not < 0.1 and Inf > Inf is false, so
    // EQUAL infinities are neither tied nor greater -- the pair (i, j), i < j, counts as "i not greater" in that sample, every time, no
    // coin -- while an infinity against anything else compares as usual.  In rank space: the -Inf genes take the lowest positions and
    // the +Inf genes the highest, each in gene order, with a tie band of the gene alone.  The codes make exactly that of the generic
    // machinery: gene g's infinity gets a code of its own inside the code space of the NaNs (which never reach a ranking: refused),
    // ascending in g; dec() of such a code is a NaN, so the reference's predicate ties it with nothing, and no finite value's band
    // reaches it (abs(x - Inf) = Inf).  log(0) = -Inf is an ordinary value of log-transformed tables.
    static constexpr uint64_t kPosInf = 0xFFF0000000000000ULL, kNegInf = 0x000FFFFFFFFFFFFFULL;   // enc(+Inf), enc(-Inf)
    __device__ static uint64_t enc(double x, uint32_t g)
    {
        const uint64_t k = enc(x);
        return k == kPosInf ? k + g : (k == kNegInf ? k - (static_cast<uint32_t>(kMaxGenes) - g) : k);   // (g <= kMaxGenes - 1)
    }
    // NaN is refused: is_greater(NaN, y) = (NaN > y) = false and is_greater(x, NaN) = false for every x, so a NaN gene would be
    // "below" every later gene and "above" every earlier one -- an outcome that depends on the row order and is no ordering.
    __device__ static bool ordered(double x) { return x == x; }
};

template <>
struct Codec<int64_t> {
    __device__ static uint64_t enc(int64_t x) { return static_cast<uint64_t>(x) ^ 0x8000000000000000ULL; }
    __device__ static int64_t dec(uint64_t k) { return static_cast<int64_t>(k ^ 0x8000000000000000ULL); }
    // abs(x - y) < 0.1 on Int64 <=> x == y
    __device__ static bool tie(int64_t x, int64_t y) { return x == y; }
    __device__ static uint64_t enc(int64_t x, uint32_t) { return enc(x); }
    __device__ static bool ordered(int64_t) { return true; }
};

#ifdef REO_WITH_ROCPRIM   // the segmented-sort form of the transform (rounds 1-4): built for A/B runs only (make ROCPRIM=1)
template <class T, class IdxT>
__global__ __launch_bounds__(256) void t_keys(const T *__restrict__ X, int64_t ld,
                                              const int32_t *__restrict__ colmap, int G, int cb0,
                                              uint64_t *__restrict__ keys, IdxT *__restrict__ idx,
                                              int32_t *__restrict__ bad, unsigned long long *__restrict__ varbits)
{
    int g = blockIdx.x * 256 + threadIdx.x;
    int c = blockIdx.y;
    uint64_t diff = 0;
    if (g < G) {
        T x = X[static_cast<int64_t>(g) + static_cast<int64_t>(colmap[cb0 + c]) * ld];
        if (!Codec<T>::ordered(x)) atomicOr(bad, 1);
        size_t o = static_cast<size_t>(c) * G + g;
        const uint64_t k = Codec<T>::enc(x, static_cast<uint32_t>(g));
        keys[o] = k;
        idx[o] = static_cast<IdxT>(g);
        diff = k ^ Codec<T>::enc(X[0], 0u);  // bits in which any key differs from one fixed key
    }
    // only the key bits that vary anywhere need sorting: OR-reduce them (wave, then one atomic per wave)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) diff |= __shfl_xor(diff, o, 64);
    // most waves find their bits already recorded: look before touching the one shared word
    if ((threadIdx.x & 63) == 0 && (diff & ~__hip_atomic_load(varbits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0)
        atomicOr(varbits, diff);
}

template <class T, class IdxT>   // IdxT: uint16_t up to 65 535 genes, uint32_t above (positions and gene indices alike)
__global__ __launch_bounds__(256) void t_bands(const uint64_t *__restrict__ keys,
                                               const IdxT *__restrict__ idx, int G, int Gp, int cb0,
                                               const int32_t *__restrict__ slots,
                                               IdxT *__restrict__ pos, IdxT *__restrict__ lo,
                                               IdxT *__restrict__ hi, int32_t *__restrict__ anytie)
{
    int p = blockIdx.x * 256 + threadIdx.x;
    int c = blockIdx.y;
    if (p >= G) return;
    const uint64_t *col = keys + static_cast<size_t>(c) * G;
    T v = Codec<T>::dec(col[p]);
    // first position of the band: smallest q <= p with tie(v_q, v)
    int l = p;
    if (p > 0 && Codec<T>::tie(Codec<T>::dec(col[p - 1]), v)) {
        int a = 0, b = p - 1;  // tie holds at b; search the first tied position
        while (a < b) {
            int m = (a + b) >> 1;
            if (Codec<T>::tie(Codec<T>::dec(col[m]), v)) b = m; else a = m + 1;
        }
        l = a;
    }
    int h = p;
    if (p + 1 < G && Codec<T>::tie(Codec<T>::dec(col[p + 1]), v)) {
        int a = p + 1, b = G - 1;  // tie holds at a; search the last tied position
        while (a < b) {
            int m = (a + b + 1) >> 1;
            if (Codec<T>::tie(Codec<T>::dec(col[m]), v)) a = m; else b = m - 1;
        }
        h = a;
    }
    if (l != p || h != p) {
        if (*anytie == 0) atomicOr(anytie, 1);
    }
    int g = idx[static_cast<size_t>(c) * G + p];
    const int slot = slots[cb0 + c];  // sample slot in the group-padded order
    const size_t o = static_cast<size_t>(slot) * Gp + g;
    pos[o] = static_cast<IdxT>(p);
    lo[o] = static_cast<IdxT>(l);
    hi[o] = static_cast<IdxT>(h + 1);
}

#endif  // REO_WITH_ROCPRIM

// One workgroup per sample, everything in LDS: when the varying key bits fit 31 bits and the genes fit
// 1024 x IPT items, the sample's column is read once, sorted with a block radix sort (rocprim block
// primitive, keys in registers, exchange through LDS), the tie bands are searched in the LDS copy of the
// sorted keys, and pos / lo / hi leave in gene order as whole coalesced rows of 16-bit numbers.  HBM
// traffic: the matrix once in, three u16 rows once out, instead of keys out, two sort passes in and out,
// keys in again.
// Key = the varying bits (begin_bit .. begin_bit + nbits - 1) of the order-preserving code; bit `nbits`
// marks padding items, which sort behind every gene.
// flags: 0 NaN in the input, 1 some tie, 4 some sample needs more than 31 key bits (the caller then
// redoes the whole transform with the segmented sort).
// the histogram form of the per-sample ranking (t_sample): integer input only, at most this many varying key bits
template <class T> constexpr bool kCountingPath = false;
template <> constexpr bool kCountingPath<int64_t> = true;
constexpr unsigned kCountBits = 15;
constexpr int kCountPer = (1 << kCountBits) / 1024;  // bins per thread in the prefix sums
constexpr size_t kCountWords = (size_t(1) << kCountBits) + (size_t(1) << (kCountBits - 5));  // the histogram, skewed by one word in 32
// the compressed form of the histogram ranking for 16 .. 24 varying key bits (t_sample)
constexpr unsigned kWideBits = 24, kWideExact = 13;
constexpr int kWideBins = 1 << 14, kWideScan = 128;   // codes; largest lossy bucket that is still scanned
constexpr size_t kWideWords = static_cast<size_t>(kWideBins) + (kWideBins >> 5);

// diagnostic builds (-DREO_STAMPS): time marks of workgroup 0 behind the flags (REO_DEBUG_STAMPS=1 prints them)
#ifdef REO_STAMPS
#define TSTAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<unsigned long long *>(flags + 8)[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TSTAMP(k) do { } while (0)
#endif
template <class T, int IPT>
__global__ __launch_bounds__(1024) void t_sample(const T *__restrict__ X, int64_t ld, const int32_t *__restrict__ colmap,
                                                 const int32_t *__restrict__ slots, int G, int Gp, int S,
                                                 uint16_t *__restrict__ pos, uint16_t *__restrict__ lo,
                                                 uint16_t *__restrict__ hi, int32_t *__restrict__ flags,
                                                 uint16_t *__restrict__ skey_rows)   // [S][Gp] scratch, IPT > 32 only (else null)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x;
    const int c = blockIdx.x;  // one sample per workgroup; its three output rows are whole lines of its own
    if (c >= S) {  // one workgroup per padding slot too (slots[S ..]): rows of zeros -- lo = hi = 0 is below no position
        const size_t o = static_cast<size_t>(slots[c]) * Gp;
        for (int q = t; q < Gp / 8; q += 1024) {
            reinterpret_cast<uint4 *>(pos + o)[q] = uint4{0, 0, 0, 0};
            reinterpret_cast<uint4 *>(lo + o)[q] = uint4{0, 0, 0, 0};
            reinterpret_cast<uint4 *>(hi + o)[q] = uint4{0, 0, 0, 0};
        }
        return;
    }
    const T *col = X + static_cast<int64_t>(colmap[c]) * ld;
    const int slot = slots[c];
    int32_t *anytie = flags + 1;
    TSTAMP(0);
    // the sample's own varying key bits (against its first gene): only those are sorted
    const uint64_t key0 = Codec<T>::enc(col[0]);
    // The low 32 bits of every code stay in registers (one word per item, reused below for key | arrival): the varying bits
    // of count data and ranks lie there, and the column is then read once.  (64-bit codes in registers -- 40 to 64 of them
    // at a 128-VGPR budget -- were spilled; the form without them read the column a second time: 20 of 60 us per sample at
    // 30 000 genes.)  If the varying bits reach beyond bit 31 the keys come from a second read.
    uint32_t kw[IPT];
    uint64_t diff = 0;
    bool bad = false;
#pragma unroll
    for (int e = 0; e < IPT; ++e) {
        const int i = e * 1024 + t;  // coalesced; which item holds which gene does not matter to the sort
        kw[e] = static_cast<uint32_t>(key0);
        if (i < G) {
            const T x = col[i];
            bad |= !Codec<T>::ordered(x);
            const uint64_t code = Codec<T>::enc(x);
            kw[e] = static_cast<uint32_t>(code);
            diff |= code ^ key0;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) diff |= __shfl_xor(diff, o, 64);
    unsigned long long *red = reinterpret_cast<unsigned long long *>(smem);
    if ((t & 63) == 0) red[t >> 6] = diff;
    if (__ballot(bad) != 0 && (t & 63) == 0) atomicOr(flags, 1);
    __syncthreads();
    diff = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) diff |= red[w];
    __syncthreads();  // red is overwritten by the histogram
    const unsigned begin_bit = diff ? static_cast<unsigned>(__builtin_ctzll(diff)) : 0u;
    const unsigned nbits = diff ? 64u - static_cast<unsigned>(__builtin_clzll(diff)) - begin_bit : 1u;
    if (nbits > 31) {  // workgroup-uniform: t_sample_wide ranks keys of any width
        if (t == 0) atomicOr(flags + 5, 1);
        return;
    }
    const uint32_t mask = (1u << nbits) - 1u;
    const bool in_regs = begin_bit + nbits <= 32u;   // (workgroup-uniform) the kept words hold every varying bit
    auto key_of = [&](uint32_t kept, int i) -> uint32_t {
        return in_regs ? (kept >> begin_bit) & mask : static_cast<uint32_t>(Codec<T>::enc(col[i]) >> begin_bit) & mask;
    };
    if constexpr (IPT > 32) {
        // 32 769 .. 65 535 genes: the histogram form with 16-BIT bins, two to a word (a count, and a prefix sum, is at most
        // G <= 65 535), for keys of up to 16 varying bits -- ranks of up to 65 535 genes are such keys.  The same LDS as the
        // 32-bit histogram of 15-bit keys.  17 to 24 bits: the compressed histogram below.
        if (kCountingPath<T> && nbits <= kCountBits + 1) {
            uint32_t *hist = reinterpret_cast<uint32_t *>(smem);
            auto at = [](uint32_t w) { return w + (w >> 5); };
            const int nword = max(1 << nbits, 2048) / 2;   // (at least one word per thread in the prefix sums)
            TSTAMP(1);
            for (int w = t; w < nword; w += 1024) hist[at(w)] = 0;
            __syncthreads();
            TSTAMP(2);
#pragma unroll
            for (int e = 0; e < IPT; ++e) {   // kw[e] becomes key | arrival << 16
                const int i = e * 1024 + t;
                if (i < G) {
                    const uint32_t key = key_of(kw[e], i), sh = (key & 1u) * 16u;
                    const uint32_t old = atomicAdd(&hist[at(key >> 1)], 1u << sh);   // (no carry into the upper bin: a count stays below 65 536)
                    kw[e] = key | (((old >> sh) & 0xFFFFu) << 16);
                }
            }
            __syncthreads();
            TSTAMP(3);
            // exclusive prefix sums of the bins, in place: thread t owns words [t wper, t wper + wper)
            const int wper = nword / 1024;
            uint32_t tot = 0;
            for (int u = 0; u < wper; ++u) { const uint32_t w = hist[at(t * wper + u)]; tot += (w & 0xFFFFu) + (w >> 16); }
            uint32_t inc = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t up = __shfl_up(inc, o, 64); if ((t & 63) >= o) inc += up; }
            uint32_t *wtot = hist + kCountWords;  // 16 words behind the histogram (part of the dynamic allocation)
            if ((t & 63) == 63) wtot[t >> 6] = inc;
            __syncthreads();
            uint32_t run = inc - tot;
            for (int w = 0; w < (t >> 6); ++w) run += wtot[w];
            for (int u = 0; u < wper; ++u) {
                const uint32_t w = hist[at(t * wper + u)], a = run, b = run + (w & 0xFFFFu);
                hist[at(t * wper + u)] = a | (b << 16);   // (a prefix sum is at most G: 16 bits)
                run = b + (w >> 16);
            }
            __syncthreads();
            TSTAMP(4);
            auto below = [&](uint32_t key) -> uint32_t {   // genes with a smaller key
                if (static_cast<int>(key) >= 2 * nword) return static_cast<uint32_t>(G);
                return (hist[at(key >> 1)] >> ((key & 1u) * 16u)) & 0xFFFFu;
            };
            bool tied = false;
            uint16_t *prow = pos + static_cast<size_t>(slot) * Gp, *lrow = lo + static_cast<size_t>(slot) * Gp, *hrow = hi + static_cast<size_t>(slot) * Gp;
#pragma unroll
            for (int e = 0; e < IPT; ++e) {
                const int i = e * 1024 + t;
                if (i >= G) continue;
                const uint32_t key = kw[e] & 0xFFFFu;
                // (the bins behind the last key hold G, except when no bin is behind it)
                const uint32_t l = below(key), h = below(key + 1);
                tied |= h - l > 1u;
                prow[i] = static_cast<uint16_t>(l + (kw[e] >> 16));
                lrow[i] = static_cast<uint16_t>(l);
                hrow[i] = static_cast<uint16_t>(h);
            }
            for (int g = G + t; g < Gp; g += 1024) { prow[g] = 0; lrow[g] = 0; hrow[g] = 0; }  // padded genes are below no band edge
            if (tied && *anytie == 0) atomicOr(anytie, 1);
            TSTAMP(5);
            return;
        }
    }
    if (IPT <= 32 && kCountingPath<T> && nbits <= kCountBits) {
        // Integer data whose varying key bits number at most 15 (ranks, small counts): ties are equalities, so a band is
        // a key value, and positions follow from a histogram -- first position of the band = number of smaller keys,
        // position inside the band = order of arrival at the histogram (any order inside a band gives the same counts:
        // a tied partner only has to lie inside the band).  No sort: about a tenth of the radix path's time.
        // (bin b lives at word b + b / 32: a thread that walks its 32 consecutive bins in the prefix sums then meets a
        //  different bank than its neighbours -- unskewed, all 64 lanes of a wave hit one bank)
        uint32_t *hist = reinterpret_cast<uint32_t *>(smem);
        auto at = [](uint32_t b) { return b + (b >> 5); };
        const int nbin = 1 << nbits;
        TSTAMP(1);
        for (int b = t; b < nbin; b += 1024) hist[at(b)] = 0;
        __syncthreads();
        TSTAMP(2);
#pragma unroll
        for (int e = 0; e < IPT; ++e) {   // kw[e] becomes key | arrival << 16 (a key has at most 15 bits, an arrival is below 32 768)
            const int i = e * 1024 + t;
            if (i < G) {
                const uint32_t key = key_of(kw[e], i);
                kw[e] = key | (atomicAdd(&hist[at(key)], 1u) << 16);
            }
        }
        __syncthreads();
        TSTAMP(3);
        // exclusive prefix sums of the bins, in place: thread t owns bins [t per, t per + per)
        const int per = nbin >= 1024 ? nbin / 1024 : 1;
        uint32_t cnt[kCountPer], tot = 0;
#pragma unroll
        for (int u = 0; u < kCountPer; ++u) {
            cnt[u] = (u < per && t * per + u < nbin) ? hist[at(t * per + u)] : 0u;
            tot += cnt[u];
        }
        uint32_t inc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t up = __shfl_up(inc, o, 64); if ((t & 63) >= o) inc += up; }
        uint32_t *wtot = hist + kCountWords;  // 16 words behind the histogram (part of the dynamic allocation)
        if ((t & 63) == 63) wtot[t >> 6] = inc;
        __syncthreads();
        uint32_t run = inc - tot;
        for (int w = 0; w < (t >> 6); ++w) run += wtot[w];
#pragma unroll
        for (int u = 0; u < kCountPer; ++u)
            if (u < per && t * per + u < nbin) { hist[at(t * per + u)] = run; run += cnt[u]; }
        __syncthreads();
        TSTAMP(4);
        bool tied = false;
        uint16_t *prow = pos + static_cast<size_t>(slot) * Gp, *lrow = lo + static_cast<size_t>(slot) * Gp, *hrow = hi + static_cast<size_t>(slot) * Gp;
#pragma unroll
        for (int e = 0; e < IPT; ++e) {
            const int i = e * 1024 + t;
            if (i >= G) continue;
            const uint32_t key = kw[e] & 0xFFFFu;
            const uint32_t l = hist[at(key)], h = static_cast<int>(key) + 1 < nbin ? hist[at(key + 1)] : static_cast<uint32_t>(G);
            tied |= h - l > 1u;
            prow[i] = static_cast<uint16_t>(l + (kw[e] >> 16));
            lrow[i] = static_cast<uint16_t>(l);
            hrow[i] = static_cast<uint16_t>(h);
        }
        for (int g = G + t; g < Gp; g += 1024) { prow[g] = 0; lrow[g] = 0; hrow[g] = 0; }  // padded genes are below no band edge
        if (tied && *anytie == 0) atomicOr(anytie, 1);
        TSTAMP(5);
        return;
    }
    if (kCountingPath<T> && nbits <= kWideBits) {
        // Integer data with 16 to 24 varying key bits (counts with a long tail): a histogram of a MONOTONE COMPRESSION of
        // the key -- exact below 2^13, above that octave and the top mb mantissa bits (kWideBins codes in all) -- puts every
        // gene into a bucket that is ordered against every other bucket; genes whose bucket dropped low bits are scattered
        // into LDS by bucket and rank themselves inside it by scanning its (few) members.  Count data is dense at small
        // values, where the code is exact, and sparse in the tail, where a bucket holds a handful of genes; a sample with a
        // crowded lossy bucket (more than kWideScan members) takes the radix sort below instead.  No sort, no payload.
        uint32_t *hist = reinterpret_cast<uint32_t *>(smem);
        auto at = [](uint32_t b) { return b + (b >> 5); };
        uint32_t *wtot = hist + kWideWords;                       // 16 wave totals, then the largest lossy bucket
        // [Gp] low bits of the genes of lossy buckets, by slot: in LDS up to 32 768 genes, above that in a scratch row (L2: a gene
        // of a lossy bucket reads its few neighbours)
        uint16_t *skey = IPT > 32 ? skey_rows + static_cast<size_t>(c) * Gp : reinterpret_cast<uint16_t *>(wtot + 32);
        const unsigned oct = nbits - kWideExact;                  // octaves above the exact range (3 .. 11)
        const unsigned mb = 31u - static_cast<unsigned>(__builtin_clz(kWideBins / 2 / oct));  // mantissa bits kept per octave
        for (int b = t; b < static_cast<int>(kWideWords); b += 1024) hist[b] = 0;
        if (t < 32) wtot[t] = 0;
        __syncthreads();
        uint32_t low[IPT];   // kw[e] becomes code | arrival << 16 (a code has 14 bits)
#pragma unroll
        for (int e = 0; e < IPT; ++e) {
            const int i = e * 1024 + t;
            low[e] = 0;
            if (i < G) {
                const uint32_t k = key_of(kw[e], i);
                uint32_t code = k;
                if (k >= (1u << kWideExact)) {
                    const unsigned ex = 31u - static_cast<unsigned>(__builtin_clz(k)), d = ex - mb;  // ex >= 13 > mb
                    code = (1u << kWideExact) + ((ex - kWideExact) << mb) + ((k >> d) & ((1u << mb) - 1u));
                    low[e] = (k & ((1u << d) - 1u)) | 0x10000u;   // bit 16: the bucket dropped bits
                }
                kw[e] = code | (atomicAdd(&hist[at(code)], 1u) << 16);
            }
        }
        __syncthreads();
        // exclusive prefix sums of the bins, in place: thread t owns bins [16 t, 16 t + 16)
        constexpr int per = kWideBins / 1024;
        uint32_t cnt[per], tot = 0;
#pragma unroll
        for (int u = 0; u < per; ++u) { cnt[u] = hist[at(t * per + u)]; tot += cnt[u]; }
        uint32_t inc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t up = __shfl_up(inc, o, 64); if ((t & 63) >= o) inc += up; }
        if ((t & 63) == 63) wtot[t >> 6] = inc;
        __syncthreads();
        uint32_t run = inc - tot;
        for (int w = 0; w < (t >> 6); ++w) run += wtot[w];
#pragma unroll
        for (int u = 0; u < per; ++u) { hist[at(t * per + u)] = run; run += cnt[u]; }
        __syncthreads();
        uint32_t crowd = 0;
        auto bucket = [&](uint32_t code, uint32_t &s0, uint32_t &s1) {   // (read again where they are needed: two more arrays of IPT words were spilled)
            s0 = hist[at(code)];
            s1 = code + 1 < static_cast<uint32_t>(kWideBins) ? hist[at(code + 1)] : static_cast<uint32_t>(G);
        };
#pragma unroll
        for (int e = 0; e < IPT; ++e) {
            const int i = e * 1024 + t;
            if (i >= G || !(low[e] >> 16)) continue;
            uint32_t s0, s1;
            bucket(kw[e] & 0xFFFFu, s0, s1);
            skey[s0 + (kw[e] >> 16)] = static_cast<uint16_t>(low[e]);
            crowd = max(crowd, s1 - s0);
        }
        if (__ballot(crowd > static_cast<uint32_t>(kWideScan)) != 0 && (t & 63) == 0) atomicOr(&wtot[16], 1u);
        __syncthreads();
        if (wtot[16] == 0) {  // workgroup-uniform
            bool tied = false;
            uint16_t *prow = pos + static_cast<size_t>(slot) * Gp, *lrow = lo + static_cast<size_t>(slot) * Gp, *hrow = hi + static_cast<size_t>(slot) * Gp;
#pragma unroll
            for (int e = 0; e < IPT; ++e) {
                const int i = e * 1024 + t;
                if (i >= G) continue;
                uint32_t s0, s1;
                bucket(kw[e] & 0xFFFFu, s0, s1);
                uint32_t l = s0, h = s1, p = s0 + (kw[e] >> 16);
                if ((low[e] >> 16) && s1 - s0 > 1u) {  // rank inside the bucket: members with smaller / equal low bits
                    const uint32_t mine = low[e] & 0xFFFFu, me = p;
                    uint32_t smaller = 0, equal = 0, before = 0;
                    for (uint32_t q = s0; q < s1; ++q) {
                        const uint32_t o = skey[q];
                        smaller += o < mine; equal += o == mine; before += (o == mine) & (q < me);
                    }
                    l = s0 + smaller; h = l + equal; p = l + before;
                }
                tied |= h - l > 1u;
                prow[i] = static_cast<uint16_t>(p);
                lrow[i] = static_cast<uint16_t>(l);
                hrow[i] = static_cast<uint16_t>(h);
            }
            for (int g = G + t; g < Gp; g += 1024) { prow[g] = 0; lrow[g] = 0; hrow[g] = 0; }  // padded genes are below no band edge
            if (tied && *anytie == 0) atomicOr(anytie, 1);
            return;
        }
    }
    // anything else -- keys wider than 24 bits, a crowded lossy bucket, Float64 (whose band is not an equality) -- is ranked
    // by t_sample_wide below: the caller launches it when this flag comes back
    if (t == 0) atomicOr(flags + 5, 1);
}

// ---------------------------------------------------------------------------------------------------------------
// t_sample_wide: ranking by SAMPLE-SPLITTER BUCKETS for keys of any width and any distribution -- Float64 input with the
// reference's 0.1 band (/root/reference/src/RankCompV3.jl:72), Int64 keys that vary in more than 24 bits, and samples whose
// compressed histogram (t_sample) met a crowded bucket.  One workgroup per sample, no sort of the genes, no library call:
//   1. 1024 of the sample's order-preserving 64-bit codes (every G/1024-th gene; the largest is replaced by the column's
//      maximum) are sorted in LDS (bitonic, one key per thread): the SPLITTERS.  Buckets follow the data's own density --
//      an arithmetic bucket function cannot: log-expression is a spike of zeros a thousand octaves below a few octaves of
//      values.  Splitter s owns SUB + 1 buckets: the codes strictly between splitters s - 1 and s, cut into SUB equal pieces
//      of the code interval (16 below 20 481 genes: one or two genes per bucket), and the codes EQUAL to splitter s (a
//      heavily repeated value is in the sample, so its copies sit in a bucket of their own and are never scanned);
//   2. a histogram of the buckets (binary search among the splitters; the returned count is the gene's slot inside its
//      bucket), exclusive prefix sums;
//   3. every gene leaves 16 bits of its code (its offset inside the bucket's interval, scaled to 16 bits) and its index at
//      its slot (LDS, bucket order);
//   4. every question the transform asks is a RANK QUERY "how many genes have a code below c": the prefix sum of c's
//      bucket + a scan of that bucket's members (16-bit compares; a member that agrees with c in all 16 bits has its whole
//      key fetched from the column, unless the interval is narrower than 2^16 codes).  pos = rank of the gene's own
//      code (equal codes in slot order), lo = rank of the lowest code still tied with it, hi = rank just past the
//      highest.  For Int64 the tie is equality and all three come out of the scan of the gene's own bucket.  For Float64
//      the two band edges are found in CODE space with the reference's own predicate abs(x - y) < 0.1 in fp64:
//      fl(x - y) is monotone in y, so the tied codes are an interval around the gene's code; start at x -+ 0.1, gallop,
//      bisect (two or three evaluations unless x - 0.1 cancels to something tiny).  Exact, like the band search in the
//      sorted keys that it replaces.
// Phase 4 walks the SLOTS, not the genes: the lanes of a wave then hold members of the same or of neighbouring buckets and
// their scans run equally long (results go to a by-slot scratch row and into gene order through LDS at the end).
// Never gives up: a crowded bucket only costs its own members a longer scan.  Arrival slot, bucket and the 16 offset bits are
// parked in the gene's pos / lo / hi rows between the phases (a thread reads back what it wrote itself).
// flags: 0 NaN in the input, 1 some tie.
constexpr int kSplit1 = 1024;                 // splitters (= threads)

// bytes of LDS in front of the two per-gene arrays: splitters + 16 wave maxima, NB bins (skewed) + end word + wave totals; 16-byte aligned
// (h16: the bins are 16-bit numbers, two to a word -- a count and a prefix sum are at most G <= 65 535)
__host__ __device__ constexpr size_t wide_lds_head(size_t nb, bool h16 = false)
{
    const size_t words = h16 ? nb / 2 + 1 : nb + 1;
    return (8 * (kSplit1 + 16) + 4 * (words + (words >> 5) + 1 + 17) + 15) / 16 * 16;
}

template <class T>
__device__ __forceinline__ bool code_tied(uint64_t c, T x) { return Codec<T>::tie(Codec<T>::dec(c), x); }

// smallest code <= cx that is tied with x (DOWN) / largest code >= cx that is (UP): the band of x in code space
template <bool UP>
__device__ __forceinline__ uint64_t band_edge_code(double x, uint64_t cx)
{
    uint64_t g = Codec<double>::enc(UP ? x + 0.1 : x - 0.1);
    if (UP ? g < cx : g > cx) g = cx;
    // invariant of both branches: `in` is tied, `out` is not (or is the end of the code space), in between unknown
    uint64_t in, out;
    if (code_tied<double>(g, x)) {  // walk away from cx while the codes stay tied
        in = g;
        uint64_t step = 1;
        while (true) {
            const bool room = UP ? (in <= ~0ULL - step) : (in >= step);
            if (!room) { out = UP ? ~0ULL : 0ULL; if (code_tied<double>(out, x)) return out; break; }
            const uint64_t n = UP ? in + step : in - step;
            if (!code_tied<double>(n, x)) { out = n; break; }
            in = n; step <<= 1;
        }
    } else {  // walk towards cx (which is tied with itself) until a code is tied
        out = g;
        uint64_t step = 1;
        while (true) {
            const uint64_t dist = UP ? out - cx : cx - out;
            if (step >= dist) { in = cx; break; }
            const uint64_t n = UP ? out - step : out + step;
            if (code_tied<double>(n, x)) { in = n; break; }
            out = n; step <<= 1;
        }
    }
    while ((UP ? out - in : in - out) > 1) {
        const uint64_t mid = UP ? in + (out - in) / 2 : out + (in - out) / 2;
        if (code_tied<double>(mid, x)) in = mid; else out = mid;
    }
    return in;
}

// GENL: the by-slot gene row lives in LDS beside the offset row (4 bytes per gene: up to 32 768 genes); else in a scratch row (L2) --
// 2 bytes per gene in LDS: up to 65 535 genes, and finer buckets below that.
template <class T, int LOGSUB, bool GENL>
__global__ __launch_bounds__(1024) void t_sample_wide(const T *__restrict__ X, int64_t ld, const int32_t *__restrict__ colmap,
                                                      const int32_t *__restrict__ slots, int G, int Gp, int S,
                                                      uint16_t *__restrict__ pos, uint16_t *__restrict__ lo,
                                                      uint16_t *__restrict__ hi, int32_t *__restrict__ flags,
                                                      uint64_t *__restrict__ oslot, uint16_t *__restrict__ bslot,
                                                      uint32_t *__restrict__ bgslot)   // !GENL: [S][Gp] by slot: bucket | exact << 15 | gene << 16
{
    constexpr int SUB = 1 << LOGSUB, PER = SUB + 1, NB = kSplit1 * PER;   // per splitter: SUB pieces of the interval below it + its equality bucket
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long *spl = reinterpret_cast<unsigned long long *>(smem);   // [1024] sorted splitters; [1024..1039] wave maxima
    uint32_t *hist = reinterpret_cast<uint32_t *>(spl + kSplit1 + 16);         // [NB] bins, skewed by one word in 32; then one end word
    constexpr bool H16 = !GENL;                                               // 16-bit bins: twice as many buckets in the same LDS
    constexpr int HW = H16 ? NB / 2 + 1 : NB + 1;                             // words of the histogram (unskewed), incl. the end bin
    uint32_t *wtot = hist + HW + (HW >> 5) + 1;                               // 16 wave totals
    uint16_t *rem = reinterpret_cast<uint16_t *>(smem + wide_lds_head(NB, H16));   // [Gp] by slot: 16 bits of the code's offset inside its bucket
    uint16_t *gen = rem + Gp;                                                 // GENL: [Gp] by slot: the gene
    uint32_t *bg = GENL ? nullptr : bgslot + static_cast<size_t>(blockIdx.x) * Gp;   // !GENL: bucket word and gene of a slot in one scratch word (ONE scattered store per gene)
    auto gene_at = [&](uint32_t q) -> uint32_t { if constexpr (GENL) return gen[q]; else return bg[q] >> 16; };
    auto at = [](uint32_t b) { return b + (b >> 5); };   // word index, skewed by one word in 32
    uint16_t *h16 = reinterpret_cast<uint16_t *>(hist);
    auto at16 = [](uint32_t b) { return ((b >> 1) + (b >> 6)) * 2 + (b & 1u); };   // 16-bit bin b: half (b & 1) of skewed word b >> 1
    // bin b after the prefix sums: the number of genes in the buckets before b
    auto HB = [&](uint32_t b) -> uint32_t { if constexpr (H16) return h16[at16(b)]; else return hist[at(b)]; };
    const int t = threadIdx.x;
    const int c = blockIdx.x;
    if (c >= S) {  // one workgroup per padding slot too: rows of zeros -- lo = hi = 0 is below no position
        const size_t o = static_cast<size_t>(slots[c]) * Gp;
        for (int q = t; q < Gp / 8; q += 1024) {
            reinterpret_cast<uint4 *>(pos + o)[q] = uint4{0, 0, 0, 0};
            reinterpret_cast<uint4 *>(lo + o)[q] = uint4{0, 0, 0, 0};
            reinterpret_cast<uint4 *>(hi + o)[q] = uint4{0, 0, 0, 0};
        }
        return;
    }
    const T *col = X + static_cast<int64_t>(colmap[c]) * ld;
    const size_t orow = static_cast<size_t>(slots[c]) * Gp;
    uint16_t *prow = pos + orow, *lrow = lo + orow, *hrow = hi + orow;
    int32_t *anytie = flags + 1;
    // ---- 1. the column's maximum, NaNs; the sample, sorted
    TSTAMP(0);
    uint64_t kmax = 0;
    bool bad = false;
#pragma unroll 8
    for (int i = t; i < G; i += 1024) {   // (eight loads in flight: the column comes from HBM here)
        const T x = col[i];
        bad |= !Codec<T>::ordered(x);
        const uint64_t k = Codec<T>::enc(x, static_cast<uint32_t>(i));
        kmax = k > kmax ? k : kmax;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint64_t b = __shfl_xor(kmax, o, 64); kmax = b > kmax ? b : kmax; }
    if ((t & 63) == 0) spl[kSplit1 + (t >> 6)] = kmax;
    if (__ballot(bad) != 0 && (t & 63) == 0) atomicOr(flags, 1);
    for (int b = t; b < HW + (HW >> 5) + 1; b += 1024) hist[b] = 0;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 16; ++w) { const uint64_t b = spl[kSplit1 + w]; kmax = b > kmax ? b : kmax; }
    TSTAMP(1);
    {
        // thread t's sample: gene floor(t G / 1024) (repeats when G < 1024: equal splitters leave empty buckets between them)
        const int gs = static_cast<int>((static_cast<int64_t>(t) * G) >> 10);
        uint64_t v = t == kSplit1 - 1 ? kmax : Codec<T>::enc(col[gs], static_cast<uint32_t>(gs));
        // bitonic sort, one key per thread: distances below 64 inside the wave, the others through LDS
        for (int k = 2; k <= kSplit1; k <<= 1) {
            const bool up = (t & k) == 0;
            for (int j = k >> 1; j > 0; j >>= 1) {
                uint64_t pv;
                if (j >= 64) {
                    __syncthreads();
                    spl[t] = v;
                    __syncthreads();
                    pv = spl[t ^ j];
                } else {
                    pv = __shfl_xor(v, j, 64);
                }
                const bool keep_min = ((t & j) == 0) == up;
                const bool take = keep_min ? pv < v : pv > v;
                v = take ? pv : v;
            }
        }
        __syncthreads();
        spl[t] = v;
    }
    __syncthreads();
    // A code k <= kmax: lb = number of splitters below it (10 steps).  It equals splitter lb (bucket lb PER + SUB: one
    // value only) or lies strictly between splitters lb - 1 and lb, in piece ((k - from) >> sh) of that interval; r16 = the
    // next 16 bits of its offset; exact: bucket and r16 are the whole code.
    auto locate = [&](uint64_t k, uint32_t &bucket, uint32_t &r16, bool &exact) {
        int lb = 0;   // (a 4-ary search -- five levels of three independent reads -- was slower: 48 against 29 us for the histogram phase)
#pragma unroll
        for (int s = kSplit1 / 2; s > 0; s >>= 1) lb += spl[lb + s - 1] < k ? s : 0;
        const uint64_t to = spl[lb];
        if (to == k) { bucket = lb * PER + SUB; r16 = 0; exact = true; return; }
        const uint64_t from = lb ? spl[lb - 1] : 0ULL, width = to - from, o = k - from;   // 0 < o < width
        const int bits = 64 - __builtin_clzll(width);
        const int sh = bits > LOGSUB ? bits - LOGSUB : 0;
        bucket = lb * PER + static_cast<uint32_t>(o >> sh);
        const int rs = sh > 16 ? sh - 16 : 0;
        r16 = static_cast<uint32_t>((o & ((1ULL << sh) - 1ULL)) >> rs);
        exact = sh <= 16;
    };
    TSTAMP(2);
    // ---- 2. histogram; arrival slot (+ `exact` in bit 15), bucket and r16 parked in the pos / lo / hi rows
#pragma unroll 2
    for (int i = t; i < G; i += 1024) {
        const uint64_t k = Codec<T>::enc(col[i], static_cast<uint32_t>(i));
        uint32_t b, r; bool ex;
        locate(k, b, r, ex);
        uint32_t arrival;   // (inside the bucket: below 65 536)
        if constexpr (H16) { const uint32_t sh = (b & 1u) * 16u; arrival = (atomicAdd(&hist[at(b >> 1)], 1u << sh) >> sh) & 0xFFFFu; }   // (no carry into the upper bin: a count stays below 65 536)
        else arrival = atomicAdd(&hist[at(b)], 1u);
        prow[i] = static_cast<uint16_t>(arrival);
        lrow[i] = static_cast<uint16_t>(b | (ex ? 0x8000u : 0u));      // (a bucket index is below 2^15: 1 024 x 17)
        hrow[i] = static_cast<uint16_t>(r);
    }
    __syncthreads();
    TSTAMP(3);
    {   // exclusive prefix sums of the bins, in place: thread t owns the PER bins of splitter t
        uint32_t cnt[PER], tot = 0;
#pragma unroll
        for (int u = 0; u < PER; ++u) { cnt[u] = HB(t * PER + u); tot += cnt[u]; }   // (16-bit bins: two threads share a word at their border -- all reads are in front of the barrier, all writes behind it)
        uint32_t inc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t up = __shfl_up(inc, o, 64); if ((t & 63) >= o) inc += up; }
        if ((t & 63) == 63) wtot[t >> 6] = inc;
        __syncthreads();
        uint32_t run = inc - tot;
        for (int w = 0; w < (t >> 6); ++w) run += wtot[w];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            if constexpr (H16) h16[at16(t * PER + u)] = static_cast<uint16_t>(run); else hist[at(t * PER + u)] = run;
            run += cnt[u];
        }
        if (t == kSplit1 - 1) { if constexpr (H16) h16[at16(NB)] = static_cast<uint16_t>(G); else hist[at(NB)] = static_cast<uint32_t>(G); }
    }
    __syncthreads();
    TSTAMP(4);
    // ---- 3. members into bucket order
    uint16_t *bs = bslot + static_cast<size_t>(c) * Gp;      // by slot: bucket | exact << 15 (global scratch: phase 4 walks the slots)
    uint64_t *os = oslot + static_cast<size_t>(c) * Gp;      // by slot: gene | pos << 16 | lo << 32 | hi << 48
#pragma unroll 4
    for (int i = t; i < G; i += 1024) {
        const uint32_t bw = lrow[i], sl = HB(bw & 0x7FFFu) + prow[i];
        rem[sl] = hrow[i];
        if constexpr (GENL) { gen[sl] = static_cast<uint16_t>(i); bs[sl] = static_cast<uint16_t>(bw); }
        else bg[sl] = bw | (static_cast<uint32_t>(i) << 16);
    }
    __syncthreads();
    // number of genes whose code is below cq (le: below or equal)
    auto rank_of = [&](uint64_t cq, bool le) -> uint32_t {
        if (cq > kmax) return static_cast<uint32_t>(G);
        uint32_t b, cr; bool exact;
        locate(cq, b, cr, exact);
        const uint32_t s0 = HB(b), s1 = HB(b + 1);
        if (b % PER == SUB) return le ? s1 : s0;   // an equality bucket: every member is cq
        uint32_t n = 0;
        auto same = [&](uint32_t q) -> uint32_t {   // a member with cq's 16 offset bits: equal when the bucket is narrower than 2^16 codes, else the whole key decides
            if (exact) return le ? 1u : 0u;
            const uint32_t gq = gene_at(q);
            const uint64_t k = Codec<T>::enc(col[gq], gq);
            return (k < cq || (le && k == cq)) ? 1u : 0u;
        };
        uint32_t q = s0;
        for (; q < s1 && (q & 3u); ++q) { const uint32_t r = rem[q]; n += r < cr ? 1u : (r == cr ? same(q) : 0u); }
        for (; q + 4 <= s1; q += 4) {   // four offsets per LDS access (the row is 8-byte aligned at multiples of four)
            const uint2 w = *reinterpret_cast<const uint2 *>(rem + q);
            const uint32_t r0 = w.x & 0xFFFFu, r1 = w.x >> 16, r2 = w.y & 0xFFFFu, r3 = w.y >> 16;
            n += (r0 < cr ? 1u : 0u) + (r1 < cr ? 1u : 0u) + (r2 < cr ? 1u : 0u) + (r3 < cr ? 1u : 0u);
            if (r0 == cr) n += same(q);
            if (r1 == cr) n += same(q + 1);
            if (r2 == cr) n += same(q + 2);
            if (r3 == cr) n += same(q + 3);
        }
        for (; q < s1; ++q) { const uint32_t r = rem[q]; n += r < cr ? 1u : (r == cr ? same(q) : 0u); }
        return s0 + n;
    };
    // ---- 4. pos, lo, hi, slot by slot: the lanes of a wave take CONSECUTIVE slots, i.e. members of the same or of neighbouring
    // buckets, so their scans run equally long (in gene order a wave ran as long as the longest of 64 unrelated buckets: mean
    // bucket 4.6 members, wave maximum 28) and their band edges land in neighbouring buckets.  The gene's value is gathered
    // from the column (L2); the results go to a by-slot scratch row and are put in gene order through LDS afterwards.
    TSTAMP(5);
    bool tied = false;
#pragma unroll 1
    for (int sl = t; sl < G; sl += 1024) {   // (fetching the next slot's value one iteration ahead changed nothing: 102 against 103 us)
        uint32_t gene, bw;
        if constexpr (GENL) { gene = gen[sl]; bw = bs[sl]; } else { const uint32_t w = bg[sl]; gene = w >> 16; bw = w & 0xFFFFu; }
        const uint32_t b = bw & 0x7FFFu, mr = rem[sl], me = static_cast<uint32_t>(sl);
        const bool exact = (bw & 0x8000u) != 0;
        const T x = col[gene];
        const uint64_t k = Codec<T>::enc(x, gene);
        const uint32_t s0 = HB(b), s1 = HB(b + 1);
        uint32_t l, h, p;
        if (b % PER == SUB) {  // one value: slot order
            l = s0; h = s1; p = me;
        } else {
            // members with smaller offset bits sort before this gene; members with the SAME 16 bits are equal when the bucket is
            // narrower than 2^16 codes, else their whole keys decide: their slots are collected first and the keys fetched
            // together (one round trip to the column in L2, not one per member -- repeated values are common in expression data)
            uint32_t smaller = 0, equal = 1, before = 0, amb[4], namb = 0;
            for (uint32_t q = s0; q < s1; ++q) {
                const uint32_t r = rem[q];
                smaller += r < mr ? 1u : 0u;
                if (r == mr && q != me) {
                    if (exact) { ++equal; before += q < me ? 1u : 0u; }
                    else if (namb < 4) amb[namb++] = q;
                    else { const uint32_t gq = gene_at(q); const uint64_t kq = Codec<T>::enc(col[gq], gq); smaller += kq < k ? 1u : 0u; equal += kq == k ? 1u : 0u; before += (kq == k && q < me) ? 1u : 0u; }
                }
            }
            if (namb) {
                uint64_t kq[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const uint32_t gq = u < static_cast<int>(namb) ? gene_at(amb[u]) : 0u; kq[u] = u < static_cast<int>(namb) ? Codec<T>::enc(col[gq], gq) : 0ULL; }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (u < static_cast<int>(namb)) { smaller += kq[u] < k ? 1u : 0u; equal += kq[u] == k ? 1u : 0u; before += (kq[u] == k && amb[u] < me) ? 1u : 0u; }
            }
            l = s0 + smaller; h = l + equal; p = l + before;
        }
        if constexpr (std::is_same<T, double>::value) {  // the band is wider than the equal values: two more rank queries
            if (!isinf(x)) {   // (an infinity's code is its own: l = p, h = p + 1 from the scan above -- tied with nothing, :72)
                l = rank_of(band_edge_code<false>(x, k), false);
                h = rank_of(band_edge_code<true>(x, k), true);
            }
        }
        tied |= h - l > 1u;
        os[sl] = static_cast<uint64_t>(gene) | (static_cast<uint64_t>(p) << 16) | (static_cast<uint64_t>(l) << 32) | (static_cast<uint64_t>(h) << 48);
    }
    TSTAMP(7);
    __syncthreads();   // every query is done: the LDS behind the splitters becomes pos16 / lo16 / hi16, indexed by gene
    {   // a row at a time through 2 Gp bytes behind the splitters (with the gene row in LDS two rows fit: pos and lo go together)
        uint16_t *a16 = reinterpret_cast<uint16_t *>(hist), *b16 = a16 + Gp;
        for (int g = G + t; g < Gp; g += 1024) { a16[g] = 0; if (GENL) b16[g] = 0; }  // padded genes are below no band edge
        for (int sl = t; sl < G; sl += 1024) {
            const uint64_t o = os[sl];
            const uint32_t gene = static_cast<uint32_t>(o & 0xFFFFu);
            a16[gene] = static_cast<uint16_t>(o >> 16);
            if (GENL) b16[gene] = static_cast<uint16_t>(o >> 32);
        }
        __syncthreads();
        for (int q = t; q < Gp / 8; q += 1024) {
            reinterpret_cast<uint4 *>(prow)[q] = reinterpret_cast<const uint4 *>(a16)[q];
            if (GENL) reinterpret_cast<uint4 *>(lrow)[q] = reinterpret_cast<const uint4 *>(b16)[q];
        }
        __syncthreads();
        if (!GENL) {
            for (int sl = t; sl < G; sl += 1024) { const uint64_t o = os[sl]; a16[o & 0xFFFFu] = static_cast<uint16_t>(o >> 32); }   // (the padded genes' zeros are still there)
            __syncthreads();
            for (int q = t; q < Gp / 8; q += 1024) reinterpret_cast<uint4 *>(lrow)[q] = reinterpret_cast<const uint4 *>(a16)[q];
            __syncthreads();
        }
        for (int sl = t; sl < G; sl += 1024) { const uint64_t o = os[sl]; a16[o & 0xFFFFu] = static_cast<uint16_t>(o >> 48); }
        __syncthreads();
        for (int q = t; q < Gp / 8; q += 1024) reinterpret_cast<uint4 *>(hrow)[q] = reinterpret_cast<const uint4 *>(a16)[q];
    }
    if (tied && *anytie == 0) atomicOr(anytie, 1);
    TSTAMP(6);
}

// ---------------------------------------------------------------------------------------------------------------
// t_sample_big (round 5): the bucket ranking of t_sample_wide for MORE THAN 65 535 GENES (up to kMaxGenes = 262 143), every input type.
// A sample's per-gene rows no longer fit the LDS (2 bytes per gene would be 512 KB), so only the splitters and the 32-bit bins stay
// there (80 KB) and the by-slot row moves to a scratch row in L2 -- ONE 64-bit record per slot: gene | bucket word << 32 | the 16 offset
// bits << 48, so a scan reads one word per member.  Positions are 18-bit numbers: 32-bit output rows (t_slice_big makes the planes).
// Same algorithm otherwise: 1 024 sample splitters, 16 sub-buckets + an equality bucket each, rank queries = prefix sum + scan of
// one bucket, Float64 band edges in code space with the reference's own predicate (src/RankCompV3.jl:72).  This replaces
// rocprim::segmented_radix_sort_pairs, the last library call on the product path.
template <class T>
__global__ __launch_bounds__(1024) void t_sample_big(const T *__restrict__ X, int64_t ld, const int32_t *__restrict__ colmap,
                                                     const int32_t *__restrict__ slots, int G, int Gp,
                                                     uint32_t *__restrict__ pos, uint32_t *__restrict__ lo, uint32_t *__restrict__ hi,
                                                     int32_t *__restrict__ flags, uint64_t *__restrict__ recs)   // [samples of the launch][Gp]
{
    constexpr int LOGSUB = 4, SUB = 1 << LOGSUB, PER = SUB + 1, NB = kSplit1 * PER;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long *spl = reinterpret_cast<unsigned long long *>(smem);   // [1024] sorted splitters; [1024..1039] wave maxima
    uint32_t *hist = reinterpret_cast<uint32_t *>(spl + kSplit1 + 16);         // [NB + 1] bins, skewed by one word in 32
    constexpr int HW = NB + 1;
    uint32_t *wtot = hist + HW + (HW >> 5) + 1;                               // 16 wave totals
    auto at = [](uint32_t b) { return b + (b >> 5); };
    auto HB = [&](uint32_t b) -> uint32_t { return hist[at(b)]; };
    const int t = threadIdx.x, c = blockIdx.x;
    const T *col = X + static_cast<int64_t>(colmap[c]) * ld;
    const size_t orow = static_cast<size_t>(slots[c]) * Gp;
    uint32_t *prow = pos + orow, *lrow = lo + orow, *hrow = hi + orow;
    uint64_t *rec = recs + static_cast<size_t>(c) * Gp;
    int32_t *anytie = flags + 1;
    // ---- 1. the column's maximum, NaNs; the sample, sorted
    uint64_t kmax = 0;
    bool bad = false;
#pragma unroll 8
    for (int i = t; i < G; i += 1024) {
        const T x = col[i];
        bad |= !Codec<T>::ordered(x);
        const uint64_t k = Codec<T>::enc(x, static_cast<uint32_t>(i));
        kmax = k > kmax ? k : kmax;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint64_t b = __shfl_xor(kmax, o, 64); kmax = b > kmax ? b : kmax; }
    if ((t & 63) == 0) spl[kSplit1 + (t >> 6)] = kmax;
    if (__ballot(bad) != 0 && (t & 63) == 0) atomicOr(flags, 1);
    for (int b = t; b < HW + (HW >> 5) + 1; b += 1024) hist[b] = 0;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 16; ++w) { const uint64_t b = spl[kSplit1 + w]; kmax = b > kmax ? b : kmax; }
    {
        const int gs = static_cast<int>((static_cast<int64_t>(t) * G) >> 10);
        uint64_t v = t == kSplit1 - 1 ? kmax : Codec<T>::enc(col[gs], static_cast<uint32_t>(gs));
        for (int k = 2; k <= kSplit1; k <<= 1) {   // bitonic sort, one key per thread
            const bool up = (t & k) == 0;
            for (int j = k >> 1; j > 0; j >>= 1) {
                uint64_t pv;
                if (j >= 64) {
                    __syncthreads();
                    spl[t] = v;
                    __syncthreads();
                    pv = spl[t ^ j];
                } else {
                    pv = __shfl_xor(v, j, 64);
                }
                const bool keep_min = ((t & j) == 0) == up;
                const bool take = keep_min ? pv < v : pv > v;
                v = take ? pv : v;
            }
        }
        __syncthreads();
        spl[t] = v;
    }
    __syncthreads();
    auto locate = [&](uint64_t k, uint32_t &bucket, uint32_t &r16, bool &exact) {   // as in t_sample_wide
        int lb = 0;
#pragma unroll
        for (int s = kSplit1 / 2; s > 0; s >>= 1) lb += spl[lb + s - 1] < k ? s : 0;
        const uint64_t to = spl[lb];
        if (to == k) { bucket = lb * PER + SUB; r16 = 0; exact = true; return; }
        const uint64_t from = lb ? spl[lb - 1] : 0ULL, width = to - from, o = k - from;
        const int bits = 64 - __builtin_clzll(width);
        const int sh = bits > LOGSUB ? bits - LOGSUB : 0;
        bucket = lb * PER + static_cast<uint32_t>(o >> sh);
        const int rs = sh > 16 ? sh - 16 : 0;
        r16 = static_cast<uint32_t>((o & ((1ULL << sh) - 1ULL)) >> rs);
        exact = sh <= 16;
    };
    // ---- 2. histogram; the arrival slot is parked in the pos row, bucket word and offset bits in the lo row (32-bit rows)
#pragma unroll 2
    for (int i = t; i < G; i += 1024) {
        const uint64_t k = Codec<T>::enc(col[i], static_cast<uint32_t>(i));
        uint32_t b, r; bool ex;
        locate(k, b, r, ex);
        prow[i] = atomicAdd(&hist[at(b)], 1u);
        lrow[i] = b | (ex ? 0x8000u : 0u) | (r << 16);
    }
    __syncthreads();
    {   // exclusive prefix sums of the bins, in place: thread t owns the PER bins of splitter t
        uint32_t cnt[PER], tot = 0;
#pragma unroll
        for (int u = 0; u < PER; ++u) { cnt[u] = HB(t * PER + u); tot += cnt[u]; }
        uint32_t inc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t up = __shfl_up(inc, o, 64); if ((t & 63) >= o) inc += up; }
        if ((t & 63) == 63) wtot[t >> 6] = inc;
        __syncthreads();
        uint32_t run = inc - tot;
        for (int w = 0; w < (t >> 6); ++w) run += wtot[w];
#pragma unroll
        for (int u = 0; u < PER; ++u) { hist[at(t * PER + u)] = run; run += cnt[u]; }
        if (t == kSplit1 - 1) hist[at(NB)] = static_cast<uint32_t>(G);
    }
    __syncthreads();
    // ---- 3. members into bucket order: one record per slot
#pragma unroll 4
    for (int i = t; i < G; i += 1024) {
        const uint32_t w = lrow[i], sl = HB(w & 0x7FFFu) + prow[i];
        rec[sl] = static_cast<uint64_t>(static_cast<uint32_t>(i)) | (static_cast<uint64_t>(w) << 32);   // gene | bucket word << 32 | offset bits << 48
    }
    __syncthreads();   // (workgroup barrier: the records written above are visible to every thread of this workgroup)
    auto rank_of = [&](uint64_t cq, bool le) -> uint32_t {   // number of genes whose code is below cq (le: below or equal)
        if (cq > kmax) return static_cast<uint32_t>(G);
        uint32_t b, cr; bool exact;
        locate(cq, b, cr, exact);
        const uint32_t s0 = HB(b), s1 = HB(b + 1);
        if (b % PER == SUB) return le ? s1 : s0;
        uint32_t n = 0;
        for (uint32_t q0 = s0; q0 < s1; q0 += 4) {   // four records per round trip
            uint64_t w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) w[u] = q0 + u < s1 ? rec[q0 + u] : ~0ULL;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (q0 + u >= s1) break;
                const uint32_t r = static_cast<uint32_t>(w[u] >> 48);
                if (r < cr) ++n;
                else if (r == cr) {
                    if (exact) n += le ? 1u : 0u;
                    else { const uint32_t gq = static_cast<uint32_t>(w[u]); const uint64_t k = Codec<T>::enc(col[gq], gq); n += (k < cq || (le && k == cq)) ? 1u : 0u; }
                }
            }
        }
        return s0 + n;
    };
    // ---- 4. pos, lo, hi, slot by slot (the lanes of a wave take consecutive slots: members of the same or of neighbouring buckets)
    bool tied = false;
#pragma unroll 1
    for (int sl = t; sl < G; sl += 1024) {
        const uint64_t w = rec[sl];
        const uint32_t gene = static_cast<uint32_t>(w), bw = static_cast<uint32_t>(w >> 32) & 0xFFFFu, mr = static_cast<uint32_t>(w >> 48), me = static_cast<uint32_t>(sl);
        const uint32_t b = bw & 0x7FFFu;
        const bool exact = (bw & 0x8000u) != 0;
        // The gene's own value is a scattered 8-byte read of a column that no cache holds at this size (256 workgroups x 0.5-2 MB): only
        // Float64 (band edges) and a bucket wider than 2^16 codes need it -- an Int64 key in an exact bucket IS (bucket, offset bits).
        const bool need_x = std::is_same<T, double>::value || !exact;
        T x = T(0);
        uint64_t k = 0;
        if (need_x) { x = col[gene]; k = Codec<T>::enc(x, gene); }
        const uint32_t s0 = HB(b), s1 = HB(b + 1);
        uint32_t l, h, p;
        if (b % PER == SUB) {  // one value: slot order
            l = s0; h = s1; p = me;
        } else {
            uint32_t smaller = 0, equal = 1, before = 0;
            for (uint32_t q0 = s0; q0 < s1; q0 += 4) {   // four records per round trip
                uint64_t wq[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) wq[u] = q0 + u < s1 ? rec[q0 + u] : ~0ULL;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t q = q0 + u;
                    if (q >= s1) break;
                    const uint32_t r = static_cast<uint32_t>(wq[u] >> 48);
                    smaller += r < mr ? 1u : 0u;
                    if (r == mr && q != me) {
                        if (exact) { ++equal; before += q < me ? 1u : 0u; }
                        else { const uint32_t gq = static_cast<uint32_t>(wq[u]); const uint64_t kq = Codec<T>::enc(col[gq], gq); smaller += kq < k ? 1u : 0u; equal += kq == k ? 1u : 0u; before += (kq == k && q < me) ? 1u : 0u; }
                    }
                }
            }
            l = s0 + smaller; h = l + equal; p = l + before;
        }
        if constexpr (std::is_same<T, double>::value) {  // the band is wider than the equal values: two more rank queries
            if (!isinf(x)) {   // (an infinity's code is its own: l = p, h = p + 1 from the scan above -- tied with nothing, :72)
                l = rank_of(band_edge_code<false>(x, k), false);
                h = rank_of(band_edge_code<true>(x, k), true);
            }
        }
        tied |= h - l > 1u;
        // (the pos / lo rows still hold the parked words of OTHER genes that later slots of this loop do not read any more: phase 3 was
        //  the last reader of a parked word, and it is behind a barrier)
        prow[gene] = p; lrow[gene] = l; hrow[gene] = h;
    }
    if (tied && *anytie == 0) atomicOr(anytie, 1);
}

// 16-bit rows -> bit planes over 32-sample blocks.  One thread per (pair of genes, block): reads the genes' 32 numbers
// of each of the three rows (coalesced over genes) and writes
//   P  [nblk][4][Gp] uint4 : planes 4q..4q+3 of pos of gene g in block b at (b * 4 + q) * Gp + g   (lane operand)
//   AL [nblk][Gp][4] uint4 : the 16 plane words of lo of gene g in block b at (b * Gp + g) * 4 ..   (tile operand)
//   AH likewise for hi.
// In AL / AH plane k sits in word (k + 15) % 16, one word off its place in P: register tuples are even-aligned, so
// the P word and the A word that meet in one v_bitop3_b32 then never share a VGPR bank (measured: a bit op whose
// three sources share a bank issues at half rate).
__global__ __launch_bounds__(256) void t_slice(const uint16_t *__restrict__ pos, const uint16_t *__restrict__ lo,
                                               const uint16_t *__restrict__ hi, int Gp, uint4 *__restrict__ P,
                                               uint4 *__restrict__ AL, uint4 *__restrict__ AH, int b0)
{
    // two adjacent genes per thread: their 32 x (16 + 16) bits of a block are one 32 x 32 bit matrix, transposed in
    // registers by five rounds of masked swaps (480 operations for both genes; picking the bits one by one was 2 x 1024)
    const int g = (blockIdx.x * 256 + threadIdx.x) * 2, b = blockIdx.y + b0;   // (b0: the first block of a group, when groups are sliced as they complete)
    const size_t row0 = (static_cast<size_t>(b) * 32 * Gp + g) / 2;  // in pairs of 16-bit numbers
    uint32_t w[32];
    auto planes = [&](const uint16_t *__restrict__ src) {
        const uint32_t *src2 = reinterpret_cast<const uint32_t *>(src);
#pragma unroll
        for (int s = 0; s < 32; ++s) w[s] = src2[row0 + static_cast<size_t>(s) * (Gp / 2)];  // gene g in the low half, g + 1 in the high half
        uint32_t m = 0x0000FFFFu;
#pragma unroll
        for (int j = 16; j != 0; j >>= 1, m ^= (m << j)) {
#pragma unroll
            for (int k = 0; k < 32; k = (k + j + 1) & ~j) {
                const uint32_t t = ((w[k] >> j) ^ w[k + j]) & m;
                w[k] ^= t << j;
                w[k + j] ^= t;
            }
        }
        // now w[k] = plane k of gene g (bit s = sample s), w[16 + k] = plane k of gene g + 1
    };
    planes(pos);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint4 *o = P + (static_cast<size_t>(b) * 4 + q) * Gp + g;
        o[0] = uint4{w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]};
        o[1] = uint4{w[16 + 4 * q], w[16 + 4 * q + 1], w[16 + 4 * q + 2], w[16 + 4 * q + 3]};
    }
    auto skewed = [&](uint4 *__restrict__ dst) {
        uint4 *o = dst + (static_cast<size_t>(b) * Gp + g) * 4;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t *v = w + 16 * h;
            o[4 * h + 0] = uint4{v[1], v[2], v[3], v[4]};
            o[4 * h + 1] = uint4{v[5], v[6], v[7], v[8]};
            o[4 * h + 2] = uint4{v[9], v[10], v[11], v[12]};
            o[4 * h + 3] = uint4{v[13], v[14], v[15], v[0]};
        }
    };
    planes(lo);
    skewed(AL);
    planes(hi);
    skewed(AH);
}

// More than 65 535 genes: 32-bit rows -> the BIG plane layout (up to 20 planes).  One thread per gene and block: its 32
// numbers of a row are a 32 x 32 bit matrix, transposed in registers; plane k of every number is word k.
//   P  [nblk][5][Gp] uint4 : planes 4q..4q+3 of pos of gene g in block b at (b * 5 + q) * Gp + g
//   AL [nblk][Gp][8] uint4 : the plane words of lo of gene g in block b at (b * Gp + g) * 8 .. (words 0..19 used, plane k in
//                            word k; rows of 128 bytes so that a tile's 32 rows are four 1-KiB LDS-DMA pieces)
//   AH likewise for hi.  (The count loop's generator places the registers itself: no word skew is needed here.)
__global__ __launch_bounds__(256) void t_slice_big(const uint32_t *__restrict__ pos, const uint32_t *__restrict__ lo,
                                                   const uint32_t *__restrict__ hi, int Gp, uint4 *__restrict__ P,
                                                   uint4 *__restrict__ AL, uint4 *__restrict__ AH)
{
    const int g = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (g >= Gp) return;
    uint32_t w[32];
    auto planes = [&](const uint32_t *__restrict__ src) {
#pragma unroll
        for (int s = 0; s < 32; ++s) w[s] = src[(static_cast<size_t>(b) * 32 + s) * Gp + g];
        uint32_t m = 0x0000FFFFu;
#pragma unroll
        for (int j = 16; j != 0; j >>= 1, m ^= (m << j)) {
#pragma unroll
            for (int k = 0; k < 32; k = (k + j + 1) & ~j) {
                const uint32_t t = ((w[k] >> j) ^ w[k + j]) & m;
                w[k] ^= t << j;
                w[k + j] ^= t;
            }
        }
        // now w[k] = plane k of gene g (bit s = sample s)
    };
    planes(pos);
#pragma unroll
    for (int q = 0; q < 5; ++q) P[(static_cast<size_t>(b) * 5 + q) * Gp + g] = uint4{w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]};
    auto rows = [&](uint4 *__restrict__ dst) {
        uint4 *o = dst + (static_cast<size_t>(b) * Gp + g) * 8;
#pragma unroll
        for (int q = 0; q < 5; ++q) o[q] = uint4{w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]};
    };
    planes(lo);
    rows(AL);
    planes(hi);
    rows(AH);
}

// A launch ranks the samples of a LIST: colmap[t] = column of X, slots[t] = its row set in the 16-bit outputs, t < n_samples;
// entries n_samples .. n_slots - 1 of `slots` are padding slots (rows of zeros).  The whole-matrix transform passes the
// sorted sample order; the pipelined upload (eager_upload below) passes the columns of one chunk.
struct SampleList {
    const int32_t *colmap, *slots;
    int n_samples, n_slots;
};

template <class T, int IPT>
int32_t launch_sample(reo_ctx *c, const T *X, const SampleList &sl, int32_t *d_flags)
{
    const size_t lds = IPT > 32 ? sizeof(uint32_t) * kCountWords + 64   // (above 32 768 genes: the 16-bit-bin histogram only)
                                : std::max({sizeof(uint32_t) * kCountWords + 64,
                                            sizeof(uint32_t) * (kWideWords + 32) + sizeof(uint16_t) * static_cast<size_t>(c->Gp)});
    // every time: the attribute belongs to the (function, device) pair and a process may use several devices
    REO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(t_sample<T, IPT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    uint16_t *skey_rows = nullptr;
    if (IPT > 32) {  // (the segmented path's buffer: it is not running)
        int32_t rc;
        if ((rc = c->t_vin.ensure(static_cast<size_t>(c->S) * c->Gp))) return rc;
        skey_rows = c->t_vin.p;
    }
    t_sample<T, IPT><<<static_cast<unsigned>(sl.n_slots), 1024, lds, c->stream>>>(X, c->ld, sl.colmap, sl.slots, static_cast<int>(c->G), c->Gp,  // (a workgroup per slot: samples, then padding)
                                                                            sl.n_samples, c->t_pos16.p, c->t_lo16.p, c->t_hi16.p, d_flags, skey_rows);
    REO_HIP_CHECK(hipGetLastError());
    return REO_OK;
}

template <class T, int LOGSUB, bool GENL>
int32_t launch_sample_wide(reo_ctx *c, const T *X, const SampleList &sl, int32_t *d_flags)
{
    constexpr size_t NB = static_cast<size_t>(kSplit1) * ((1 << LOGSUB) + 1);
    const size_t lds = wide_lds_head(NB, !GENL) + sizeof(uint16_t) * (GENL ? 2 : 1) * static_cast<size_t>(c->Gp);
    if (lds > 160 * 1024) { set_error("t_sample_wide: %zu bytes of LDS for %d genes", lds, c->Gp); return REO_EINVAL; }
    REO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(t_sample_wide<T, LOGSUB, GENL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    // by-slot scratch rows of every sample (the segmented path's buffers: it is not running)
    int32_t rc;
    const size_t n = static_cast<size_t>(c->S) * c->Gp;
    if ((rc = c->t_kin.ensure(n))) return rc;
    if (GENL ? (rc = c->t_vin.ensure(n)) : (rc = c->t_vin32.ensure(n))) return rc;
    t_sample_wide<T, LOGSUB, GENL><<<static_cast<unsigned>(sl.n_slots), 1024, lds, c->stream>>>(X, c->ld, sl.colmap, sl.slots, static_cast<int>(c->G), c->Gp,
                                                                                    sl.n_samples, c->t_pos16.p, c->t_lo16.p, c->t_hi16.p, d_flags, c->t_kin.p, c->t_vin.p,
                                                                                    GENL ? nullptr : c->t_vin32.p);
    REO_HIP_CHECK(hipGetLastError());
    return REO_OK;
}

// the in-LDS ranking of every sample: the histogram forms for Int64 (t_sample), the bucket form for everything else
template <class T>
int32_t launch_lds_ranking(reo_ctx *c, const T *X, const SampleList &d_order, int32_t *d_flags, bool wide)
{
    if (d_order.n_slots <= 0) return REO_OK;
    const int64_t G = c->G;
    if (!wide) {
        if constexpr (kCountingPath<T>) {
            if (G <= 8 * 1024) return launch_sample<T, 8>(c, X, d_order, d_flags);
            if (G <= 20 * 1024) return launch_sample<T, 20>(c, X, d_order, d_flags);
            if (G <= 24 * 1024) return launch_sample<T, 24>(c, X, d_order, d_flags);
            if (G <= 32 * 1024) return launch_sample<T, 32>(c, X, d_order, d_flags);
            return launch_sample<T, 64>(c, X, d_order, d_flags);   // (16-bit bins; the compressed histogram keeps its low-bit rows in L2)
        }
    }
    // sub-buckets per splitter interval: what fits the LDS beside the by-slot rows -- 4 bytes per gene up to 24 576 genes, 2 above
    // (the gene ids then live in L2, packed with the bucket word, and the bins are 16-bit numbers)
    auto fits = [&](int logsub) { return wide_lds_head(static_cast<size_t>(kSplit1) * ((1 << logsub) + 1), true) + sizeof(uint16_t) * static_cast<size_t>(c->Gp) <= 160 * 1024; };
    // (measured, Float64 x 1 000 samples, <LOGSUB, GENL>: 20 000 genes 1.00 ms with <4, true> against 1.07 with <4, false>; 24 000 genes
    //  1.35 against 1.34; 30 000 genes 2.28 with <2, true> (round 4's first form) against 1.80 with <4, false>; 40 000 / 50 000 / 60 000 /
    //  65 535 genes 2.61 / 3.44 / 5.41 / 6.19 ms; the library's segmented sort: 8.9 at 60 000)
    if (G <= 20 * 1024) return launch_sample_wide<T, 4, true>(c, X, d_order, d_flags);
    if (G <= 24 * 1024) return launch_sample_wide<T, 3, true>(c, X, d_order, d_flags);
    if (fits(4)) return launch_sample_wide<T, 4, false>(c, X, d_order, d_flags);   // (up to 59 392 genes: 16-bit bins)
    return launch_sample_wide<T, 3, false>(c, X, d_order, d_flags);
}

#ifdef REO_WITH_ROCPRIM
struct SegOff {
    unsigned G;
    __host__ __device__ unsigned operator()(unsigned i) const { return i * G; }
};
#endif

template <class T>
int32_t transform_impl(reo_ctx *c)
{
    const int G = static_cast<int>(c->G), S = static_cast<int>(c->S), Gp = c->Gp;
    hipStream_t st = c->stream;

    // sorted sample order: groups contiguous, original order inside a group
    std::vector<int32_t> order(S);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(),
                     [&](int a, int b) { return c->group_id[a] < c->group_id[b]; });
    c->goff.assign(c->ngroups + 1, 0);
    for (int s = 0; s < S; ++s) c->goff[c->group_id[s] + 1]++;
    for (int g = 0; g < c->ngroups; ++g) c->goff[g + 1] += c->goff[g];
    // every group is padded to whole blocks of 32 sample slots: the pair kernel compares 32 samples of a pair
    // per instruction.  Padding slots hold lo = hi = 0, which no position is below, so they add nothing to any count.
    c->goff32.assign(c->ngroups + 1, 0);
    for (int g = 0; g < c->ngroups; ++g) c->goff32[g + 1] = c->goff32[g] + (c->goff[g + 1] - c->goff[g] + 31) / 32 * 32;
    const int S32 = c->goff32[c->ngroups], nblk = S32 / 32;
    std::vector<int32_t> slots(S), goff_blocks(c->ngroups + 1);
    for (int t = 0; t < S; ++t) {
        const int g = c->group_id[order[t]];
        slots[t] = c->goff32[g] + (t - c->goff[g]);
    }
    for (int g = 0; g < c->ngroups; ++g)  // behind the samples: the padding slots of every group (t_sample zeroes their rows)
        for (int sl = c->goff32[g] + (c->goff[g + 1] - c->goff[g]); sl < c->goff32[g + 1]; ++sl) slots.push_back(sl);
    for (int g = 0; g <= c->ngroups; ++g) goff_blocks[g] = c->goff32[g] / 32;

    // scratch lives in the context (grow-only): hipMalloc/hipFree per call cost milliseconds
    DevBuf<int32_t> &d_order = c->t_order, &d_flags = c->t_flags;
    int32_t rc;
    if ((rc = d_order.ensure(S)) || (rc = d_flags.ensure(32)) || (rc = c->t_slots.ensure(slots.size()))) return rc;
    if ((rc = c->goff_dev.ensure(c->ngroups + 1))) return rc;
    // sample order, slots and group offsets depend on the group labels only: uploaded when they (or the buffers) change
    if (order != c->t_order_host || slots != c->t_slots_host || goff_blocks != c->goff_blocks_host || d_order.p != c->t_meta_ptr[0] ||
        c->t_slots.p != c->t_meta_ptr[1] || c->goff_dev.p != c->t_meta_ptr[2]) {
        REO_HIP_CHECK(hipMemcpyAsync(d_order.p, order.data(), sizeof(int32_t) * S, hipMemcpyHostToDevice, st));
        REO_HIP_CHECK(hipMemcpyAsync(c->t_slots.p, slots.data(), sizeof(int32_t) * slots.size(), hipMemcpyHostToDevice, st));
        REO_HIP_CHECK(hipMemcpyAsync(c->goff_dev.p, goff_blocks.data(), sizeof(int32_t) * (c->ngroups + 1),
                                     hipMemcpyHostToDevice, st));
        REO_HIP_CHECK(hipStreamSynchronize(st));  // the sources are locals
        c->t_order_host = order; c->t_slots_host = slots; c->goff_blocks_host = goff_blocks;
        c->t_meta_ptr[0] = d_order.p; c->t_meta_ptr[1] = c->t_slots.p; c->t_meta_ptr[2] = c->goff_dev.p;
    }
    REO_HIP_CHECK(hipMemsetAsync(d_flags.p, 0, 6 * sizeof(int32_t), st));
    unsigned long long *d_varbits = reinterpret_cast<unsigned long long *>(d_flags.p + 2);

    const bool big = G > 65535;  // 32-bit positions and gene indices, the big plane layout
    const size_t n = static_cast<size_t>(S32) * Gp;            // numbers per intermediate row set
    const size_t nq = static_cast<size_t>(nblk) * Gp * 4;      // uint4 per plane set
    if (!big) {
        if ((rc = c->t_pos16.ensure(n)) || (rc = c->t_lo16.ensure(n)) || (rc = c->t_hi16.ensure(n)) ||
            (rc = c->pos.ensure(nq)) || (rc = c->lo.ensure(nq)) || (rc = c->hi.ensure(nq)))
            return rc;
        // (rows of padding slots and padded genes must read as zero, below no band edge: t_sample writes them itself; the
        //  segmented path clears the three row sets first, below)
    } else {
        if ((rc = c->t_pos32.ensure(n)) || (rc = c->t_lo32.ensure(n)) || (rc = c->t_hi32.ensure(n)) ||
            (rc = c->pos.ensure(static_cast<size_t>(nblk) * Gp * 5)) || (rc = c->lo.ensure(static_cast<size_t>(nblk) * Gp * 8)) ||
            (rc = c->hi.ensure(static_cast<size_t>(nblk) * Gp * 8)))
            return rc;
        REO_HIP_CHECK(hipMemsetAsync(c->t_pos32.p, 0, n * sizeof(uint32_t), st));
        REO_HIP_CHECK(hipMemsetAsync(c->t_lo32.p, 0, n * sizeof(uint32_t), st));
        REO_HIP_CHECK(hipMemsetAsync(c->t_hi32.p, 0, n * sizeof(uint32_t), st));
        REO_HIP_CHECK(hipMemsetAsync(c->lo.p, 0, static_cast<size_t>(nblk) * Gp * 8 * sizeof(uint4), st));  // (words 20..31 of a row are never written)
        REO_HIP_CHECK(hipMemsetAsync(c->hi.p, 0, static_cast<size_t>(nblk) * Gp * 8 * sizeof(uint4), st));
    }
    const T *X = static_cast<const T *>(c->dX);
    auto slice = [&]() -> int32_t {
        if (big) t_slice_big<<<dim3(Gp / 256, nblk), 256, 0, st>>>(c->t_pos32.p, c->t_lo32.p, c->t_hi32.p, Gp, c->pos.p, c->lo.p, c->hi.p);
        else t_slice<<<dim3(Gp / 512, nblk), 256, 0, st>>>(c->t_pos16.p, c->t_lo16.p, c->t_hi16.p, Gp, c->pos.p, c->lo.p, c->hi.p, 0);
        REO_HIP_CHECK(hipGetLastError());
        return REO_OK;
    };
    auto finish = [&](int has_ties, bool sliced) -> int32_t {
        if (!sliced) { const int32_t rs = slice(); if (rs) return rs; }
        c->has_ties = has_ties;
        c->transformed = true;
        return REO_OK;
    };

    // first choice: every sample ranked inside one workgroup's LDS -- t_sample's histograms for Int64 keys of at most 24
    // varying bits, t_sample_wide's buckets for everything else (Float64; wider keys; a crowded lossy bucket)
    const char *env = getenv("REO_TRANSFORM");  // "segmented": always the device-wide segmented sort; "wide": never the histogram forms (A/B tests)
    c->transform_in_lds = 0;
    // (up to 65 535 genes: positions are 16-bit numbers.  Above 32 768 genes the histogram forms use 16-bit bins / keep their low-bit
    //  rows in L2, and the bucket form keeps its gene row in L2.)
    if (G <= 65535 && !(env && env[0] == 's')) {
        if (!c->host_flags) REO_HIP_CHECK(pool_alloc(reinterpret_cast<void **>(&c->host_flags), 8 * sizeof(int32_t), true));
        if (!c->ev_flags) REO_HIP_CHECK(handle_event(&c->ev_flags, 0));
        int32_t *fl = c->host_flags;
        bool wide = !kCountingPath<T> || (env && env[0] == 'w');
        for (int attempt = 0; attempt < 2; ++attempt) {
            if ((rc = launch_lds_ranking<T>(c, X, SampleList{d_order.p, c->t_slots.p, S, S32}, d_flags.p, wide))) return rc;
            // the slicing is enqueued before the host looks at the flags (it runs while the host wakes up; if the flags send
            // the data elsewhere it is simply done again)
            if ((rc = slice())) return rc;
            // The host waits for the flags only (an event behind their copy into pinned memory); behind that event the clearing
            // of the class table is already queued, so the GPU has work while the host wakes up and launches the pair kernel.
            REO_HIP_CHECK(hipMemcpyAsync(fl, d_flags.p, 6 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
            REO_HIP_CHECK(hipEventRecord(c->ev_flags, st));
            if (!c->table_prezeroed && c->table.p && c->table.n >= static_cast<size_t>(c->G) * 4 * c->Wp) {
                REO_HIP_CHECK(hipMemsetAsync(c->table.p, 0, static_cast<size_t>(c->G) * 4 * c->Wp * sizeof(uint32_t), st));
                c->table_prezeroed = true;
            }
            REO_HIP_CHECK(hipEventSynchronize(c->ev_flags));
            if (fl[0]) {
                set_error(kNaNMessage);
                return REO_EINVAL;
            }
            if (c->debug_stamps && wide) {  // diagnostic builds (-DREO_STAMPS): marks of workgroup 0 of t_sample_wide, 10 ns units
                unsigned long long stv[8];
                REO_HIP_CHECK(hipMemcpy(stv, d_flags.p + 8, sizeof stv, hipMemcpyDeviceToHost));
                fprintf(stderr, "stamps t_sample_wide (max + flags, sample sort, histogram, prefix sums, scatter, ranks in slot order, into gene order + rows):");
                for (int k = 1; k <= 5; ++k) fprintf(stderr, " %lld", (long long)(stv[k] - stv[k - 1]));
                fprintf(stderr, " %lld %lld", (long long)(stv[7] - stv[5]), (long long)(stv[6] - stv[7]));
                fprintf(stderr, "  (x 10 ns)\n");
            }
            if (c->debug_stamps && !wide) {  // ... of t_sample's histogram form
                unsigned long long stv[6];
                REO_HIP_CHECK(hipMemcpy(stv, d_flags.p + 8, sizeof stv, hipMemcpyDeviceToHost));
                fprintf(stderr, "stamps t_sample, histogram form (column read + varying bits, clear, keys + histogram, prefix sums, rows out):");
                for (int k = 1; k <= 5; ++k) fprintf(stderr, " %lld", (long long)(stv[k] - stv[k - 1]));
                fprintf(stderr, "  (x 10 ns)\n");
            }
            if (!fl[4] && !fl[5]) {
                c->transform_in_lds = wide ? 2 : 1;
                return finish(fl[1], true);
            }
            REO_HIP_CHECK(hipMemsetAsync(d_flags.p, 0, 6 * sizeof(int32_t), st));
            if (wide) { set_error("the bucket form of the ranking flagged a sample: internal error"); return REO_EHIP; }   // (it takes every input: never seen)
            wide = true;               // some sample needs the bucket form: all of them take it
        }
    }
    if (big && !(env && env[0] == 's')) {
        // more than 65 535 genes: the bucket ranking with its by-slot records in L2 (t_sample_big); rows of padding slots and padded
        // genes keep the zeros of the memsets above.  Launched in batches when the records of all samples would not fit 4 GiB.
        const size_t lds = 8 * (kSplit1 + 16) + 4 * (static_cast<size_t>(kSplit1) * 17 + 1 + ((static_cast<size_t>(kSplit1) * 17 + 1) >> 5) + 1 + 17) + 16;
        REO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(t_sample_big<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        const int batch = static_cast<int>(std::max<size_t>(1, std::min<size_t>(static_cast<size_t>(S), (size_t(4) << 30) / (static_cast<size_t>(Gp) * 8))));
        if ((rc = c->t_kin.ensure(static_cast<size_t>(batch) * Gp))) return rc;
        for (int s0 = 0; s0 < S; s0 += batch) {
            const int ns = std::min(batch, S - s0);
            t_sample_big<T><<<static_cast<unsigned>(ns), 1024, lds, st>>>(X, c->ld, d_order.p + s0, c->t_slots.p + s0, G, Gp, c->t_pos32.p, c->t_lo32.p, c->t_hi32.p,
                                                                      d_flags.p, c->t_kin.p);
            REO_HIP_CHECK(hipGetLastError());
        }
        if ((rc = slice())) return rc;
        int32_t flags[2] = {0, 0};
        REO_HIP_CHECK(hipMemcpyAsync(flags, d_flags.p, sizeof flags, hipMemcpyDeviceToHost, st));
        REO_HIP_CHECK(hipStreamSynchronize(st));
        if (flags[0]) {
            set_error(kNaNMessage);
            return REO_EINVAL;
        }
        c->transform_in_lds = 3;   // (ranked inside one workgroup per sample; rows in L2)
        return finish(flags[1], true);
    }
#ifdef REO_WITH_ROCPRIM
    if (!big) {  // the segmented path writes the genes of the samples only: padding slots and padded genes read as zero
        REO_HIP_CHECK(hipMemsetAsync(c->t_pos16.p, 0, n * sizeof(uint16_t), st));
        REO_HIP_CHECK(hipMemsetAsync(c->t_lo16.p, 0, n * sizeof(uint16_t), st));
        REO_HIP_CHECK(hipMemsetAsync(c->t_hi16.p, 0, n * sizeof(uint16_t), st));
    }

    // column batches: rocprim takes a 32-bit element count
    const int CB = std::max(1, std::min(std::min(S, 65535), static_cast<int>((1u << 27) / static_cast<unsigned>(G))));  // (grid y <= 65535)
    DevBuf<uint64_t> &k_in = c->t_kin, &k_out = c->t_kout;
    const size_t bn = static_cast<size_t>(CB) * G;
    if ((rc = k_in.ensure(bn)) || (rc = k_out.ensure(bn))) return rc;
    auto seg_begin = rocprim::make_transform_iterator(rocprim::counting_iterator<unsigned>(0),
                                                      SegOff{static_cast<unsigned>(G)});
    auto segmented = [&](auto &v_in, auto &v_out, auto *pos, auto *lo, auto *hi) -> int32_t {
        using IdxT = std::remove_pointer_t<decltype(pos)>;
        int32_t rc2;
        if ((rc2 = v_in.ensure(bn)) || (rc2 = v_out.ensure(bn))) return rc2;
        size_t temp_bytes = 0;
        REO_HIP_CHECK(rocprim::segmented_radix_sort_pairs(nullptr, temp_bytes, k_in.p, k_out.p, v_in.p, v_out.p,
                                                          static_cast<unsigned>(bn), static_cast<unsigned>(CB),
                                                          seg_begin, seg_begin + 1, 0, 64, st));
        DevBuf<unsigned char> &temp = c->t_temp;
        if ((rc2 = temp.ensure(std::max<size_t>(temp_bytes, 16)))) return rc2;
        for (int cb0 = 0; cb0 < S; cb0 += CB) {
            const int nc = std::min(CB, S - cb0);
            dim3 grid((G + 255) / 256, nc);
            REO_HIP_CHECK(hipMemsetAsync(d_varbits, 0, sizeof(unsigned long long), st));
            t_keys<T, IdxT><<<grid, 256, 0, st>>>(X, c->ld, d_order.p, G, cb0, k_in.p, v_in.p, d_flags.p, d_varbits);
            // radix-sort only the bit range in which the keys of this batch differ at all (rank-like data: 15 bits)
            unsigned long long vb = 0;
            REO_HIP_CHECK(hipMemcpyAsync(&vb, d_varbits, sizeof vb, hipMemcpyDeviceToHost, st));
            REO_HIP_CHECK(hipStreamSynchronize(st));
            unsigned begin_bit = 0, end_bit = 64;
            if (vb == 0) { begin_bit = 0; end_bit = 1; }
            else { begin_bit = static_cast<unsigned>(__builtin_ctzll(vb)); end_bit = 64u - static_cast<unsigned>(__builtin_clzll(vb)); }
            size_t tb = temp_bytes;
            REO_HIP_CHECK(rocprim::segmented_radix_sort_pairs(temp.p, tb, k_in.p, k_out.p, v_in.p, v_out.p,
                                                              static_cast<unsigned>(static_cast<size_t>(nc) * G),
                                                              static_cast<unsigned>(nc), seg_begin, seg_begin + 1,
                                                              begin_bit, end_bit, st));
            t_bands<T, IdxT><<<grid, 256, 0, st>>>(k_out.p, v_out.p, G, Gp, cb0, c->t_slots.p, pos, lo, hi, d_flags.p + 1);
        }
        return REO_OK;
    };
    if (big) rc = segmented(c->t_vin32, c->t_vout32, c->t_pos32.p, c->t_lo32.p, c->t_hi32.p);
    else rc = segmented(c->t_vin, c->t_vout, c->t_pos16.p, c->t_lo16.p, c->t_hi16.p);
    if (rc) return rc;
    REO_HIP_CHECK(hipGetLastError());
    int32_t flags[2] = {0, 0};
    REO_HIP_CHECK(hipMemcpyAsync(flags, d_flags.p, sizeof flags, hipMemcpyDeviceToHost, st));
    REO_HIP_CHECK(hipStreamSynchronize(st));
    if (flags[0]) {
        set_error(kNaNMessage);
        return REO_EINVAL;
    }
    return finish(flags[1], false);
#else
    (void)finish; (void)d_varbits;
    set_error("REO_TRANSFORM=segmented: this build of libreo_hip carries no segmented sort (the library's rocprim call left the product path in "
              "round 5; `make ROCPRIM=1` in csrc builds the A/B variant)");
    return REO_EINVAL;
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Narrowed upload of Int64 matrices (round 5).  PCIe is the floor of the drop-in call's upload (57 GB/s measured from pageable memory:
// 2.8 ms for the 160 MB of config 3), and Matrix(df_expr) of count data or ranks is Int64 numbers that fit 16 or 32 bits.  A few host
// threads convert each chunk into a pinned staging buffer as 16-bit (then 32-bit) numbers -- 140 GB/s of source with 8 threads,
// tools/microbench_narrow.cpp -- the link carries a quarter (half) of the bytes, and a kernel widens them into the Int64 matrix in
// HBM that every other kernel reads.  Exact: a chunk with a value that does not fit is converted again one width up (the widths only
// grow: 2 -> 4 -> 8 bytes = the caller's array itself), so what arrives is the caller's matrix, bit for bit.
// NUMA (round 6): a GPU box of the pool has two sockets, the GPU hangs off one of them, and the narrowing threads and their pinned staging
// slots work best on that socket's memory (measured: the whole process bound to the GPU's node 6.37 ms per drop-in call at config 3, to
// the other 6.55: profiles/r6_n_numa_probe.txt).  The library binds what is its own -- the pool's worker threads, and the calling thread
// for the moment in which it allocates the staging slots -- to the CPUs of the GPU's node, intersected with the CPUs the process may
// use; the caller's threads and arrays are the caller's.  REO_NUMA=0 switches it off; any failure leaves things as they were.
static bool device_node_cpus(int dev, cpu_set_t *out)
{
    static std::mutex mu;
    static std::vector<std::pair<int, cpu_set_t>> *cache = new std::vector<std::pair<int, cpu_set_t>>();   // (device, its node's CPUs; empty set: unknown)
    std::lock_guard<std::mutex> lk(mu);
    for (const auto &e : *cache)
        if (e.first == dev) { *out = e.second; return CPU_COUNT(out) > 0; }
    cpu_set_t set;
    CPU_ZERO(&set);
    const char *sw = getenv("REO_NUMA");
    char bdf[64] = {0};
    if (!(sw && sw[0] == '0') && hipDeviceGetPCIBusId(bdf, sizeof bdf, dev) == hipSuccess) {
        for (char *q = bdf; *q; ++q) *q = static_cast<char>(tolower(*q));
        char path[160];
        snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
        int node = -1;
        if (FILE *f = fopen(path, "r")) { if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
        if (node >= 0) {
            snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
            if (FILE *f = fopen(path, "r")) {   // "0-63,128-191"
                int a, b;
                char sep;
                while (fscanf(f, "%d", &a) == 1) {
                    b = a;
                    if (fscanf(f, "%c", &sep) == 1 && sep == '-') { if (fscanf(f, "%d", &b) != 1) b = a; if (fscanf(f, "%c", &sep) != 1) sep = 0; }
                    for (int cpu = a; cpu <= b && cpu < CPU_SETSIZE; ++cpu) CPU_SET(cpu, &set);
                    if (sep != ',') break;
                }
                fclose(f);
            }
            cpu_set_t allowed;
            if (sched_getaffinity(0, sizeof allowed, &allowed) == 0) CPU_AND(&set, &set, &allowed);   // never outside what the process may use
        }
    }
    cache->emplace_back(dev, set);
    *out = set;
    return CPU_COUNT(out) > 0;
}

struct ScopedNodeAffinity {   // the calling thread on the device's node for the lifetime of the object (first touch of pinned memory)
    cpu_set_t saved;
    bool active = false;
    explicit ScopedNodeAffinity(int dev)
    {
        cpu_set_t want;
        if (!device_node_cpus(dev, &want) || sched_getaffinity(0, sizeof saved, &saved) != 0) return;
        active = sched_setaffinity(0, sizeof want, &want) == 0;
    }
    ~ScopedNodeAffinity() { if (active) (void)sched_setaffinity(0, sizeof saved, &saved); }
};

class HostPool {   // process-wide worker threads (started on first use, asleep otherwise, never joined: the process ends with them)
public:
    // the workers move to these CPUs before their next job (the node of the device whose context asked last; contexts of GPUs on
    // different sockets in one process just move them back and forth between jobs)
    void bind(const cpu_set_t &set)
    {
        std::lock_guard<std::mutex> lk(mu_);
        if (aff_gen_ != 0 && CPU_EQUAL(&set, &aff_)) return;
        aff_ = set; ++aff_gen_;
    }
    static HostPool &get(int want)
    {
        static HostPool *inst = new HostPool();
        HostPool *p = inst;
        std::lock_guard<std::mutex> lk(p->mu_);
        while (static_cast<int>(p->th_.size()) < want - 1) {   // (the caller is a worker too)
            p->th_.emplace_back([p] { p->worker(); });
            p->th_.back().detach();
        }
        return *p;
    }
    // fn(task) for task = 0 .. ntasks - 1, on the workers and the calling thread; returns when every task is done
    void run(int ntasks, const std::function<void(int)> &fn)
    {
        std::lock_guard<std::mutex> one_job(run_mu_);   // contexts of different threads share the pool: their jobs take turns
        {
            // a worker of the LAST job may still be on its way out of work() (its final fetch of a task index): the counters are not
            // touched before it has left, or that fetch could land between the two stores below and hand out a task twice
            std::unique_lock<std::mutex> lk(mu_);
            idle_.wait(lk, [&] { return active_ == 0; });
            job_ = &fn; total_.store(ntasks); next_.store(0); pending_ = ntasks; ++gen_;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [&] { return pending_ == 0; });
        job_ = nullptr;
    }
private:
    void work()
    {
        for (;;) {
            const int t = next_.fetch_add(1);
            if (t >= total_.load()) return;
            (*job_)(t);
            std::lock_guard<std::mutex> lk(mu_);
            if (--pending_ == 0) done_.notify_all();
        }
    }
    void worker()
    {
        uint64_t seen = 0, my_aff = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (!job_) continue;   // (woken late: that job is over.  A job that IS set is completely set up: decided under the lock)
                ++active_;
                if (my_aff != aff_gen_) { my_aff = aff_gen_; (void)sched_setaffinity(0, sizeof aff_, &aff_); }
            }
            work();
            std::lock_guard<std::mutex> lk(mu_);
            if (--active_ == 0) idle_.notify_all();
        }
    }
    std::mutex mu_, run_mu_;
    std::condition_variable cv_, done_, idle_;
    cpu_set_t aff_;                // where the workers run (bind)
    uint64_t aff_gen_ = 0;
    int active_ = 0;               // workers inside work()
    std::vector<std::thread> th_;
    const std::function<void(int)> *job_ = nullptr;
    std::atomic<int> next_{0}, total_{0};
    int pending_ = 0;
    uint64_t gen_ = 0;
};

template <class N>
static bool narrow_columns(const int64_t *src, int64_t ld, int64_t G, int col0, int col1, N *dst)   // false: some value does not fit N
{
    int64_t bad = 0;
    for (int cidx = col0; cidx < col1 && !bad; ++cidx) {   // (a column that does not fit ends this thread's share: the chunk is redone wider)
        const int64_t *s = src + static_cast<int64_t>(cidx) * ld;
        N *d = dst + static_cast<int64_t>(cidx) * G;
        for (int64_t i = 0; i < G; ++i) { const int64_t v = s[i]; const N w = static_cast<N>(v); d[i] = w; bad |= v ^ static_cast<int64_t>(w); }
    }
    return bad == 0;
}

template <class N, class W>
__global__ __launch_bounds__(256) void t_widen(const N *__restrict__ src, W *__restrict__ dst, size_t n)
{
    const size_t base = static_cast<size_t>(blockIdx.x) * 2048 + threadIdx.x;   // consecutive lanes, consecutive numbers: no alignment to care about
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const size_t i = base + static_cast<size_t>(k) * 256;
        if (i < n) dst[i] = static_cast<W>(src[i]);   // (int16 / int32 -> Int64 or Float64, float -> Float64: all exact)
    }
}

// Float64 columns whose values are integers (counts read into a Float64 DataFrame) or float32 numbers (data that was stored in single
// precision) cross the link as int16 / int32 / float32 too: the chunk is narrowed only if EVERY value converts back to the same
// bits (so -0.0, NaN and anything with more mantissa than the narrow type keeps the chunk on the wider form).
template <class N>
static bool narrow_columns_f64(const double *src, int64_t ld, int64_t G, int col0, int col1, N *dst)
{
    bool ok = true;
    for (int cidx = col0; cidx < col1 && ok; ++cidx) {
        const double *s = src + static_cast<int64_t>(cidx) * ld;
        N *d = dst + static_cast<int64_t>(cidx) * G;
        if constexpr (std::is_same<N, float>::value) {
            for (int64_t i = 0; i < G; ++i) { const double v = s[i]; const float q = static_cast<float>(v); d[i] = q; ok &= static_cast<double>(q) == v; }
        } else {
            constexpr double lo = static_cast<double>(std::numeric_limits<N>::min()), hi = static_cast<double>(std::numeric_limits<N>::max());
            for (int64_t i = 0; i < G; ++i) {
                const double v = s[i];
                const bool in = v >= lo && v <= hi;            // (false for NaN; the conversion below is undefined outside the range)
                const N q = in ? static_cast<N>(v) : N(0);
                d[i] = q;
                ok &= in && static_cast<double>(q) == v && !(v == 0.0 && std::signbit(v));
            }
        }
    }
    return ok;
}

// One chunk of columns of a host matrix into its place in the device matrix, on the context's upload stream: Int64 chunks narrowed
// (host thread pool -> pinned staging slot -> link -> t_widen), and so do Float64 chunks whose values are all integers or all float32
// numbers; everything else -- other Float64 data, REO_UPLOAD_THREADS=0, a chunk with a value beyond 32 bits and every chunk after it --
// goes straight from the caller's array.  send() returns the event behind which the chunk is
// in place.  Used by the pipelined reo_set_matrix (eager_upload), by the plain one and by the dense pseudo-bulk call (upload_columns).
template <class T>
struct ChunkUploader {
    static constexpr int kStage = 3;
    enum Form { I16 = 0, I32 = 1, F32 = 2, RAW = 3 };   // what a chunk is on the link; the form only moves up this ladder (F32: Float64 input only)
    reo_ctx *c;
    const T *hX;
    int64_t hld, G;
    T *dX;
    int form = RAW, nslot = 0, nthreads = 1, nraw = 0;

    int32_t init(reo_ctx *ctx, const T *host, int64_t ld, int64_t genes, T *dev, int max_cols)
    {
        c = ctx; hX = host; hld = ld; G = genes; dX = dev;
        int32_t rc;
        if ((rc = ensure_upload_streams(c))) return rc;
        c->narrowed_bytes = 0;
        if (c->upload_threads > 0) {
            if ((rc = ensure_staging(c, static_cast<size_t>(max_cols) * G * 4))) return rc;
            nthreads = std::max(1, std::min<int>(c->upload_threads, static_cast<int>(std::thread::hardware_concurrency())));
            cpu_set_t node;
            if (device_node_cpus(c->device, &node)) HostPool::get(nthreads).bind(node);
            form = I16;
        }
        return REO_OK;
    }

    int32_t send(int c0, int nc, hipEvent_t *ready)
    {
        // the slot's pinned half is free when the copy that read it is done (ev_stage); its device half when the widening is (the next
        // copy into it follows on the same stream)
        while (form != RAW) {
            if (form == F32 && !std::is_same<T, double>::value) { form = RAW; break; }
            {   // a probe first: the head of the chunk's first column decides in microseconds what real Float64 data (or wide integers)
                // would otherwise find out after a whole converted chunk per form
                unsigned char probe[256 * 4];
                const int64_t np_ = std::min<int64_t>(G, 256);
                bool ok;
                if constexpr (std::is_same<T, int64_t>::value) {
                    const int64_t *src = hX + static_cast<int64_t>(c0) * hld;
                    ok = form == I16 ? narrow_columns<int16_t>(src, hld, np_, 0, 1, reinterpret_cast<int16_t *>(probe))
                                     : narrow_columns<int32_t>(src, hld, np_, 0, 1, reinterpret_cast<int32_t *>(probe));
                } else {
                    const double *src = hX + static_cast<int64_t>(c0) * hld;
                    ok = form == I16 ? narrow_columns_f64<int16_t>(src, hld, np_, 0, 1, reinterpret_cast<int16_t *>(probe))
                       : form == I32 ? narrow_columns_f64<int32_t>(src, hld, np_, 0, 1, reinterpret_cast<int32_t *>(probe))
                                     : narrow_columns_f64<float>(src, hld, np_, 0, 1, reinterpret_cast<float *>(probe));
                }
                if (!ok) { ++form; continue; }
            }
            const int sl = nslot % kStage;
            if (nslot >= kStage) REO_HIP_CHECK(hipEventSynchronize(c->ev_stage[sl]));
            unsigned char *hs = c->stage_h[sl];
            std::atomic<int> fits{1};
            const int per = (nc + nthreads - 1) / nthreads;
            const int f = form;
            HostPool::get(nthreads).run(nthreads, [&](int t) {
                const int a = std::min(nc, t * per), b = std::min(nc, a + per);
                if (a >= b) return;
                bool ok;
                if constexpr (std::is_same<T, int64_t>::value) {
                    const int64_t *src = hX + static_cast<int64_t>(c0) * hld;
                    ok = f == I16 ? narrow_columns<int16_t>(src, hld, G, a, b, reinterpret_cast<int16_t *>(hs))
                                  : narrow_columns<int32_t>(src, hld, G, a, b, reinterpret_cast<int32_t *>(hs));
                } else {
                    const double *src = hX + static_cast<int64_t>(c0) * hld;
                    ok = f == I16 ? narrow_columns_f64<int16_t>(src, hld, G, a, b, reinterpret_cast<int16_t *>(hs))
                       : f == I32 ? narrow_columns_f64<int32_t>(src, hld, G, a, b, reinterpret_cast<int32_t *>(hs))
                                  : narrow_columns_f64<float>(src, hld, G, a, b, reinterpret_cast<float *>(hs));
                }
                if (!ok) fits.store(0);
            });
            if (!fits.load()) { ++form; continue; }   // (this chunk again, one form up)
            const size_t nel = static_cast<size_t>(nc) * G, width = form == I16 ? 2 : 4;
            // copy and widening both on the upload stream: the staging ring turns whatever the other streams are waiting for
            REO_HIP_CHECK(hipMemcpyAsync(c->stage_d[sl].p, hs, nel * width, hipMemcpyHostToDevice, c->up));
            REO_HIP_CHECK(hipEventRecord(c->ev_stage[sl], c->up));
            T *dst = dX + static_cast<size_t>(c0) * G;
            const unsigned grid = static_cast<unsigned>((nel + 2047) / 2048);
            const unsigned char *ds = c->stage_d[sl].p;
            if (form == I16) t_widen<int16_t, T><<<grid, 256, 0, c->up>>>(reinterpret_cast<const int16_t *>(ds), dst, nel);
            else if (form == I32) t_widen<int32_t, T><<<grid, 256, 0, c->up>>>(reinterpret_cast<const int32_t *>(ds), dst, nel);
            else t_widen<float, T><<<grid, 256, 0, c->up>>>(reinterpret_cast<const float *>(ds), dst, nel);
            REO_HIP_CHECK(hipGetLastError());
            REO_HIP_CHECK(hipEventRecord(c->ev_widen[sl], c->up));
            *ready = c->ev_widen[sl];
            ++nslot;
            c->narrowed_bytes += static_cast<int64_t>(nel * width);
            return REO_OK;
        }
        // the caller's array as it is: pageable source, so the call returns when the runtime has staged the chunk
        if (hld == G) REO_HIP_CHECK(hipMemcpyAsync(dX + static_cast<size_t>(c0) * G, hX + static_cast<size_t>(c0) * hld, static_cast<size_t>(nc) * G * sizeof(T), hipMemcpyHostToDevice, c->up));
        else REO_HIP_CHECK(hipMemcpy2DAsync(dX + static_cast<size_t>(c0) * G, G * sizeof(T), hX + static_cast<size_t>(c0) * hld, hld * sizeof(T), G * sizeof(T), nc, hipMemcpyHostToDevice, c->up));
        hipEvent_t ev = c->ev_up[nraw++ % 8];
        REO_HIP_CHECK(hipEventRecord(ev, c->up));
        *ready = ev;
        c->narrowed_bytes += static_cast<int64_t>(nc) * G * 8;
        return REO_OK;
    }
};

// ---------------------------------------------------------------------------------------------------------------------------------
// Pipelined upload (round 5): reo_set_matrix_i64 / _f64 from HOST memory when groups (and thresholds) are already known.
// The drop-in call hands over a pageable column-major host matrix (julia/RankCompV3HIP.jl at src/RankCompV3.jl:652); the transform
// is per sample and the pair kernel's items are per side (= per group), so nothing has to wait for the whole matrix:
//   * the columns travel in chunks on an upload stream; as soon as a chunk has arrived its samples are ranked (the same
//     t_sample / t_sample_wide launches as run_transform, on a list of that chunk's columns);
//   * when the last sample of a group has been ranked the group's blocks are sliced into bit planes, and -- two groups, one
//     GPU, thresholds set -- the pair kernel's items of THAT side are launched (launch_k1 with a side mask) while the other
//     group's columns are still on their way;
//   * the call returns when the whole matrix has been read (the ownership rule of reo_hip.h: no host pointer is kept), with the
//     second side's pair kernel possibly still running (like reo_build_pairs on one GPU); reo_build_pairs then has nothing to do.
// Anything the in-LDS ranking cannot take (a flagged sample: another form is needed) falls back to run_transform on the
// resident copy at reo_build_pairs -- same results, no overlap.
template <class T>
int32_t eager_upload_impl(reo_ctx *c, const T *hX, int64_t hld, bool with_k1)
{
    const auto w_begin = std::chrono::steady_clock::now();
    auto stamp = [&](const char *what) {   // REO_DEBUG_PASSES: where the host is, microseconds since the call began
        if (c->debug_passes) fprintf(stderr, "  eager_upload: %s at %.0f us\n", what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w_begin).count());
    };
    const int G = static_cast<int>(c->G), S = static_cast<int>(c->S), Gp = c->Gp;
    hipStream_t st = c->stream;
    int32_t rc;
    // sample layout: as transform_impl, but the samples are LISTED IN COLUMN ORDER (a chunk of columns = a run of the list)
    c->goff.assign(c->ngroups + 1, 0);
    for (int s = 0; s < S; ++s) c->goff[c->group_id[s] + 1]++;
    for (int g = 0; g < c->ngroups; ++g) c->goff[g + 1] += c->goff[g];
    c->goff32.assign(c->ngroups + 1, 0);
    for (int g = 0; g < c->ngroups; ++g) c->goff32[g + 1] = c->goff32[g] + (c->goff[g + 1] - c->goff[g] + 31) / 32 * 32;
    const int S32 = c->goff32[c->ngroups], nblk = S32 / 32;
    std::vector<int32_t> lists(static_cast<size_t>(S) + S32), goff_blocks(c->ngroups + 1), seen(c->ngroups, 0);
    int32_t *cols = lists.data(), *slots = lists.data() + S;   // [S] columns (identity), then [S samples + padding] slots
    for (int s = 0; s < S; ++s) {
        const int g = c->group_id[s];
        cols[s] = s;
        slots[s] = c->goff32[g] + seen[g]++;   // the order inside a group = column order, like the stable sort of transform_impl
    }
    int npad = 0;
    for (int g = 0; g < c->ngroups; ++g)
        for (int sl = c->goff32[g] + (c->goff[g + 1] - c->goff[g]); sl < c->goff32[g + 1]; ++sl) slots[S + npad++] = sl;
    for (int g = 0; g <= c->ngroups; ++g) goff_blocks[g] = c->goff32[g] / 32;
    DevBuf<int32_t> &d_flags = c->t_flags;
    if ((rc = c->e_lists.ensure(lists.size())) || (rc = d_flags.ensure(32)) || (rc = c->goff_dev.ensure(c->ngroups + 1))) return rc;
    REO_HIP_CHECK(hipMemcpyAsync(c->e_lists.p, lists.data(), sizeof(int32_t) * (static_cast<size_t>(S) + S + npad), hipMemcpyHostToDevice, st));
    REO_HIP_CHECK(hipMemcpyAsync(c->goff_dev.p, goff_blocks.data(), sizeof(int32_t) * (c->ngroups + 1), hipMemcpyHostToDevice, st));
    REO_HIP_CHECK(hipStreamSynchronize(st));  // the sources are locals
    c->goff_blocks_host = goff_blocks; c->t_meta_ptr[2] = c->goff_dev.p;
    const int32_t *d_cols = c->e_lists.p, *d_slots = c->e_lists.p + S;
    REO_HIP_CHECK(hipMemsetAsync(d_flags.p, 0, 6 * sizeof(int32_t), st));
    const size_t n = static_cast<size_t>(S32) * Gp, nq = static_cast<size_t>(nblk) * Gp * 4;
    if ((rc = c->t_pos16.ensure(n)) || (rc = c->t_lo16.ensure(n)) || (rc = c->t_hi16.ensure(n)) ||
        (rc = c->pos.ensure(nq)) || (rc = c->lo.ensure(nq)) || (rc = c->hi.ensure(nq)))
        return rc;
    if (!c->host_flags) REO_HIP_CHECK(pool_alloc(reinterpret_cast<void **>(&c->host_flags), 8 * sizeof(int32_t), true));
    // Three streams.  `up` carries the copies and the widening of narrowed chunks and never waits for a kernel of the other two.  `rk`
    // (high priority) ranks and slices the chunks as they arrive.  The context's own stream `st` runs the pair kernel's sides, each
    // behind the slicing of its group (an event), so that side 0 counts while group 1's chunks are still being copied, widened and
    // ranked beside it -- on one stream the ranking of group 1 (and, with it, the staging ring of the narrowed upload) queued up behind
    // side 0.
    if ((rc = ensure_upload_streams(c))) return rc;
    hipStream_t rk = c->rk;
    struct OnStream {   // the transform's launchers enqueue on the context's stream: it is `rk` while one of these lives
        reo_ctx *c; hipStream_t saved;
        OnStream(reo_ctx *ctx, hipStream_t s) : c(ctx), saved(ctx->stream) { c->stream = s; }
        ~OnStream() { c->stream = saved; }
    };
    T *dX = reinterpret_cast<T *>(c->dX_owned.p);
    const bool wide = !kCountingPath<T>;
    c->transform_in_lds = 0;
    c->table_prezeroed = false;
    if (with_k1) {
        // the pair kernel's unit map and both sides' item lists go up NOW, while the context's stream is still empty: launch_k1 uploads
        // them with a host wait on that stream, which behind a side's WaitEvent would block the host until that group is ranked (and no
        // further chunk would be converted or sent meanwhile)
        for (int side = 0; side < 2; ++side)
            if ((rc = launch_k1(c, 0, 1 << side, true, nullptr, nullptr, true))) return rc;
    }
    REO_HIP_CHECK(hipEventRecord(c->ev_rk[0], st));            // (the cleared flags, the lists)
    REO_HIP_CHECK(hipStreamWaitEvent(rk, c->ev_rk[0], 0));
    if (with_k1) {   // the class table is cleared once, in front of everything; the side launches keep what is there
        REO_HIP_CHECK(hipMemsetAsync(c->table.p, 0, static_cast<size_t>(c->G) * 4 * c->Wp * sizeof(uint32_t), st));
    }
    // (Round 6, tried and dropped: the pair kernel on a stream whose CU mask leaves 4 / 8 / 16 / 32 compute units to the whole-CU ranking
    //  workgroups of the chunks that arrive while a range is being counted -- hipExtStreamCreateWithCUMask.  Config 4, one context per
    //  process: 41.8 / 32.0 / 40.9 / 26.5 ms against 26.1-26.5 without a mask; config 3 6.8 / 8.1 (8 / 16 CUs) against 6.4.  The item list
    //  is dealt for workgroup b on XCD b & 7, which a masked queue no longer honours: profiles/r6_o_cu_mask_and_ranges_one_context_per_process.txt.)
    if (npad) {   // rows of zeros for the padding slots of every group (no data needed)
        OnStream on(c, rk);
        tic(c, 0);
        rc = launch_lds_ranking<T>(c, dX, SampleList{d_cols, d_slots + S, 0, npad}, d_flags.p, wide);
        toc(c);
        if (rc) return rc;
    }
    // Chunks: a ranking launch is one workgroup per sample and takes a sample's time (30 us for counts, 180 us for Float64) however few
    // samples it has, so a chunk wants enough samples to fill the 256 CUs (measured at 20 000 x 1 000, tools/from_host_breakdown.py:
    // chunks of 52 columns made the Float64 ranking 4.4 ms in all instead of 1.0); and it ends where the group label changes, so
    // that a group's last chunk -- the one that releases its side of the pair kernel -- is not held up by columns of the next group.
    int CH = static_cast<int>(std::max<int64_t>(256, (int64_t(8) << 20) / (static_cast<int64_t>(G) * 8)));
    if (c->eager_chunk > 0) CH = c->eager_chunk;   // REO_EAGER_CHUNK (experiments)
    std::vector<int> left(c->ngroups);
    for (int g = 0; g < c->ngroups; ++g) left[g] = c->goff[g + 1] - c->goff[g];
    // RANGE ITEMS (round 6): a side of the pair kernel need not wait for its whole group.  The slots of a group fill in column order, so
    // after any chunk the group's first (ranked samples / 32) blocks are complete: they are sliced and the side's items are launched over
    // THAT RANGE of blocks, their counts parked (kernels.hip, K1Args::park) until the launch of the group's last range adds everything up
    // and classifies.  A range costs every item its fixed part again and 2 x 2 bytes per pair of parking traffic, so there are few: the
    // first one early (it is what the call's exposed upload time shrinks to), the rest large.
    std::vector<int> ranked_g(c->ngroups, 0), done_blk(c->ngroups, 0);
    c->eager_range_launches = 0;
    const int min_last = c->eager_ranges > 0 ? 1 : 4;   // blocks that must be left for the last range (an explicit REO_EAGER_RANGES: tests cut anywhere)
    // the park buffers hold 2 x 16 KB per work item and side: 4 Gp^2 bytes for both sides (0.6 GB at config 3, 3.8 at config 4, 17 GB at
    // 65 535 genes).  Ranges are a speed-up, not a need: where that is more than a quarter of the free device memory the sides stay whole.
    bool park_fits = true;
    if (with_k1 && c->eager_ranges != 1) {
        size_t free_b = 0, total_b = 0;
        const size_t have = (c->k1_park[0].n + c->k1_park[1].n) * sizeof(uint32_t), want = size_t(4) * Gp * Gp;
        if (want > have && hipMemGetInfo(&free_b, &total_b) == hipSuccess && want - have > free_b / 4) park_fits = false;
    }
    auto range_need = [&](int total_blk, bool first) -> int {   // complete, unlaunched blocks a range waits for (the last range takes what is left)
        int n = c->eager_ranges;
        if (n < 0) n = total_blk >= 32 ? 3 : (total_blk >= 12 ? 2 : 1);
        if (!park_fits) n = 1;
        if (n <= 1 || !with_k1) return total_blk;
        const int f = std::max(1, total_blk >> (n - 1));                 // first range: 1/2, 1/4, 1/8 ... of the side
        return first ? f : std::max(1, (total_blk - f + n - 2) / (n - 1));   // the others share the rest
    };
    int sides_done = 0, ranked = 0, ranked_at_read = -1, nsides_launched = 0;
    bool fallback = false, bad_values = false;
    int32_t *fl = c->host_flags;
    auto read_flags = [&]() -> int32_t {   // the flags so far (waits for the rankings queued so far, not for the pair kernel)
        REO_HIP_CHECK(hipMemcpyAsync(fl, d_flags.p, 6 * sizeof(int32_t), hipMemcpyDeviceToHost, rk));
        REO_HIP_CHECK(hipStreamSynchronize(rk));
        ranked_at_read = ranked;
        if (fl[0]) bad_values = true;
        if (fl[4] || fl[5]) fallback = true;
        return REO_OK;
    };
    ChunkUploader<T> upl;
    if ((rc = upl.init(c, hX, hld, G, dX, std::min(CH, S)))) return rc;
    stamp("buffers, streams, lists ready");
    for (int c0 = 0; c0 < S;) {
        int nc = std::min(CH, S - c0);
        for (int s = c0 + 1; s < c0 + nc; ++s)   // cut at the first change of label that leaves a chunk worth launching
            if (c->group_id[s] != c->group_id[s - 1] && s - c0 >= std::min(CH, 64)) { nc = s - c0; break; }
        hipEvent_t arrived = nullptr;
        if ((rc = upl.send(c0, nc, &arrived))) return rc;
        REO_HIP_CHECK(hipStreamWaitEvent(rk, arrived, 0));   // the upload stream never waits for a kernel of the other two
        const int cbeg = c0;
        c0 += nc;
        if (fallback || bad_values) continue;   // (the rest of the matrix still has to arrive)
        // ranges of blocks that this chunk completes: (group, first block, one past the last, first / last range of the group)
        struct Ready { int g, b0, b1; bool first, last; };
        std::vector<Ready> ready;
        {
            OnStream on(c, rk);
            tic(c, 0);
            rc = launch_lds_ranking<T>(c, dX, SampleList{d_cols + cbeg, d_slots + cbeg, nc, nc}, d_flags.p, wide);
            ranked = c0;
            std::vector<int> touched;
            for (int s = cbeg; s < cbeg + nc; ++s) {
                const int g = c->group_id[s];
                ++ranked_g[g]; --left[g];
                if ((touched.empty() || touched.back() != g) && std::find(touched.begin(), touched.end(), g) == touched.end()) touched.push_back(g);
            }
            for (int g : touched) {
                if (rc) break;
                const int total_blk = (c->goff32[g + 1] - c->goff32[g]) / 32;
                const int complete = left[g] == 0 ? total_blk : ranked_g[g] / 32;
                const bool whole_sides = !(with_k1 && g < 2);
                const bool last = left[g] == 0;
                if (!last && (whole_sides || complete - done_blk[g] < range_need(total_blk, done_blk[g] == 0) || total_blk - complete < min_last)) continue;
                if (complete <= done_blk[g]) continue;   // (a group without samples has no blocks)
                const int b0 = c->goff32[g] / 32 + done_blk[g], nb = complete - done_blk[g];
                t_slice<<<dim3(Gp / 512, nb), 256, 0, rk>>>(c->t_pos16.p, c->t_lo16.p, c->t_hi16.p, Gp, c->pos.p, c->lo.p, c->hi.p, b0);
                if (hipGetLastError() != hipSuccess) { set_error("t_slice launch failed"); rc = REO_EHIP; }
                ready.push_back(Ready{g, done_blk[g], complete, done_blk[g] == 0, last});
                done_blk[g] = complete;
            }
            toc(c);
            if (rc) return rc;
        }
        for (const Ready &r : ready) {   // two groups: side 0 counts group 0's samples, side 1 group 1's (k = 0)
            if (!with_k1 || r.g > 1) continue;
            // The side (or a range of its blocks) is launched behind the slicing of those blocks (an event) as k1w_pairs_gated, which looks
            // at the transform's flags on the device and takes the tie form they ask for -- "ties seen so far" covers every sample ranked
            // so far, and the tie-free loop is exact on samples without ties.  The host does not wait for anything here (it used to read
            // the flags: 70 us of idle GPU per side, profiles/r5_pipelined_upload_timeline_config4.txt); a flagged sample makes the
            // launch return at once, and the host finds out at the end.
            const int mask = 1 << r.g;
            const K1Range range{r.b0, r.b1, r.first, r.last};
            const K1Range *rp = (r.first && r.last) ? nullptr : &range;
            if (c->eager_gate) {
                hipEvent_t ev = c->ev_rk[2 + (nsides_launched++ & 1)];
                REO_HIP_CHECK(hipEventRecord(ev, rk));
                REO_HIP_CHECK(hipStreamWaitEvent(st, ev, 0));
                if ((rc = launch_k1(c, 0, mask, true, d_flags.p, rp))) return rc;
            } else {   // REO_EAGER_GATE=0: the host reads the flags and picks the tie form itself (the planes are in place: it has waited for rk)
                if ((rc = read_flags())) return rc;
                if (fallback || bad_values) break;
                c->has_ties = fl[1];
                if ((rc = launch_k1(c, 0, mask, true, nullptr, rp))) return rc;
            }
            if (rp) ++c->eager_range_launches;
            stamp(r.last ? "a side of the pair kernel launched (its last range)" : "a range of a side of the pair kernel launched");
            if (r.last) sides_done |= mask;
        }
    }
    REO_HIP_CHECK(hipStreamSynchronize(c->up));   // the whole matrix has been read: the caller may have its array back
    stamp("upload stream idle");
    // the flags are final if they were read behind the last ranking (the boundary of the last group: the usual case) -- reading
    // them again would wait for the second side's pair kernel, which the host has no need to wait for here
    if (ranked_at_read != S && !bad_values && !fallback && (rc = read_flags())) return rc;
    // whatever follows on the context's stream (reo_build_pairs, the fall-back transform) is ordered behind the ranking stream
    REO_HIP_CHECK(hipEventRecord(c->ev_rk[1], rk));
    REO_HIP_CHECK(hipStreamWaitEvent(st, c->ev_rk[1], 0));
    c->transformed = false;
    if (bad_values) {
        set_error(kNaNMessage);
        return REO_EINVAL;
    }
    if (fallback) return REO_OK;   // some sample wants another form of the ranking: run_transform on the resident copy (reo_build_pairs)
    c->has_ties = fl[1];
    c->transform_in_lds = wide ? 2 : 1;
    c->transformed = true;
    if (with_k1 && sides_done == 3) c->eager_k1 = true;   // reo_build_pairs(0) finds its table made (or being made, on this stream)
    return REO_OK;
}

}  // namespace

void host_parallel(int nthreads, int ntasks, const std::function<void(int)> &fn) { HostPool::get(nthreads).run(ntasks, fn); }

int32_t ensure_upload_streams(reo_ctx *c)
{
    if (c->up) return REO_OK;
    REO_HIP_CHECK(handle_stream(&c->up, 0));
    REO_HIP_CHECK(handle_stream(&c->rk, 1));   // high priority
    for (auto &e : c->ev_up) REO_HIP_CHECK(handle_event(&e, 0));
    for (auto &e : c->ev_rk) REO_HIP_CHECK(handle_event(&e, 0));
    return REO_OK;
}

int32_t ensure_staging(reo_ctx *c, size_t slot_bytes)
{
    int32_t rc;
    if (c->stage_cap < slot_bytes) {
        // all three pinned slots are renewed together, and the recorded capacity is 0 until all three exist: a failed allocation leaves
        // no slot behind that a later, smaller request would take for a usable one (and reo_destroy releases every slot with the size
        // the block cache booked for it)
        const size_t old_cap = c->stage_cap;
        c->stage_cap = 0;
        for (int q = 0; q < 3; ++q)
            if (c->stage_h[q]) { pool_free(c->stage_h[q], old_cap, true); c->stage_h[q] = nullptr; }
        {
            ScopedNodeAffinity on_node(c->device);   // (the pinned pages come from the node of the thread that asks for them)
            for (int q = 0; q < 3; ++q) REO_HIP_CHECK(pool_alloc(reinterpret_cast<void **>(&c->stage_h[q]), slot_bytes, true));
        }
        c->stage_cap = slot_bytes;
    }
    for (int q = 0; q < 3; ++q) {
        if (!c->stage_h[q]) { set_error("a staging slot of the narrowed upload is missing"); return REO_EHIP; }
        if ((rc = c->stage_d[q].ensure(slot_bytes))) return rc;
        if (!c->ev_stage[q]) REO_HIP_CHECK(handle_event(&c->ev_stage[q], 0));
        if (!c->ev_widen[q]) REO_HIP_CHECK(handle_event(&c->ev_widen[q], 0));
    }
    return REO_OK;
}

// A whole host matrix (G x ncols, leading dimension hld) into a device matrix of leading dimension G, in chunks on the upload stream
// (Int64 narrowed: ChunkUploader); the context's stream is ordered behind the last chunk, and the call returns when the host array has
// been read.  dtype: 1 Float64, 2 Int64.
int32_t upload_columns(reo_ctx *c, const void *hX, int64_t hld, int64_t G, int64_t ncols, void *dX, int dtype)
{
    auto go = [&](auto *host, auto *dev) -> int32_t {
        using T = std::remove_pointer_t<decltype(dev)>;
        const int CH = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(ncols, std::max<int64_t>(16, (int64_t(32) << 20) / (G * 8)))));   // about 32 MB of source per chunk
        ChunkUploader<T> upl;
        int32_t rc = upl.init(c, host, hld, G, dev, CH);
        if (rc) return rc;
        hipEvent_t last = nullptr;
        for (int64_t c0 = 0; c0 < ncols; c0 += CH)
            if ((rc = upl.send(static_cast<int>(c0), static_cast<int>(std::min<int64_t>(CH, ncols - c0)), &last))) return rc;
        if (last) REO_HIP_CHECK(hipStreamWaitEvent(c->stream, last, 0));
        REO_HIP_CHECK(hipStreamSynchronize(c->up));
        return REO_OK;
    };
    return dtype == 1 ? go(static_cast<const double *>(hX), static_cast<double *>(dX)) : go(static_cast<const int64_t *>(hX), static_cast<int64_t *>(dX));
}

int32_t eager_upload(reo_ctx *c, const void *hX, int64_t hld, bool with_k1)
{
    return c->dtype == 1 ? eager_upload_impl<double>(c, static_cast<const double *>(hX), hld, with_k1)
                         : eager_upload_impl<int64_t>(c, static_cast<const int64_t *>(hX), hld, with_k1);
}

int32_t run_transform(reo_ctx *c)
{
    tic(c, 0);
    int32_t rc = c->dtype == 1 ? transform_impl<double>(c) : transform_impl<int64_t>(c);
    toc(c);
    return rc;
}

}  // namespace reo
