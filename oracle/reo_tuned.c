/*
 * reo_tuned.c -- "R2" of SURVEY.md section 8d: a TUNED CPU implementation of the same hot path, for the reported
 * CPU baseline only (bench.py cpu_baseline.tuned).  TEST INFRASTRUCTURE like everything under oracle/: the product
 * never loads it.
 *
 * Same results as reo_oracle.c (the reference-faithful restatement "R1"; tests/test_oracle.py compares the two),
 * different organisation -- what a CPU implementer would do after profiling the reference's loop nest
 * (/root/reference/src/RankCompV3.jl:366-392,402-407):
 *   - per sample, sort once and replace values by 16-bit positions and tie-band edges (tie <=> |x - y| < 0.1, :72),
 *     so that the comparator is an integer compare;
 *   - gene-major rows (one pair's samples are contiguous), SIMD compares over 32 samples at a time
 *     (gcc vector extensions; target_clones picks AVX2 or the baseline at load time);
 *   - the class table as four bit planes per gene row (like R's bit planes, one bit per ordered pair), upper
 *     triangle from the pair loop, lower triangle by a blocked bit transpose (mirror rule :386);
 *   - tallies by AND + popcount against the reference-set mask (:403);
 *   - the per-pass statistics of reo_oracle.c with the per-gene part in parallel.
 * Two groups (comparison 0 = group 0 vs group 1), like BASELINE configs 1-5.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

uint32_t oracle_tie_wins(uint64_t seed, uint32_t i, uint32_t j, uint32_t g, uint32_t n_eq);
int32_t oracle_threshold(int32_t n, double pval_reo);
int oracle_mccullagh9(const int32_t *cont, double *out);
double oracle_trimmed_std(const double *d, int64_t G, int32_t *err);
void oracle_bh(const double *p, int64_t n, double *padj);

typedef int16_t v16i16 __attribute__((vector_size(32)));  /* positions are stored biased by 0x8000: signed SIMD compares exist everywhere */

typedef struct { double v; int32_t g; } keyed;
static int cmp_keyed(const void *a, const void *b)
{
    const keyed *x = (const keyed *)a, *y = (const keyed *)b;
    return x->v < y->v ? -1 : (x->v > y->v ? 1 : (x->g - y->g));
}

/* #{s : pos[s] < edge[s]} over n (multiple of 32) samples.  (gcc 11 scalarises 64-byte generic vectors; 32-byte ones
 * map to one AVX2 compare each.) */
__attribute__((target_clones("avx2", "default")))
static int count_below(const uint16_t *pos, const uint16_t *edge, int n)
{
    v16i16 acc0 = {0}, acc1 = {0};
    for (int s = 0; s < n; s += 32) {
        v16i16 p0, e0, p1, e1;
        memcpy(&p0, pos + s, 32); memcpy(&e0, edge + s, 32);
        memcpy(&p1, pos + s + 16, 32); memcpy(&e1, edge + s + 16, 32);
        acc0 -= (p0 < e0);  /* true lanes are -1 */
        acc1 -= (p1 < e1);
    }
    acc0 += acc1;
    int t = 0;
    for (int l = 0; l < 16; ++l) t += acc0[l];
    return t;
}

static inline int classify3(int nre, int size, int m) { return nre >= m ? 2 : (size - nre >= m ? 0 : 1); }

/* transpose a 64 x 64 bit block (rows = 64 words) in place */
static void transpose64(uint64_t a[64])
{
    uint64_t m = 0x00000000FFFFFFFFULL;
    for (int j = 32; j != 0; j >>= 1, m ^= (m << j))
        for (int k = 0; k < 64; k = (k + j + 1) & ~j) {
            const uint64_t t = ((a[k] >> j) ^ a[k + j]) & m;
            a[k] ^= t << j;
            a[k + j] ^= t;
        }
}

/* The class table of comparison 0 as bit planes: T[G][cL cH tL tH][W] 64-bit words, W = ceil(G / 64) (caller-allocated,
 * G * 4 * W words); :363-392. */
int32_t tuned_build_table(const double *X, int64_t G, int64_t S, int64_t ld, const int32_t *group_id, int32_t ngroups,
                          double pval_reo, uint64_t seed, uint64_t *T)
{
    if (ngroups != 2 || G > 65535) return -2;
    int32_t n0 = 0;
    for (int64_t s = 0; s < S; ++s) n0 += group_id[s] == 0;
    const int32_t n1 = (int32_t)S - n0;
    const int32_t m1 = oracle_threshold(n0, pval_reo), m2 = oracle_threshold(n1, pval_reo); /* :362 */
    if (m1 < 0 || m2 < 0) return -1;
    const int p0 = (n0 + 31) / 32 * 32, p1 = (n1 + 31) / 32 * 32, SP = p0 + p1;  /* ctrl slots, then treat slots */
    uint16_t *POS = (uint16_t *)aligned_alloc(64, (size_t)G * SP * 2), *LO = (uint16_t *)aligned_alloc(64, (size_t)G * SP * 2),
             *HI = (uint16_t *)aligned_alloc(64, (size_t)G * SP * 2);
    const int64_t W = (G + 63) / 64;
    if (!POS || !LO || !HI || !T) return -4;
    memset(T, 0, (size_t)G * 4 * W * 8);
    for (size_t x = 0; x < (size_t)G * SP; ++x) { POS[x] = 0x7FFF; LO[x] = 0x8000; HI[x] = 0x8000; }  /* padding slots (biased): the largest position, the smallest edges */
    int any_tie = 0;
    const int dbg = getenv("REO_TUNED_DEBUG") != NULL;
    double t0 = omp_get_wtime();
#define MARK(what) do { if (dbg) { const double t1 = omp_get_wtime(); fprintf(stderr, "tuned: %-12s %.3f s\n", what, t1 - t0); t0 = t1; } } while (0)
    /* ---- per-sample rank / tie-band transform (is_greater, :71-77) */
    {
        int *slot = (int *)malloc(sizeof(int) * S);
        int c0 = 0, c1 = 0;
        for (int64_t s = 0; s < S; ++s) slot[s] = group_id[s] == 0 ? c0++ : p0 + c1++;
#pragma omp parallel reduction(| : any_tie)
        {
            keyed *k = (keyed *)malloc(sizeof(keyed) * G);
#pragma omp for schedule(dynamic, 1)
            for (int64_t s = 0; s < S; ++s) {
                for (int64_t g = 0; g < G; ++g) { k[g].v = X[g + s * ld]; k[g].g = (int32_t)g; }
                qsort(k, (size_t)G, sizeof(keyed), cmp_keyed);
                int64_t l = 0, h = 0;
                for (int64_t p = 0; p < G; ++p) {
                    while (l < p && !(fabs(k[l].v - k[p].v) < 0.1)) ++l;    /* first position of the band (an infinity is tied with nothing, itself included: abs(Inf - Inf) = NaN, :72 -- equal infinities keep the sort's gene order) */
                    if (h < p) h = p;
                    while (h + 1 < G && fabs(k[h + 1].v - k[p].v) < 0.1) ++h;  /* last position of the band */
                    const size_t o = (size_t)k[p].g * SP + slot[s];
                    POS[o] = (uint16_t)p ^ 0x8000; LO[o] = (uint16_t)l ^ 0x8000; HI[o] = (uint16_t)(h + 1) ^ 0x8000;
                    any_tie |= (l != p) | (h != p);
                }
            }
            free(k);
        }
        free(slot);
    }
    MARK("transform");
    /* ---- pair loop (:366-392): upper triangle, forward bits of row i */
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t i = 0; i < G; ++i) {
        const uint16_t *lo = LO + (size_t)i * SP, *hi = HI + (size_t)i * SP;
        uint64_t *row = T + (size_t)i * 4 * W;
        for (int64_t j = i + 1; j < G; ++j) {
            const uint16_t *pj = POS + (size_t)j * SP;
            int gc = count_below(pj, lo, p0), gt = count_below(pj + p0, lo + p0, p1);
            if (any_tie) {
                const int ec = count_below(pj, hi, p0) - gc, et = count_below(pj + p0, hi + p0, p1) - gt;
                if (ec) gc += (int)oracle_tie_wins(seed, (uint32_t)i, (uint32_t)j, 0, (uint32_t)ec);
                if (et) gt += (int)oracle_tie_wins(seed, (uint32_t)i, (uint32_t)j, 1, (uint32_t)et);
            }
            const int ic = classify3(gc, n0, m1), it = classify3(gt, n1, m2);  /* :376-377 */
            const uint64_t bit = 1ULL << (j & 63);
            if (ic == 0) row[0 * W + (j >> 6)] |= bit; else if (ic == 2) row[1 * W + (j >> 6)] |= bit;
            if (it == 0) row[2 * W + (j >> 6)] |= bit; else if (it == 2) row[3 * W + (j >> 6)] |= bit;
        }
    }
    MARK("pairs");
    /* ---- mirror rule (:386): class(j,i) = 8 - class(i,j): lower triangle = transpose with L and H swapped */
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t bi = 0; bi < W; ++bi)
        for (int64_t bj = bi; bj < W; ++bj)
            for (int pl = 0; pl < 4; ++pl) {
                uint64_t blk[64];
                for (int r = 0; r < 64; ++r) { const int64_t i = bi * 64 + r; blk[r] = i < G ? T[((size_t)i * 4 + pl) * W + bj] : 0; }
                if (bi == bj) for (int r = 0; r < 64; ++r) blk[r] &= r < 63 ? ~0ULL << (r + 1) : 0;  /* strictly above the diagonal */
                transpose64(blk);
                const int sw = pl ^ 1;  /* L <-> H */
                for (int r = 0; r < 64; ++r) { const int64_t j = bj * 64 + r; if (j < G && blk[r]) T[((size_t)j * 4 + sw) * W + bi] |= blk[r]; }
            }
    free(POS); free(LO); free(HI);
    MARK("mirror");
    return 0;
}

/* The iteration driver (:396-425) on a table made by tuned_build_table. */
int32_t tuned_iterate(const uint64_t *T, int64_t G, double pval_deg, double padj_deg, const uint8_t *ref0, int32_t n_iter,
                      int32_t n_conv, double *result, int32_t *iters_run, int32_t *trace)
{
    const int64_t W = (G + 63) / 64;
    const int dbg = getenv("REO_TUNED_DEBUG") != NULL;
    double t0 = omp_get_wtime();
    /* ---- iteration driver (:396-425) */
    uint8_t *ref = (uint8_t *)malloc((size_t)G), *inds = (uint8_t *)malloc((size_t)G);
    uint64_t *mask = (uint64_t *)malloc(sizeof(uint64_t) * W);
    int32_t *cont = (int32_t *)malloc(sizeof(int32_t) * 9 * (size_t)G);
    memcpy(ref, ref0, (size_t)G);
    memset(result, 0, sizeof(double) * 15 * (size_t)G);
    int32_t i_iter = 0, passes = 0, rc = 0;
    double *d1 = result + 11 * G;
    while (i_iter < n_iter) {
        int64_t nref = 0;
        memset(mask, 0, sizeof(uint64_t) * W);
        for (int64_t g = 0; g < G; ++g) if (ref[g]) { mask[g >> 6] |= 1ULL << (g & 63); ++nref; }
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < G; ++i) {  /* tallies (:403) + McCullagh (:404-405) */
            const uint64_t *r = T + (size_t)i * 4 * W;
            int cl = 0, ch = 0, tl = 0, th = 0, ll = 0, lh = 0, hl = 0, hh = 0;
            for (int64_t w = 0; w < W; ++w) {
                const uint64_t m = mask[w], a = r[w] & m, b = r[W + w] & m, c = r[2 * W + w] & m, d = r[3 * W + w] & m;
                cl += __builtin_popcountll(a); ch += __builtin_popcountll(b); tl += __builtin_popcountll(c); th += __builtin_popcountll(d);
                ll += __builtin_popcountll(a & c); lh += __builtin_popcountll(a & d); hl += __builtin_popcountll(b & c); hh += __builtin_popcountll(b & d);
            }
            int32_t *c9 = cont + i * 9;
            c9[0] = ll; c9[2] = lh; c9[6] = hl; c9[8] = hh;
            c9[1] = cl - ll - lh; c9[7] = ch - hl - hh; c9[3] = tl - ll - hl; c9[5] = th - lh - hh;
            c9[4] = (int32_t)(nref - (ref[i] ? 1 : 0)) - (cl + ch + c9[3] + c9[5]);
            double o[5];
            oracle_mccullagh9(c9, o);
            result[i] = o[0]; result[G + i] = 1.0;
            for (int t = 0; t < 9; ++t) result[(2 + t) * G + i] = (double)c9[t];
            result[11 * G + i] = o[1]; result[12 * G + i] = o[2]; result[13 * G + i] = o[3]; result[14 * G + i] = o[4];
        }
        int32_t err;
        const double se = oracle_trimmed_std(d1, G, &err);  /* :409-411 */
        if (err) { rc = -1; break; }
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < G; ++i) {  /* :412, the expression of reo_oracle.c's norm_two_sided */
            const double x = d1[i], z = (se == 0.0 && x == 0.0) ? INFINITY : x / se;
            const double lo = 0.5 * erfc(-z * M_SQRT1_2), hi = 0.5 * erfc(z * M_SQRT1_2);
            const double p = 2.0 * (lo < hi ? lo : hi);
            result[i] = p > 1.0 ? 1.0 : p;
        }
        oracle_bh(result, G, result + G);  /* :413 */
        int64_t nn = 0;
        for (int64_t i = 0; i < G; ++i) { inds[i] = !(result[i] <= pval_deg && result[G + i] <= padj_deg); nn += inds[i]; }
        if (trace) { trace[2 * passes] = (int32_t)(G - nn); trace[2 * passes + 1] = (int32_t)nn; }
        ++passes;
        if (llabs(nref - nn) < n_conv) break;
        ++i_iter;
        memcpy(ref, inds, (size_t)G);
    }
    MARK("passes");
    if (iters_run) *iters_run = passes;
    free(ref); free(inds); free(mask); free(cont);
    return rc;
}

/* class codes 0..8 (255 on the diagonal) of the block [i0, i1) x [j0, j1) of a table, row-major (like k_decode) */
void tuned_decode(const uint64_t *T, int64_t G, int64_t i0, int64_t i1, int64_t j0, int64_t j1, uint8_t *code)
{
    const int64_t W = (G + 63) / 64;
#pragma omp parallel for schedule(static)
    for (int64_t i = i0; i < i1; ++i) {
        const uint64_t *r = T + (size_t)i * 4 * W;
        for (int64_t j = j0; j < j1; ++j) {
            const int64_t w = j >> 6; const int sh = (int)(j & 63);
            const int l = (int)((r[w] >> sh) & 1), h = (int)((r[W + w] >> sh) & 1), tl = (int)((r[2 * W + w] >> sh) & 1), th = (int)((r[3 * W + w] >> sh) & 1);
            const int ic = l ? 0 : (h ? 2 : 1), it = tl ? 0 : (th ? 2 : 1);
            code[(size_t)(i - i0) * (size_t)(j1 - j0) + (size_t)(j - j0)] = i == j ? 255 : (uint8_t)(3 * ic + it);
        }
    }
}

int32_t tuned_identify_degs(const double *X, int64_t G, int64_t S, int64_t ld, const int32_t *group_id, int32_t ngroups,
                            double pval_reo, double pval_deg, double padj_deg, const uint8_t *ref0, int32_t n_iter,
                            int32_t n_conv, uint64_t seed, double *result, int32_t *iters_run, int32_t *trace)
{
    if (ngroups != 2 || G > 65535) return -2;
    const int64_t W = (G + 63) / 64;
    uint64_t *T = (uint64_t *)malloc((size_t)G * 4 * W * 8);
    if (!T) return -4;
    int32_t rc = tuned_build_table(X, G, S, ld, group_id, ngroups, pval_reo, seed, T);
    if (!rc) rc = tuned_iterate(T, G, pval_deg, padj_deg, ref0, n_iter, n_conv, result, iters_run, trace);
    free(T);
    return rc;
}
