/*
 * reo_oracle.c -- CPU restatement of RankCompV3.jl's REO hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file, and only as the checker / reported baseline.
 *
 * Parity status: the reference is Julia 1.7 and no Julia toolchain exists in
 * this image, so the reference itself cannot be executed here.  This
 * restatement is pinned by the ONE known-answer test the reference holds
 * (McCullagh 1977 Table 1: input test/McCullagh_test.jl:39, expected 5-tuple
 * src/RankCompV3.jl:207-209, expected N/R :213-221) and cross-checked
 * against an independent numpy/scipy restatement (oracle/reo_numpy.py).
 * Everything else on the path (binomial thresholds, BH, trimmed std, the
 * full identify_degs output) is "parity unpinned" by the reference's own
 * tests -- see DESIGN.md.
 *
 * Every function cites the reference lines (relative to /root/reference) it
 * follows.  The loop nest is the reference's (pair-at-a-time, sample-at-a-
 * time literal comparator); only the storage differs (one byte code per
 * ordered pair instead of a 9-plane BitArray) which does not change results.
 *
 * Randomness: the reference resolves every tie with an unseeded rand(Bool)
 * (src/RankCompV3.jl:73).  Per pair and group that is n_gt + Binomial(n_eq,
 * 1/2).  The contract used here (and by the HIP path) draws that binomial
 * from a counter-based generator keyed by (seed, i, j, group): see
 * oracle_tie_wins().
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------------------------------------------------------------- RNG -- */

static inline uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* Number of tied samples (out of n_eq) of unordered pair (i<j), group g, that
 * is_greater() would have reported as "i greater" (src/RankCompV3.jl:72-73:
 * one fair coin per tied sample).  One bit of the keyed stream per tie. */
uint32_t oracle_tie_wins(uint64_t seed, uint32_t i, uint32_t j, uint32_t g, uint32_t n_eq)
{
    uint64_t base = mix64(seed ^ mix64(((uint64_t)i << 32) | (uint64_t)j)) + ((uint64_t)g << 40);
    uint32_t wins = 0;
    for (uint64_t w = 0; n_eq > 0; ++w) {
        uint64_t bits = mix64(base + w);
        uint32_t take = n_eq < 64 ? n_eq : 64;
        if (take < 64) bits &= (((uint64_t)1) << take) - 1;
        wins += (uint32_t)__builtin_popcountll(bits);
        n_eq -= take;
    }
    return wins;
}

/* ---------------------------------------------------------- thresholds -- */

/* cdf of Binomial(n, 1/2) at x, exact summation in long double. */
static long double binom_half_cdf(int n, int x)
{
    if (x < 0) return 0.0L;
    if (x >= n) return 1.0L;
    /* sum_{t<=x} C(n,t) 2^-n, built from logs to survive n in the thousands */
    long double acc = 0.0L;
    long double logc = 0.0L; /* log C(n,0) */
    const long double ln2 = 0.693147180559945309417232121458L;
    for (int t = 0; t <= x; ++t) {
        if (t > 0) logc += logl((long double)(n - t + 1)) - logl((long double)t);
        acc += expl(logc - (long double)n * ln2);
    }
    return acc > 1.0L ? 1.0L : acc;
}

/* HypothesisTests.pvalue(Binomial(n), x; tail=:both) for a discrete
 * distribution = min(1, 2*min(ccdf(x-1), cdf(x)))  (call sites
 * src/RankCompV3.jl:83,85). */
static double binom_two_sided(int n, int x)
{
    long double lo = binom_half_cdf(n, x);
    long double hi = 1.0L - binom_half_cdf(n, x - 1);
    long double p = 2.0L * (lo < hi ? lo : hi);
    return (double)(p > 1.0L ? 1.0L : p);
}

/* get_major_reo_lower_count, src/RankCompV3.jl:81-92.  Returns n itself on
 * the WARN branch (:87-90). */
int32_t oracle_threshold(int32_t n, double pval_reo)
{
    double pmin = binom_two_sided(n, 0);
    if (pmin < pval_reo) {
        /* findfirst(p .> thr) over x = 0..n/2 (:85-86).  The two-sided p is non-decreasing in x on that range, so the
         * first x above the threshold is found by bisection (each probe is still the exact cdf sum): O(n log n) instead
         * of O(n^2), which matters for the 66 000-sample test. */
        if (!(binom_two_sided(n, n / 2) > pval_reo)) return -1; /* findfirst -> nothing: the reference would throw */
        int lo = 0, hi = n / 2; /* p(lo) <= thr < p(hi) */
        while (hi - lo > 1) {
            const int mid = lo + (hi - lo) / 2;
            if (binom_two_sided(n, mid) > pval_reo) hi = mid; else lo = mid;
        }
        return n - hi + 1; /* -idx+2+n, idx=x+1 */
    }
    return n;
}

/* ---------------------------------------------------------- comparator -- */

/* is_greater without the coin: 0 = not greater, 1 = greater, 2 = tie
 * (src/RankCompV3.jl:71-77). */
static inline int cmp3(double x, double y)
{
    if (fabs(x - y) < 0.1) return 2;
    return x > y ? 1 : 0;
}

/* Deterministic per-pair per-group counts over a block of pairs.
 * X is G x S column-major with leading dimension ld (the layout Julia hands
 * over, src/RankCompV3.jl:652).  group_id[s] in [0,ngroups) in order of first
 * appearance (unique(), :353).  Outputs are [(i-i0)][(j-j0)][g]. */
void oracle_pair_counts(const double *X, int64_t G, int64_t S, int64_t ld,
                        const int32_t *group_id, int32_t ngroups,
                        int64_t i0, int64_t i1, int64_t j0, int64_t j1,
                        uint16_t *n_gt, uint16_t *n_eq)
{
    (void)G;
    int64_t nj = j1 - j0;
    for (int64_t i = i0; i < i1; ++i)
        for (int64_t j = j0; j < j1; ++j) {
            uint16_t *gt = n_gt + ((i - i0) * nj + (j - j0)) * ngroups;
            uint16_t *eq = n_eq + ((i - i0) * nj + (j - j0)) * ngroups;
            for (int g = 0; g < ngroups; ++g) gt[g] = eq[g] = 0;
            for (int64_t s = 0; s < S; ++s) {
                int c = cmp3(X[i + s * ld], X[j + s * ld]); /* :372 */
                if (c == 1) gt[group_id[s]]++;
                else if (c == 2) eq[group_id[s]]++;
            }
        }
}

/* Classification of one group-vs-rest comparison, src/RankCompV3.jl:376-377
 * (same as sum_reo :120-121).  Returns 3*(ic-1)+(it-1) in 0..8. */
static inline int classify(int nre, int not_, int s1, int s2, int m1, int m2)
{
    int ic = nre >= m1 ? 3 : ((s1 - nre) >= m1 ? 1 : 2);
    int it = not_ >= m2 ? 3 : ((s2 - not_) >= m2 ? 1 : 2);
    return 3 * (ic - 1) + (it - 1);
}

/* REO table build, src/RankCompV3.jl:363-392, for comparison k (0-based;
 * ctrl = group k, treat = every other sample).  code is G x G row-major, one
 * byte per ORDERED pair: code[i*G+j] in 0..8, 0xFF on the diagonal (the
 * reference never sets a diagonal bit).  Mirror rule :386: code(j,i) =
 * 8 - code(i,j).  thr = {m1, m2} = threshold[1,k], threshold[2,k] (:362). */
void oracle_build_codes(const double *X, int64_t G, int64_t S, int64_t ld,
                        const int32_t *group_id, int32_t ngroups, int32_t k,
                        const int32_t *thr, uint64_t seed, uint8_t *code)
{
    int32_t *gs = (int32_t *)calloc((size_t)ngroups, sizeof(int32_t));  /* any number of levels (:353) */
    for (int64_t s = 0; s < S; ++s) gs[group_id[s]]++;
    int s1 = gs[k], s2 = (int)S - gs[k];
    free(gs);
#pragma omp parallel
    {
    int *gt = (int *)malloc(sizeof(int) * (size_t)ngroups), *eq = (int *)malloc(sizeof(int) * (size_t)ngroups);
#pragma omp for schedule(dynamic, 8)
    for (int64_t i = 0; i < G; ++i) {
        code[i * G + i] = 0xFF;
        for (int64_t j = i + 1; j < G; ++j) {
            for (int g = 0; g < ngroups; ++g) gt[g] = eq[g] = 0;
            for (int64_t s = 0; s < S; ++s) {
                int c = cmp3(X[i + s * ld], X[j + s * ld]);
                if (c == 1) gt[group_id[s]]++;
                else if (c == 2) eq[group_id[s]]++;
            }
            int tot = 0, nre_k = 0;
            for (int g = 0; g < ngroups; ++g) {
                int nre = gt[g] + (eq[g] ? (int)oracle_tie_wins(seed, (uint32_t)i, (uint32_t)j, (uint32_t)g, (uint32_t)eq[g]) : 0);
                tot += nre; /* :373 */
                if (g == k) nre_k = nre;
            }
            int c = classify(nre_k, tot - nre_k /* :374 */, s1, s2, thr[0], thr[1]);
            code[i * G + j] = (uint8_t)c;       /* :385 */
            code[j * G + i] = (uint8_t)(8 - c); /* :386 */
        }
    }
    free(gt); free(eq);
    }
}

/* Per-gene 3x3 contingency builder, src/RankCompV3.jl:403: cont[i][c] =
 * #{ j : ref[j] and code(i,j) == c }.  cont is G x 9 row-major. */
void oracle_tally(const uint8_t *code, int64_t G, const uint8_t *ref, int32_t *cont)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < G; ++i) {
        int32_t *c = cont + i * 9;
        for (int t = 0; t < 9; ++t) c[t] = 0;
        for (int64_t j = 0; j < G; ++j)
            if (ref[j] && code[i * G + j] < 9) c[code[i * G + j]]++;
    }
}

/* ------------------------------------------------------- McCullagh test -- */

static double norm_two_sided(double x, double sigma)
{
    /* pvalue(Normal(0,sigma), x; tail=:both) = min(1, 2 min(cdf, ccdf));
     * cdf(z) = erfc(-z/sqrt2)/2  (call sites src/RankCompV3.jl:255,412). */
    /* sigma == 0 (all trimmed delta1 equal): StatsFuns 1.1.1 normcdf/normccdf
     * special-case x == mu to z = +Inf, otherwise z = +-Inf: the smaller tail is 0. */
    double z = (sigma == 0.0 && x == 0.0) ? INFINITY : x / sigma;
    double lo = 0.5 * erfc(-z * M_SQRT1_2), hi = 0.5 * erfc(z * M_SQRT1_2);
    double p = 2.0 * (lo < hi ? lo : hi);
    return p > 1.0 ? 1.0 : p;
}

/* General k x k McCullagh test, src/RankCompV3.jl:225-259.  mat row-major.
 * out = {pval, d1, d2, se, z1}; Nout (k-1)^2 and Rout (k-1) optional.
 * Returns 1 on the singular branch (:242-243). */
int oracle_mccullagh(int32_t k, const int64_t *mat, double *out, int64_t *Nout, int64_t *Rout)
{
    enum { KM = 16 };
    int m = k - 1;
    double N[KM][KM], A[KM][2 * KM], n[KM], R[KM], w2[KM];
    if (k < 2 || k > KM) return -1;
    for (int i = 0; i < m; ++i)
        for (int j = i; j < m; ++j) { /* :230-234 */
            int64_t s = 0;
            for (int a = 0; a <= i; ++a)
                for (int b = j + 1; b < k; ++b) s += mat[a * k + b] + mat[b * k + a];
            N[i][j] = N[j][i] = (double)s;
            if (Nout) Nout[i * m + j] = Nout[j * m + i] = s;
        }
    for (int i = 0; i < m; ++i) { /* :237-240 */
        int64_t s = 0;
        for (int a = 0; a <= i; ++a)
            for (int b = i + 1; b < k; ++b) s += mat[a * k + b];
        R[i] = (double)s;
        n[i] = N[i][i];
        if (Rout) Rout[i] = s;
    }
    /* Gauss-Jordan with partial pivoting: det and inverse together */
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) { A[i][j] = N[i][j]; A[i][m + j] = (i == j); }
    double det = 1.0;
    for (int c = 0; c < m; ++c) {
        int p = c;
        for (int r = c + 1; r < m; ++r) if (fabs(A[r][c]) > fabs(A[p][c])) p = r;
        if (A[p][c] == 0.0) { det = 0.0; break; }
        if (p != c) { for (int j = 0; j < 2 * m; ++j) { double t = A[p][j]; A[p][j] = A[c][j]; A[c][j] = t; } det = -det; }
        det *= A[c][c];
        double inv = 1.0 / A[c][c];
        for (int j = 0; j < 2 * m; ++j) A[c][j] *= inv;
        for (int r = 0; r < m; ++r) if (r != c) {
            double f = A[r][c];
            if (f != 0.0) for (int j = 0; j < 2 * m; ++j) A[r][j] -= f * A[c][j];
        }
    }
    if (fabs(det) <= 2.220446049250313e-16) { /* :242-243 */
        out[0] = 1.0; out[1] = out[2] = out[3] = out[4] = 0.0;
        return 1;
    }
    double nw = 0.0;
    for (int i = 0; i < m; ++i) { /* w2 = inv(N) n, :246 */
        double s = 0.0;
        for (int j = 0; j < m; ++j) s += A[i][m + j] * n[j];
        w2[i] = s; nw += n[i] * s;
    }
    double nu = 1.0 / nw; /* :247 */
    double d1 = 0.0, a = 0.0, b = 0.0;
    for (int i = 0; i < m; ++i) {
        d1 += (n[i] * w2[i] * nu) * log((R[i] + 0.5) / (n[i] - R[i] + 0.5)); /* :248-249 */
        a += w2[i] * R[i];
        b += w2[i] * (n[i] - R[i]);
    }
    double d2 = log((0.5 + a) / (0.5 + b));    /* :250 */
    double v1 = 4.0 * (1.0 + 0.25 * d1 * d1) * nu; /* :251 */
    double v2 = 4.0 * (1.0 + 0.25 * d2 * d2) * nu; /* :252 */
    double se = sqrt((v1 + v2) * 0.5);          /* :253 */
    double z1 = d1 / se;                        /* :254 */
    out[0] = norm_two_sided(z1, 1.0);           /* :255 */
    out[1] = d1; out[2] = d2; out[3] = se; out[4] = z1;
    return 0;
}

/* 3x3 table straight from the 9 tallies n11..n33 (row = ctrl state, the
 * reshape(cont,3,3)' of src/RankCompV3.jl:404). */
int oracle_mccullagh9(const int32_t *cont, double *out)
{
    int64_t mat[9];
    for (int t = 0; t < 9; ++t) mat[t] = cont[t];
    return oracle_mccullagh(3, mat, out, NULL, NULL);
}

/* ------------------------------------------------------ iteration stats -- */

static int cmp_double(const void *a, const void *b)
{
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

/* Julia round(Int, x): round-half-even in the default FP environment. */
static int64_t jl_round(double x) { return (int64_t)nearbyint(x); }

/* std(sort(d)[round(Int,G*0.05) : round(Int,G*0.95)]), src/RankCompV3.jl:
 * 409-411 (1-based inclusive slice, n-1 estimator).  Returns NaN with *err=1
 * when the slice would be out of bounds (reference: BoundsError). */
double oracle_trimmed_std(const double *d, int64_t G, int32_t *err)
{
    int64_t a = jl_round((double)G * 0.05), b = jl_round((double)G * 0.95);
    *err = 0;
    if (a < 1 || b > G) { *err = 1; return NAN; }
    if (b < a) return NAN; /* std of an empty slice */
    double *t = (double *)malloc(sizeof(double) * (size_t)G);
    memcpy(t, d, sizeof(double) * (size_t)G);
    qsort(t, (size_t)G, sizeof(double), cmp_double);
    int64_t n = b - a + 1;
    long double sum = 0.0L;
    for (int64_t i = a - 1; i < b; ++i) sum += t[i];
    long double mean = sum / (long double)n, ss = 0.0L;
    for (int64_t i = a - 1; i < b; ++i) { long double e = t[i] - mean; ss += e * e; }
    free(t);
    return (double)sqrtl(ss / (long double)(n - 1));
}

typedef struct { double p; int64_t i; } pidx_t;
static int cmp_pidx(const void *a, const void *b)
{
    const pidx_t *x = (const pidx_t *)a, *y = (const pidx_t *)b;
    if (x->p < y->p) return -1;
    if (x->p > y->p) return 1;
    return (x->i > y->i) - (x->i < y->i);
}

/* MultipleTesting.adjust(p, BenjaminiHochberg()) (0.5.1; call site
 * src/RankCompV3.jl:413): sort ascending, p_(r) * (n / r), reverse
 * cumulative minimum, clamp to 1, unsort. */
void oracle_bh(const double *p, int64_t n, double *padj)
{
    if (n <= 1) { for (int64_t i = 0; i < n; ++i) padj[i] = p[i]; return; }
    pidx_t *v = (pidx_t *)malloc(sizeof(pidx_t) * (size_t)n);
    for (int64_t i = 0; i < n; ++i) { v[i].p = p[i]; v[i].i = i; }
    qsort(v, (size_t)n, sizeof(pidx_t), cmp_pidx);
    double run = v[n - 1].p * ((double)n / (double)n);
    padj[v[n - 1].i] = run < 1.0 ? run : 1.0;
    for (int64_t r = n - 1; r >= 1; --r) { /* rank r (1-based) */
        double a = v[r - 1].p * ((double)n / (double)r);
        if (a < run) run = a;
        padj[v[r - 1].i] = run < 1.0 ? run : 1.0;
    }
    free(v);
}

/* One pass of the iteration body, src/RankCompV3.jl:402-417, given the
 * tallies.  result is G x 15 column-major: [pval padj n11..n33 d1 d2 se z1]
 * (:398,405,415-416).  inds[i] = 1 for non-DEG (:417).  Returns the number of
 * non-DEGs, or -1 on the out-of-bounds slice. */
int64_t oracle_iter_stats(const int32_t *cont, int64_t G, double pval_deg, double padj_deg,
                          double *result, uint8_t *inds)
{
    double *d1 = result + 11 * G;
    for (int64_t i = 0; i < G; ++i) {
        double o[5];
        oracle_mccullagh9(cont + i * 9, o);
        result[i] = o[0];
        result[G + i] = 1.0;
        for (int t = 0; t < 9; ++t) result[(2 + t) * G + i] = (double)cont[i * 9 + t];
        result[11 * G + i] = o[1]; result[12 * G + i] = o[2];
        result[13 * G + i] = o[3]; result[14 * G + i] = o[4];
    }
    int32_t err;
    double se = oracle_trimmed_std(d1, G, &err);
    if (err) return -1;
    for (int64_t i = 0; i < G; ++i) result[i] = norm_two_sided(d1[i], se); /* :412,415 */
    oracle_bh(result, G, result + G);                                      /* :413,416 */
    int64_t nn = 0;
    for (int64_t i = 0; i < G; ++i) {
        inds[i] = !(result[i] <= pval_deg && result[G + i] <= padj_deg);
        nn += inds[i];
    }
    return nn;
}

/* Iteration driver, src/RankCompV3.jl:396-425, for one comparison whose code
 * table is given.  trace is n_iter x 2 (#DEG, #non-DEG per executed pass,
 * the :418 log line).  Returns 0, or -1 on the reference's error path. */
int32_t oracle_iterate(const uint8_t *code, int64_t G, const uint8_t *ref0,
                       double pval_deg, double padj_deg, int32_t n_iter, int32_t n_conv,
                       double *result, int32_t *iters_run, int32_t *trace)
{
    uint8_t *ref = (uint8_t *)malloc((size_t)G), *inds = (uint8_t *)malloc((size_t)G);
    int32_t *cont = (int32_t *)malloc(sizeof(int32_t) * 9 * (size_t)G);
    memcpy(ref, ref0, (size_t)G);
    memset(result, 0, sizeof(double) * 15 * (size_t)G); /* :398 */
    int32_t i_iter = 0, passes = 0, rc = 0;
    while (i_iter < n_iter) { /* :400 */
        oracle_tally(code, G, ref, cont);
        int64_t nn = oracle_iter_stats(cont, G, pval_deg, padj_deg, result, inds);
        if (nn < 0) { rc = -1; break; }
        int64_t nref = 0;
        for (int64_t i = 0; i < G; ++i) nref += ref[i];
        if (trace) { trace[2 * passes] = (int32_t)(G - nn); trace[2 * passes + 1] = (int32_t)nn; }
        ++passes;
        if (llabs(nref - nn) < n_conv) break; /* :419-422 */
        ++i_iter;                              /* :423 */
        memcpy(ref, inds, (size_t)G);          /* :424 */
    }
    if (iters_run) *iters_run = passes;
    free(ref); free(inds); free(cont);
    return rc;
}

/* identify_degs for comparison k, src/RankCompV3.jl:339-438 minus the string
 * labelling (:426-430, done by the caller from z1/pval/padj). */
int32_t oracle_identify_degs(const double *X, int64_t G, int64_t S, int64_t ld,
                             const int32_t *group_id, int32_t ngroups, int32_t k,
                             double pval_reo, double pval_deg, double padj_deg,
                             const uint8_t *ref0, int32_t n_iter, int32_t n_conv, uint64_t seed,
                             double *result, int32_t *iters_run, int32_t *trace)
{
    int32_t gs = 0;
    for (int64_t s = 0; s < S; ++s) gs += (group_id[s] == k);
    int32_t thr[2] = { oracle_threshold(gs, pval_reo), oracle_threshold((int32_t)S - gs, pval_reo) }; /* :362 */
    if (thr[0] < 0 || thr[1] < 0) return -1;
    uint8_t *code = (uint8_t *)malloc((size_t)G * (size_t)G);
    if (!code) return -4;
    oracle_build_codes(X, G, S, ld, group_id, ngroups, k, thr, seed, code);
    int32_t rc = oracle_iterate(code, G, ref0, pval_deg, padj_deg, n_iter, n_conv, result, iters_run, trace);
    free(code);
    return rc;
}

int32_t oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
