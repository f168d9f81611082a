/* A plain C client of include/reo_hip.h: the call sequence a Julia/C/Fortran host would make in place of
 * identify_degs (/root/reference/src/RankCompV3.jl:339-438), with no Python and no torch in the process.
 * Reads a small problem from stdin (text), prints the trace and the statistics; tests/test_gpu_parity.py
 * compiles it with gcc, runs it on the GPU box and compares its output with the ctypes path and the oracle.
 *
 * argv[1] (optional) selects how the context is made: "single" (default) reo_create; "rccl" reo_create + the
 * in-library RCCL path as rank 0 of a world of 1 (reo_comm_unique_id, reo_comm_init_rank: reo_build_pairs then ends
 * with an ncclAllReduce of the class table); "multi" reo_create_multi over all visible GPUs.
 *
 * stdin:  G S ngroups seed pval_reo pval_deg padj_deg n_iter n_conv
 *         S group ids, G reference flags, then G*S Int64 values column-major
 * stdout: "passes P", P lines "trace DEG NONDEG", G lines of 15 statistics (%.17g) */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "reo_hip.h"

#define CHECK(call)                                                                         \
    do {                                                                                    \
        int32_t rc_ = (call);                                                               \
        if (rc_ != REO_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, reo_last_error()); return 2; } \
    } while (0)

int main(int argc, char **argv)
{
    const char *mode = argc > 1 ? argv[1] : "single";
    long long G, S, ngroups, n_iter, n_conv;
    unsigned long long seed;
    double pval_reo, pval_deg, padj_deg;
    if (scanf("%lld %lld %lld %llu %lf %lf %lf %lld %lld", &G, &S, &ngroups, &seed, &pval_reo, &pval_deg, &padj_deg, &n_iter, &n_conv) != 9) return 1;
    int32_t *gid = malloc(sizeof(int32_t) * S);
    uint8_t *ref = malloc(G);
    int64_t *X = malloc(sizeof(int64_t) * G * S);
    double *result = malloc(sizeof(double) * 15 * G);
    int32_t *trace = malloc(sizeof(int32_t) * 2 * (n_iter > 0 ? n_iter : 1));
    if (!gid || !ref || !X || !result || !trace) return 1;
    for (long long s = 0; s < S; ++s) { int v; if (scanf("%d", &v) != 1) return 1; gid[s] = v; }
    for (long long g = 0; g < G; ++g) { int v; if (scanf("%d", &v) != 1) return 1; ref[g] = (uint8_t)v; }
    for (long long e = 0; e < G * S; ++e) { long long v; if (scanf("%lld", &v) != 1) return 1; X[e] = v; }

    reo_ctx *ctx = NULL;
    if (!strcmp(mode, "multi")) {
        CHECK(reo_create_multi(&ctx, 0, seed));                 /* 0 = every visible GPU */
    } else {
        CHECK(reo_create(&ctx, -1, seed));
        if (!strcmp(mode, "rccl")) {
            unsigned char id[REO_UNIQUE_ID_BYTES];
            CHECK(reo_comm_unique_id(id));                      /* rank 0 makes it, every rank receives it */
            CHECK(reo_comm_init_rank(ctx, id, 0, 1));
        }
    }
    CHECK(reo_set_groups(ctx, gid, S, (int32_t)ngroups));       /* unique(group), :353-357 */
    CHECK(reo_compute_thresholds(ctx, pval_reo));               /* :362 */
    CHECK(reo_set_matrix_i64(ctx, X, G, S, G));                 /* Matrix(df_expr), :652 -- after the groups: the upload is pipelined with the ranking and the pair kernel */
    CHECK(reo_build_pairs(ctx, 0));                             /* :363-392 */
    int32_t passes = 0;
    CHECK(reo_identify_degs(ctx, ref, pval_deg, padj_deg, (int32_t)n_iter, (int32_t)n_conv, result, &passes, trace));  /* :396-425 */
    /* error paths behave like the reference's exceptions: 'data' and 'group' of different lengths (:355) */
    CHECK(reo_set_groups(ctx, gid, S - 1, (int32_t)ngroups));
    if (reo_build_pairs(ctx, 0) != REO_EINVAL) { fprintf(stderr, "length mismatch not refused\n"); return 3; }
    reo_destroy(ctx);

    printf("passes %d\n", passes);
    for (int p = 0; p < passes; ++p) printf("trace %d %d\n", trace[2 * p], trace[2 * p + 1]);
    for (long long g = 0; g < G; ++g) {
        for (int c = 0; c < 15; ++c) printf("%s%.17g", c ? " " : "", result[(size_t)c * G + g]);  /* column-major G x 15 */
        printf("\n");
    }
    free(gid); free(ref); free(X); free(result); free(trace);
    return 0;
}
