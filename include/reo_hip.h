/*
 * reo_hip.h -- C ABI of libreo_hip.so, the MI355X (gfx950) implementation of
 * RankCompV3.jl's REO pairwise-comparison hot path.
 *
 * The reference has no FFI: the seam this library replaces is the ordinary
 * Julia call identify_degs(data, group, gene_names, pval_reo, pval_deg,
 * padj_deg, ref_gene, n_iter, n_conv) at /root/reference/src/RankCompV3.jl:
 * 339-350 (called once from reoa at :652-662).  A <=100-line Julia shim with
 * that exact signature drives the entry points below through ccall (see
 * INTEGRATION.md and julia/RankCompV3HIP.jl); the Python mirror in
 * rankcompv3.jl_amd/ binds the same symbols through ctypes.
 *
 * Conventions
 *   - every function returns 0 on success or a negative reo_status; the
 *     message of the last failure on the calling thread is reo_last_error().
 *     REO_EINVAL corresponds to the reference's DimensionMismatch /
 *     ArgumentError / BoundsError paths (src/RankCompV3.jl:355-356,411).
 *   - the caller owns every array it passes; host inputs are copied during
 *     the call and no host pointer is retained after return; outputs are
 *     caller-allocated.
 *   - a context is single-owner (not thread-safe); distinct contexts may be
 *     used from distinct threads.  Every call blocks until its outputs are in
 *     the caller's arrays and its inputs have been read.  ONE exception:
 *     on one GPU with nothing to exchange reo_build_pairs returns with the
 *     pair kernel still running (see there); all work of a context is ordered
 *     on one stream, so later calls need no synchronisation by the caller, a
 *     host timer around reo_build_pairs alone measures the launch only, and an
 *     asynchronous failure of that kernel is reported (REO_EHIP) by the next
 *     call that waits, which also drops the class table so that a retry
 *     rebuilds it.
 *   - matrices are column-major (the layout Julia hands over at :652).
 */
#ifndef REO_HIP_H
#define REO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct reo_ctx reo_ctx;

enum reo_status {
    REO_OK = 0,
    REO_EINVAL = -1, /* bad argument / shape mismatch / reference error path */
    REO_EHIP = -2,   /* HIP runtime failure (including "no GPU")           */
    REO_ECOMM = -3,  /* RCCL or the all-reduce hook failed / table not exchanged */
    REO_ENOMEM = -4  /* host or device allocation failed                   */
};

/* Library version, major*10000 + minor*100 + patch. */
int32_t reo_version(void);

/* Message of the last error on this thread (library-owned, valid until the
 * next failing call on the same thread). */
const char *reo_last_error(void);

/* Create a context on HIP device `device` (-1 = the current device).  `seed`
 * keys the tie coin stream that stands in for the reference's unseeded
 * rand(Bool) in is_greater (src/RankCompV3.jl:71-77).  Fails with REO_EHIP
 * when no gfx950 device is usable: there is no CPU fallback. */
int32_t reo_create(reo_ctx **out, int32_t device, uint64_t seed);
void reo_destroy(reo_ctx *ctx);

/* Memory of destroyed contexts is kept for the next one: a caller that makes a context per identify_degs call (the Julia shim
 * does) would otherwise spend as long in hipMalloc / hipFree as in the computation (6 of 14 ms at 20 000 x 1 000).  Released
 * device and pinned-host blocks wait in a process-wide cache, at most REO_DEVICE_CACHE_MB megabytes of them (environment,
 * default 16384; 0 = no cache, every release is a hipFree), and so do the streams and events of destroyed contexts;
 * reo_trim_memory() returns all of it to the driver now. */
int32_t reo_trim_memory(void);

/* ---- several GPUs ---------------------------------------------------------------------------------------
 * The reference is one process with shared-memory threads (src/RankCompV3.jl:368,402) and has no multi-device
 * path; this is the build's own.  The pair tiles of the G x G triangle are dealt to `world` shards; every shard
 * builds the class-table words of its tiles, ONE exchange per class table (the shards' bits are disjoint: an
 * all-gather of their own words, or an in-place sum) gives every shard the whole table, and the iteration passes
 * run with no further collective.
 *
 * (a) one process, all GPUs: reo_create_multi(&ctx, n_gpus (0 = all visible), seed) returns a context that is
 *     used exactly like a one-GPU context; it drives one device context each; inside reo_build_pairs the peers pack
 *     the table words of their own pair tiles and hand them to device 0 (grouped ncclSend / ncclRecv), which unpacks
 *     them, derives the mirror words and runs the passes.  This form is NOT pipelined (the hand-over starts when every
 *     device has finished its tiles) and the passes run on the leader only: for throughput use (b).  NOT yet run on more than one GPU (no such box in this
 *     project's pool): the orchestration is tested with the shards sharing one device (REO_MULTI_ONE_DEVICE=1).
 * (b) one process per GPU: rank 0 calls reo_comm_unique_id and hands the 128 bytes to the other ranks by any
 *     means; every rank calls reo_comm_init_rank(ctx, id, rank, world) (ncclCommInitRank + reo_set_shard).
 *     reo_build_pairs then ends with an ncclAllGather of the shards' own table words on the context's stream (see
 *     reo_set_allgather for the protocol); all ranks get identical results.  A rank that fails inside reo_build_pairs
 *     aborts its communicator (ncclCommAbort) so that its peers' collective ends with REO_ECOMM instead of waiting;
 *     the wait itself watches ncclCommGetAsyncError and gives up after REO_COMM_TIMEOUT_S seconds (default 300).  After
 *     such a failure the context needs a new communicator.  NOT yet run with more than one rank on hardware.
 * (c) bring your own collective: reo_set_shard + reo_set_allgather or reo_set_allreduce (hooks below).  */
enum { REO_UNIQUE_ID_BYTES = 128 };
int32_t reo_create_multi(reo_ctx **out, int32_t n_gpus, uint64_t seed);
int32_t reo_comm_unique_id(void *id /* REO_UNIQUE_ID_BYTES */);
int32_t reo_comm_init_rank(reo_ctx *ctx, const void *id, int32_t rank, int32_t world);

/* This context builds only the pair tiles it owns out of `world` shards.  Default (0, 1) = everything.
 * With world > 1 and neither a communicator nor a hook attached, reo_build_pairs leaves the shard's own part
 * of the table (reo_get_codes shows it; pairs of other shards read as class 4) and reo_tally /
 * reo_identify_degs refuse with REO_ECOMM. */
int32_t reo_set_shard(reo_ctx *ctx, int32_t rank, int32_t world);

/* Hook called once per reo_build_pairs when world > 1 and no communicator is attached: it must arrange for
 * `count` int32 values at device pointer `dev_buf` (the class table) to be summed in place across all shards,
 * ORDERED ON `stream` (a hipStream_t): everything the library enqueued on `stream` before the call has to
 * precede the sum, and the sum has to precede whatever is enqueued on `stream` afterwards.  A hook that enqueues
 * the collective on `stream` (RCCL ncclAllReduce(..., stream), or torch.distributed.all_reduce under
 * torch.cuda.stream(ExternalStream(stream))) needs no host synchronisation; a hook that works on the host must
 * synchronise `stream` itself before and after.  Return 0 on success.  A hook that fails must abort its own
 * communicator: the library cannot release peers that wait inside a caller's collective.  The table a hook delivers is
 * scanned for consistency (a pair in two states, bits outside the table: REO_ECOMM). */
typedef int32_t (*reo_allreduce_fn)(void *dev_buf, int64_t count, void *stream, void *user);
int32_t reo_set_allreduce(reo_ctx *ctx, reo_allreduce_fn fn, void *user);

/* The cheaper form of the same exchange, and what the in-library RCCL path does: every shard packs the table words of
 * ITS OWN pair tiles (upper triangle only, about a quarter of what an in-place sum of the whole table moves), the
 * packs are gathered, and every shard unpacks the others' words and derives the mirror words (the pair seen from the
 * other gene, src/RankCompV3.jl:386) itself.  The hook is an all-gather: `bytes_per_rank` bytes at device pointer
 * `send` of every shard have to arrive at `recv + r * bytes_per_rank` of every shard, r = the sender's shard number,
 * ordered on `stream` like the sum above (RCCL: ncclAllGather(send, recv, bytes_per_rank, ncclUint8, comm, stream)).
 * When both hooks are set this one is used.
 * CALLED SEVERAL TIMES PER BUILD, ON ANOTHER STREAM: with two groups the shard's pair tiles are counted in up to 8 waves
 * (REO_EXCHANGE_WAVES, default 4) and the hook is called once per wave -- `bytes_per_rank` differs from call to call but is the
 * same on every shard for the same call -- with `stream` = the context's EXCHANGE stream, not the stream of other calls; the
 * ordering rule above holds per call, on the stream that call names.  Every shard must make the same number of calls: the
 * number of waves follows from REO_EXCHANGE_WAVES / REO_K1_STAMPS / REO_K1_WAVE in the environment, which must therefore be
 * equal on all shards (the in-library communicator checks that in reo_comm_init_rank and answers REO_EINVAL; with a caller's
 * hook the caller has to see to it).  REO_EXCHANGE_WAVES=1: one call per build, on the context's own stream. */
typedef int32_t (*reo_allgather_fn)(const void *send, void *recv, int64_t bytes_per_rank, void *stream, void *user);
int32_t reo_set_allgather(reo_ctx *ctx, reo_allgather_fn fn, void *user);

/* Expression matrix, G genes x S samples, column-major with leading dimension
 * ld >= G: the `data` argument of identify_degs (src/RankCompV3.jl:340) as
 * Matrix(df_expr) produces it (:652), eltype Float64 or Int64.  G must be in [2, 262143] and S in [2, 1048576];
 * with more than two groups G and S may not both exceed 65535; values must be finite.  reo_set_matrix_f64 / _i64
 * copy from host memory (pageable is fine) and have read all of it when they return; *_dev uses a buffer already resident in HBM (it
 * must stay valid until reo_build_pairs returns).
 * ORDER OF CALLS.  Any order of reo_set_matrix_*, reo_set_groups, reo_compute_thresholds works.  From HOST memory the cheap order is
 * groups and thresholds FIRST, the matrix last: the call then uploads the columns in chunks and pipelines them with the rest of
 * the work -- the samples of a chunk are ranked while the next chunk is crossing PCIe, and with two groups on one GPU the pair
 * kernel's items of a group (its "side" of every pair) are counted over ranges of that group's 32-sample blocks as its chunks arrive
 * (their 16-bit counts wait in device memory for the group's last range, which classifies), while the rest is still on its way.  The call still returns only when the whole matrix has been read (no host pointer is retained); the pair
 * kernel may be running then, exactly as after reo_build_pairs on one GPU, and the reo_build_pairs(ctx, 0) that follows has nothing
 * left to do (it is still the call that makes the class table current: keep it).  Results are bit-identical in every order.
 * VALUES.  +-Inf are accepted and compared as the reference's is_greater compares them (src/RankCompV3.jl:71-77): equal infinities
 * are neither tied nor greater (abs(Inf - Inf) = NaN is not < 0.1, Inf > Inf is false -- the pair (i, j), i < j, counts as "i not
 * greater" in that sample, no coin), an infinity against any other value compares as usual; log(0) = -Inf tables run unchanged.
 * A NaN is refused (REO_EINVAL, by whichever call reads the matrix: this one in the pipelined case): every comparison with a NaN
 * is false, which makes a NaN gene below every later and above every earlier gene -- row order, not an ordering.
 * A context that is used for several matrices should keep to that order each time: a matrix handed over while the groups of the LAST
 * problem are still set is ranked and paired with those, and all of it is done again when the new groups arrive (correct, but wasted).
 * REO_EAGER_UPLOAD=0 in the environment switches the pipelining off, =1 keeps it to the ranking. */
int32_t reo_set_matrix_f64(reo_ctx *ctx, const double *X, int64_t G, int64_t S, int64_t ld);
int32_t reo_set_matrix_i64(reo_ctx *ctx, const int64_t *X, int64_t G, int64_t S, int64_t ld);
int32_t reo_set_matrix_dev_f64(reo_ctx *ctx, const void *dX, int64_t G, int64_t S, int64_t ld);
int32_t reo_set_matrix_dev_i64(reo_ctx *ctx, const void *dX, int64_t G, int64_t S, int64_t ld);

/* Group of each sample: the `group` argument (src/RankCompV3.jl:341) recoded
 * to 0-based ids in order of first appearance (unique(), :353).  Length must
 * equal S (else REO_EINVAL = the DimensionMismatch of :355); ngroups must be
 * >= 2 (:356); any number of levels that the samples allow. */
int32_t reo_set_groups(reo_ctx *ctx, const int32_t *group_id, int64_t len, int32_t ngroups);

/* Stable-REO thresholds: get_major_reo_lower_count (src/RankCompV3.jl:81-92)
 * applied as at :362.  m is 2 x ngroups column-major: m[2k] for group k,
 * m[2k+1] for the rest.  reo_compute_thresholds derives them from pval_reo,
 * reo_set_thresholds overrides them, reo_get_thresholds reads them back. */
int32_t reo_compute_thresholds(reo_ctx *ctx, double pval_reo);
int32_t reo_set_thresholds(reo_ctx *ctx, const int32_t *m);
int32_t reo_get_thresholds(reo_ctx *ctx, int32_t *m);
/* The threshold function itself (host arithmetic), for tests. */
int32_t reo_threshold(int32_t sample_size, double pval_reo);

/* REO table build for comparison k (group k vs every other sample; with two
 * groups the reference only runs k = 0, :387-389): replaces the pair loop
 * src/RankCompV3.jl:363-392.  Runs the per-sample rank/band transform and the
 * pair-compare kernel and leaves the 4-bit class table in HBM.  The input matrix has been read when the call
 * returns.  On one GPU with nothing to exchange the pair kernel may still be running then: every later call on the
 * context is ordered behind it (one stream), and an asynchronous failure of the kernel is reported by the next call
 * that waits (REO_EHIP).  With shards (several GPUs, hooks) the call returns after the exchange has finished. */
int32_t reo_build_pairs(reo_ctx *ctx, int32_t k);

/* Parity hook: deterministic per-pair per-group counts for the ordered pairs
 * (i, j), i in [i0,i1), j in [j0,j1): n_gt = #samples with x_i > x_j and not
 * tied, n_eq = #tied samples (|x_i - x_j| < 0.1, src/RankCompV3.jl:72), each
 * laid out [(i-i0)][(j-j0)][group].  Computed from the same bit planes with the
 * same borrow chain as reo_build_pairs.  Needs matrix + groups only. */
int32_t reo_pair_counts(reo_ctx *ctx, int64_t i0, int64_t i1, int64_t j0, int64_t j1,
                        uint16_t *n_gt, uint16_t *n_eq);

/* Parity hook: class codes 3*(ic-1)+(it-1) in 0..8 of the ordered pairs of a
 * block (255 on the diagonal, which the table never sets; pairs owned by
 * another shard read as class 4 = all four bits clear), row-major [(i-i0)][(j-j0)] -- the column index of the
 * reference's R BitArray minus one (src/RankCompV3.jl:383-386). */
int32_t reo_get_codes(reo_ctx *ctx, int64_t i0, int64_t i1, int64_t j0, int64_t j1, uint8_t *code);

/* Per-gene 3x3 contingency builder, src/RankCompV3.jl:403: cont is G x 9
 * row-major (n11 n12 n13 n21 ... n33), ref_mask has one byte per gene
 * (non-zero = reference gene).  Needs the complete class table: with world > 1 and no exchange done by
 * reo_build_pairs (communicator or hook) it refuses with REO_ECOMM. */
int32_t reo_tally(reo_ctx *ctx, const uint8_t *ref_mask, int32_t *cont);

/* Iteration driver + McCullagh test, src/RankCompV3.jl:396-425 (and :225-259)
 * for the comparison built by reo_build_pairs.  result is G x 15 column-major
 * Float64: [pval padj n11..n33 delta1 delta2 se z1] (:398,405,415-416,665).
 * iters_run = number of executed passes of the while loop (:400); trace is
 * n_iter x 2 int32, (#DEG, #non-DEG) per pass (the :418 log line); either may
 * be NULL.  REO_EINVAL when the reference would throw (G < 10: the slice of
 * :411 is out of bounds).
 * A loop that does not converge repeats itself sooner or later: a pass depends on its reference set alone, so once the set in
 * front of a pass equals an earlier one (checked exactly, bit for bit) the remaining passes are the last period over and over.
 * The call then skips whole periods and executes only the remainder -- iters_run, trace and result are what n_iter executed
 * passes give (reo_get_info 16-18 says what happened; REO_CYCLE=0 in the environment executes every pass). */
int32_t reo_identify_degs(reo_ctx *ctx, const uint8_t *ref0, double pval_deg, double padj_deg,
                          int32_t n_iter, int32_t n_conv, double *result,
                          int32_t *iters_run, int32_t *trace);

/* McCullagh test on 3x3 tables given as 9 tallies each (n x 9 row-major),
 * evaluated by the device routine the iteration uses; out is n x 5 row-major
 * (pval, delta1, delta2, se, z1) -- src/RankCompV3.jl:225-259. */
int32_t reo_mccullagh(reo_ctx *ctx, const int32_t *cont, int64_t n, double *out);

/* Pseudo-bulk front end: pseudobulk_group (src/RankCompV3.jl:56-67, call site :608-612) for all
 * groups at once.  `order` lists the cells (0-based columns) of every output profile back to back --
 * the reference's shuffled partition, sample(1:c, c) + Iterators.partition (:62) -- and
 * chunk_ptr[o] .. chunk_ptr[o+1] delimits profile o (n_out + 1 entries, chunk_ptr[n_out] = n_order).
 * Each profile is the row-wise sum of its cells taken in that order (:63): exact for Int64,
 * bit-reproducible for Float64.  out is G x n_out column-major, caller-allocated.
 * dense: X is the G x C cell matrix, column-major (what the reference holds after CSV.read);
 * csc:   this build's container for sparse single-cell counts (colptr C+1, rowidx/val nnz).  The entries are uploaded in chunks:
 *        host threads (REO_UPLOAD_THREADS) check the row indices and narrow them (16 bits when G <= 65 536) and the Int64 values
 *        (16 / 32 bits when they fit) into pinned staging, the device widens them -- exact; 60 M entries: 28 -> 9 ms. */
int32_t reo_pseudobulk_dense_f64(reo_ctx *ctx, const double *X, int64_t G, int64_t C, int64_t ld,
                                 const int32_t *order, int64_t n_order, const int32_t *chunk_ptr,
                                 int32_t n_out, double *out);
int32_t reo_pseudobulk_dense_i64(reo_ctx *ctx, const int64_t *X, int64_t G, int64_t C, int64_t ld,
                                 const int32_t *order, int64_t n_order, const int32_t *chunk_ptr,
                                 int32_t n_out, int64_t *out);
int32_t reo_pseudobulk_csc_f64(reo_ctx *ctx, int64_t G, int64_t C, const int64_t *colptr, const int32_t *rowidx,
                               const double *val, const int32_t *order, int64_t n_order,
                               const int32_t *chunk_ptr, int32_t n_out, double *out);
int32_t reo_pseudobulk_csc_i64(reo_ctx *ctx, int64_t G, int64_t C, const int64_t *colptr, const int32_t *rowidx,
                               const int64_t *val, const int32_t *order, int64_t n_order,
                               const int32_t *chunk_ptr, int32_t n_out, int64_t *out);

/* Stage timers (HIP events on the library's stream), milliseconds, summed
 * since the last reo_reset_timings.  Index: 0 rank/band transform, 1 pair
 * kernel K1, 2 tally stage K2 (full scan or incremental update, sum), 3 iteration passes in total (K2 + the
 * statistics kernels K3, sum), 4 number of K2 launches (passes enqueued after
 * convergence return at once and are counted too), 5 number of K1 launches,
 * 6 exchange of the class table between shards (HIP events), 7 pseudo-bulk kernel, 8 K2 stage of the passes that scanned the whole table (sum), 9 their
 * number, 10 K2 stage of the passes that updated the tallies incrementally (sum), 11 host wall time inside reo_set_matrix_f64 / _i64
 * (the upload from host memory, with whatever was pipelined behind it). */
enum { REO_NTIMINGS = 12 };
int32_t reo_set_profiling(reo_ctx *ctx, int32_t on);
int32_t reo_reset_timings(reo_ctx *ctx);
int32_t reo_get_timings(reo_ctx *ctx, double *ms, int32_t n);

/* Facts about the current problem for roofline accounting and for mirroring
 * the shard ownership rule: 0 G, 1 S, 2 padded G (table row pitch in bits),
 * 3 class-table bytes, 4 data-has-ties flag, 5 pair tiles owned by this
 * shard, 6 pair tiles total, 7 gene rows per pair tile, 8 gene columns per
 * workgroup, 9 column chunks per panel, 10 tiles per work-unit column,
 * 11 padded sample slots, 12 the last class table came from the shared
 * per-group counts (more than two groups: the one-vs-rest comparisons count
 * every group once and keep the counts in HBM; REO_SHARE_GROUP_COUNTS=0 in the
 * environment recounts per comparison), 13 bytes held by those counts, 14 the last transform ranked every sample
 * inside one workgroup, 15 the iteration passes keep their rank histogram per XCD (the self-test of reo_create passed;
 * REO_XCC_LOCAL=0 switches it off), 16-18 the last reo_identify_degs: the period p of the cycle its iteration was found in
 * (0: none found), the pass in front of which the reference set equalled that of p passes earlier, and the passes that
 * were then skipped instead of executed (see reo_identify_degs; REO_CYCLE=0 switches the watch off), 19 the bytes that the last
 * reo_set_matrix_i64 / _f64 put on the PCIe link (chunks whose values all fit travel as int16 / int32 -- Float64 chunks of
 * integer-valued or single-precision numbers too, as int16 / int32 / float32 -- converted by REO_UPLOAD_THREADS host threads,
 * default 12, and widened on the device: bit-exact; 0 threads = the caller's array as it is), 20 the launches of the pair kernel that
 * the last pipelined reo_set_matrix_* made over a RANGE of a group's sample blocks (the pair kernel then starts before the whole group
 * has arrived; the counts of a range wait in HBM for the group's last range, which classifies -- REO_EAGER_RANGES=1 in the environment
 * launches whole sides only, as in round 5; 2..6 asks for that many ranges per side; default: by the number of blocks). */
int32_t reo_get_info(reo_ctx *ctx, int64_t *info, int32_t n);

#ifdef __cplusplus
}
#endif
#endif /* REO_HIP_H */
