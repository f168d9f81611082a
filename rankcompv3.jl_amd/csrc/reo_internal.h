// Internal declarations shared by the translation units of libreo_hip.so.
// Public C ABI: include/reo_hip.h.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "reo_hip.h"

namespace reo {

constexpr int kTileJ = 256;   // lanes per workgroup along j (4 waves x 64 lanes)
constexpr int kTileI = 32;    // gene rows per pair tile (one mirror word)
constexpr int kPlanes = 4;    // cL cH tL tH
constexpr int kUnitH = 32;    // i-tiles per K1 work unit
constexpr int kRJ = 4;        // genes j per lane in the tie-free pair kernel
constexpr int kRJTies = 2;    // genes j per lane in the tie-rich pair kernel (two band edges per pair)
constexpr int kMaxGenes = 262143;  // 18 position planes (above 65 535 genes: 32-bit transform rows, k1w_pairs<17/18>, light passes in the two-launch form)
constexpr int kGenePad = 1024;  // Gp is a multiple of this (= kTileJ * kRJ: every lane's genes exist)
constexpr int kRaw = 8;       // raw tally counters per gene (see k2_tally)
constexpr int kDeltaMax = 128;   // at most this many changed reference genes: update the tallies incrementally
constexpr int kSortChunk = 1024; // genes per bitonic sort in the ranking stage

// Device-resident loop state of the iteration driver (src/RankCompV3.jl:396-425),
// so that passes can be enqueued back to back without a host round trip.
struct IterState {
    int32_t done;      // convergence reached (:419-422): later launches return at once
    int32_t passes;    // executed passes of the while loop
    int32_t nref;      // sum(ref_gene_vec) of the current pass
    int32_t i_iter;    // :397,423
    int32_t ticket;    // finished workgroups of the kernel that ends a pass
    int32_t nn_acc;    // running sum(inds)
    int32_t delta_cnt[2];  // genes whose mask bit changes for the pass of that parity (see delta_genes in kernels.hip)
    int32_t need_full; // the next pass must run on the sorting path (tallies need a table scan, or no quantile windows)
    int32_t raw_pass;  // the tally counters hold the tallies of this pass (-1: none)
    int32_t nref_prev; // nref of the last executed pass (for recomputing its outputs)
    int32_t last_full; // the last executed pass ran on the sorting path: every output column is in place
    int32_t fault;     // 0, or why the passes cannot be trusted (codes below); api.hip answers REO_EHIP
    int32_t kstar;     // the BH cut (number of genes with padj <= padj_deg) of the last executed pass (-1: unknown)
    int32_t kcut_acc;  // running count for kstar (sorting path)
    // cycle watch of the light passes (kernels.hip, "cycle watch"): Brent's search for a pass whose mask equals an earlier one
    int32_t cyc_period; // 0: watching; p > 0: the mask in front of pass `passes` equals the one p passes earlier; -1: not watching
    int32_t cyc_pow;    // passes until the snapshot is renewed (0: no snapshot yet, or a sorting pass has stepped the mask since)
    int32_t cyc_lam;    // mask steps since the snapshot
    int32_t cyc_sel;    // which of the two snapshot buffers holds it
    int32_t cyc_ndiff;  // genes whose mask bit differs from the snapshot's
    int32_t pad[4];
};
constexpr int kFaultBarrier = 1;   // the persistent light kernel gave up at a grid barrier (bounded spin expired)
constexpr int kFaultTallies = 2;   // a gene's tallies broke their invariant: not a class table (kernels.hip, tallies_from)

// Light passes as two launches per pass (kernels.hip, kl_head / kl_rank): the loop state of one batch of launches as an
// append-only log, so that no kernel reads a word that another workgroup of the same launch writes.  rec[b] = the state
// in front of light pass b of the batch (rec of slot nlight: behind the last one), written by workgroup 0 of launch b,
// read by the launches after it.
constexpr int kLightBatch = 128; // most light passes per batch of launches (the first batch of a call: 32)
struct LightRec {
    int32_t active;     // pass `t` is to run as a light pass
    int32_t t;          // passes executed so far = index of the next pass
    int32_t nref, nref_prev, done, need_full, raw_pass;  // as in IterState
    int32_t ran;        // a light pass of this batch has been completed
    int32_t dcnt;       // genes whose mask bit changed in front of pass t
    int32_t kstar;      // the BH cut of the mask step that made this record (-1: none, the record comes from a sorting pass)
    int32_t cper, cpow, clam, csel;  // cycle watch, as IterState::cyc_*
};                      // (no padding array inside: copying one through registers made the compiler keep it in LDS, indexed by a
                        //  thread id that it computed from the dispatch packet -- a 3 us read of host memory at kernel start)
// Counters of one light pass.  A launch cannot end before its atomics have been performed, and atomics on one address
// are performed one after the other (about 12 ns each): 316 waves adding to one word kept kl_rank alive for 3 us after
// its last instruction.  Sums that need no return value are therefore spread over kSpread cache lines (by workgroup)
// and added up by their readers.
constexpr int kHistParts = 8;   // partial histograms of the BH ranks, one per XCD (kernels.hip, kl_rank)
constexpr int kSpread = 8;
constexpr int kListCap = 16;   // genes near the BH cut that one workgroup of kl_rank can list
constexpr int kListWgs = 1024;  // workgroups of 256 genes that the lists and the block moments are sized for (262 144 genes)
static_assert(kMaxGenes <= kListWgs * 256, "the light passes' lists and block moments (clist, part) hold kListWgs workgroups of 256 genes");
constexpr int kPartPer = kListWgs / 256;  // block moments per thread when every workgroup combines all of them
constexpr int kListStride = kListWgs + kListWgs * kListCap * 2;  // int32 per parity: counts by workgroup, then (word, gene) pairs
constexpr int kOneListCap = 16;   // one-launch form (kl_one): genes that one workgroup can list per pass ...
constexpr int kOneListMax = 512;  // ... and that a mask step takes in all (more: the pass goes to the sorting path)
constexpr int kOneStride = 256 + 256 * kOneListCap * 4;  // int32 per parity: counts by workgroup, then entries of 16 bytes (gene | bit << 31, 0, delta1)
struct LightCnt {
    int32_t cnt_a, cnt_b;            // members of the two quantile windows (slot allocation: returned values)
    int32_t pad[30];
    int32_t below_a[kSpread][32];    // values below window A   ([k][0] used: one cache line per part)
    int32_t below_b[kSpread][32];
    int32_t sig[kSpread][32];        // genes with a finite BH rank
    int32_t nsure[kSpread][32];      // one-launch form: genes outside the lists whose BH rank is surely within the cut's band
    int32_t nlow[kSpread][32];       // two-launch form: genes whose BH rank lies below the histogram's first bin (kernels.hip, hist_first)
};
// everything the launches after pass b of a batch need of it: rec = the state in FRONT of pass b; the rest is made by
// pass b itself.  Zeroed by the host in front of the batch.
struct LightSlot {
    LightRec rec;
    int32_t bfail;      // pass b lost a quantile window: it is redone on the sorting path
    int32_t cdiff;      // cycle watch: mask bits that differ from the snapshot after launch b's mask step (workgroup 0 keeps it)
    int32_t cfound;     // ... and, when none does, the period (the launch after this one stops the batch)
    int32_t pad0[5];
    double wnext[4];    // quantile windows for the pass after pass b
    double se_base;     // one-launch form: the se that launch b bracketed its genes' p-values around (the se of the pass before)
    double eta;         // ... and the half width of the bracket, relative
    double pad1[15];
    LightCnt lc;
};
struct LightState { LightSlot slot[kLightBatch + 2]; };
static_assert(sizeof(LightSlot) % 128 == 0, "LightSlot layout");

void set_error(const char *fmt, ...);

#define REO_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) {                                                          \
            ::reo::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),      \
                             __FILE__, __LINE__);                                        \
            return e_ == hipErrorOutOfMemory ? REO_ENOMEM : REO_EHIP;                    \
        }                                                                                \
    } while (0)

// Process-wide cache of device (and pinned host) blocks (api.hip).  The drop-in call makes a context, runs and destroys it
// (julia/RankCompV3HIP.jl): about forty hipMalloc / hipFree pairs per call cost 6 ms at config 3 -- as much as the whole step.  A
// released block goes to a free list instead (up to REO_DEVICE_CACHE_MB, default 16384; 0 switches the cache off) and the next
// request of that size on that device takes it from there; reo_trim_memory() gives everything back.  hipFree waits for the device
// before it unmaps; a cached block must be just as idle before someone else gets it, so release() waits for the device unless the
// caller says it has waited already (reo_destroy: after its streams, once for all of its buffers).
hipError_t pool_alloc(void **p, size_t bytes, bool pinned);
void pool_free(void *p, size_t bytes, bool pinned);
extern thread_local bool tl_release_synced;   // the releasing thread has waited for everything that used the blocks it releases
// Streams and events of destroyed contexts wait in the same kind of store (a context per call made and destroyed ~6 streams and ~40
// events: 1.2 ms of an 8 ms call).  kind: streams 0 normal / 1 high priority; events 0 without timing / 1 with.  A handle goes back only
// from reo_destroy, behind the waits for every stream of the context.
hipError_t handle_stream(hipStream_t *s, int kind);
hipError_t handle_event(hipEvent_t *e, int kind);
void release_stream(hipStream_t s, int kind);
void release_event(hipEvent_t e, int kind);

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }  // locals are freed on every return path; context members with the context
    int32_t ensure(size_t count)
    {
        if (count <= n && p) return REO_OK;
        release();
        REO_HIP_CHECK(pool_alloc(reinterpret_cast<void **>(&p), count * sizeof(T), false));
        n = count;
        return REO_OK;
    }
    void release()
    {
        if (p) {
            if (!tl_release_synced) (void)hipDeviceSynchronize();   // (what hipFree did implicitly)
            pool_free(p, n * sizeof(T), false);
        }
        p = nullptr;
        n = 0;
    }
};

struct StageTimer {
    hipEvent_t a = nullptr, b = nullptr;
};

}  // namespace reo

struct reo_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    uint64_t seed = 0;
    int rank = 0, world = 1;
    reo_allreduce_fn ar = nullptr;
    void *ar_user = nullptr;
    reo_allgather_fn ag = nullptr;
    void *ag_user = nullptr;

    // problem
    int64_t G = 0, S = 0, ld = 0;
    int dtype = 0;  // 0 none, 1 f64, 2 i64
    const void *dX = nullptr;
    reo::DevBuf<unsigned char> dX_owned;
    std::vector<int32_t> group_id;
    int ngroups = 0;
    std::vector<int32_t> thr;  // 2 x ngroups
    bool thr_set = false;

    // rank/band transform output (samples re-ordered so groups are contiguous)
    int Gp = 0, Wp = 0;  // padded gene count, 32-bit words per bit row
    std::vector<int32_t> goff;   // ngroups+1 offsets into the sorted sample order
    std::vector<int32_t> goff32; // the same with every group padded to whole blocks of 32 sample slots
    // bit planes over 32-sample blocks (transform.hip, t_slice); nblk = goff32.back() / 32
    reo::DevBuf<uint4> pos;     // [nblk][4][Gp]  planes 4q..4q+3 of the position of gene g in its sample's sorted order
    reo::DevBuf<uint4> lo;      // [nblk][Gp][4]  16 plane words of the first position of the tie band (plane k in word (k+15)%16)
    reo::DevBuf<uint4> hi;      // [nblk][Gp][4]  the same for one past the last position of the tie band
    reo::DevBuf<uint16_t> t_pos16, t_lo16, t_hi16;  // [S32][Gp] the three numbers before slicing (scratch)
    reo::DevBuf<uint32_t> t_pos32, t_lo32, t_hi32;  // the same for more than 65 535 genes (32-bit positions; transform.hip, t_slice_big)
    reo::DevBuf<uint32_t> t_vin32, t_vout32;        // by-slot scratch of the bucket ranking (!GENL); gene indices of the A/B build's segmented sort, 32-bit
    reo::DevBuf<int32_t> goff_dev;  // group offsets in blocks
    // transform scratch (grow-only, freed with the context)
    reo::DevBuf<uint64_t> t_kin, t_kout;
    reo::DevBuf<uint16_t> t_vin, t_vout;
    reo::DevBuf<unsigned char> t_temp;
    reo::DevBuf<int32_t> t_order, t_flags, t_slots;
    reo::DevBuf<uint32_t> unit_map;  // K1 work units: panel << 16 | i-range
    std::vector<uint32_t> unit_map_host;          // what unit_map holds (launch_k1 uploads only a different list)
    std::vector<int32_t> t_order_host, t_slots_host, goff_blocks_host;  // what t_order / t_slots / goff_dev hold (transform: uploaded when they change)
    const int32_t *t_meta_ptr[3] = {nullptr, nullptr, nullptr};
    int32_t *host_flags = nullptr;       // pinned: the transform's flags on their way to the host
    hipEvent_t ev_flags = nullptr;       // ... and the point of the stream at which they have arrived
    bool table_prezeroed = false;        // the transform has already queued the clearing of the class table (behind its flags copy)
    const uint32_t *unit_map_uploaded = nullptr;  // ... and into which allocation
    size_t k1_items_n = 0;           // wave form of K1: items of the launch being made (one work item per workgroup: side << 31 | wave chunk << 16 | i-tile; lists: k1_wave_items)
    bool transformed = false;
    // pipelined upload of a host matrix (transform.hip, eager_upload): chunks on an upload stream, ranked as they arrive
    int eager_mode = 2;                  // REO_EAGER_UPLOAD: 0 off, 1 transform only, 2 (default) the pair kernel's sides as well
    int eager_chunk = 0;                 // REO_EAGER_CHUNK: columns per chunk (0: about 8 MB)
    int eager_gate = 1;                  // REO_EAGER_GATE=0: the host reads the transform's flags before it launches a side (A/B)
    hipStream_t up = nullptr, rk = nullptr;   // the upload stream; the stream that widens, ranks and slices the chunks (high priority)
    hipEvent_t ev_up[8] = {nullptr};     // chunk k has arrived (ring)
    hipEvent_t ev_rk[4] = {nullptr, nullptr, nullptr, nullptr};   // fork from / join into the context's stream; a side's planes are in place (two sides)
    reo::DevBuf<int32_t> e_lists;        // [S] columns, [S + padding] slots, in column order
    // narrowed upload of Int64 matrices: host threads convert a chunk to 16- or 32-bit numbers in a pinned staging slot, a kernel widens it
    int upload_threads = 12;             // REO_UPLOAD_THREADS (0: the caller's array goes over the link as it is)
    unsigned char *stage_h[3] = {nullptr, nullptr, nullptr};   // pinned staging slots (from the block cache)
    size_t stage_cap = 0;                // bytes of each
    reo::DevBuf<unsigned char> stage_d[3];
    hipEvent_t ev_stage[3] = {nullptr, nullptr, nullptr}, ev_widen[3] = {nullptr, nullptr, nullptr};
    int64_t narrowed_bytes = 0;          // bytes the last pipelined upload put on the link (reo_get_info 19)
    bool eager_k1 = false;               // reo_set_matrix has already launched the pair kernel of comparison 0 on this data, groups and thresholds
    int has_ties = 0;
    int transform_in_lds = 0;  // the last transform sorted each sample inside one workgroup's LDS (transform.hip)

    // class table: [G][4 planes][Wp] 32-bit words
    reo::DevBuf<uint32_t> table;
    int built_k = -1;
    bool table_complete = false;        // world > 1: the shards' parts have been summed (api.hip, exchange_table)
    void *comm = nullptr;               // ncclComm_t of the in-library RCCL path (comm.hip), or null
    int spin_wait = 0;                  // REO_SPIN_WAIT=1: poll the stream on the hot path instead of the blocking wait (api.hip, stream_wait; measured: 0.04 ms per step)
    int check_hook_table = 1;           // REO_CHECK_HOOK_TABLE=0: skip the consistency scan of a table delivered by a caller's hook (timing tools)
    bool comm_dead = false;             // the communicator was aborted after a failure: every later build answers REO_ECOMM
    bool in_multi = false;              // this context is the leader or a peer of a reo_create_multi context (its exchange is multi_build_pairs', not launch_k1's)
    bool multi_one_device = false;      // reo_create_multi under REO_MULTI_ONE_DEVICE=1 (test seam: shards share one device, no RCCL)
    std::vector<reo_ctx *> peers;       // reo_create_multi: the contexts of devices 1.. owned by this (leader) context
    // one-vs-rest with > 2 groups: per-group pair counts shared by the comparisons (kernels.hip, k1_group_counts)
    reo::DevBuf<uint16_t> gcounts;      // [ngroups + 1][Gp/32][4][Gp][8]
    bool gc_valid = false;
    int k1_wave = 1;                    // REO_K1_WAVE=0: the workgroup form of the pair kernel (round 2) instead of the wave form
    int k1_order = 0;                   // REO_K1_ORDER=1 (experiment): the wave form's items i-tile-fastest inside a chunk instead of chunk-fastest inside an i-tile
    int k1_half = 1;                    // REO_K1_HALF=0: no half-height items in the last round of the wave form's launch
    int n_cus = 256;                    // compute units of the device (the wave form's item slots = CUs x 4 SIMDs x waves per SIMD)
    int share_counts = 1;               // REO_SHARE_GROUP_COUNTS=0 recounts per comparison instead
    int last_k1_shared = 0;             // reo_get_info: how the last class table was built
    int64_t tiles_owned = 0, tiles_total = 0;
    int k1_cj = 0, k1_q = 0;  // K1 geometry of the last build: genes j per workgroup, j-chunks per panel
    // gather form of the exchange (kernels.hip, x_pack / x_expand_*): every work unit of the last build (panel << 16 | i-range,
    // owner = index % world), pack and gather buffers
    std::vector<uint32_t> units_all_host, units_all_dev;   // ... and what the device copy holds
    const uint32_t *units_all_uploaded = nullptr;
    reo::DevBuf<uint32_t> units_all, xsend, xrecv;
    // pipelined exchange (several shards, wave form of the pair kernel): the shard's units are counted in waves, and wave w is
    // packed, gathered and unpacked on a second stream while wave w + 1 is being counted (kernels.hip, launch_k1)
    int x_waves = 4;                    // REO_EXCHANGE_WAVES (1: the whole exchange behind the pair kernel, as in round 3)
    bool x_pipelined = false;           // the last launch_k1 has already exchanged the table
    hipStream_t k1s[2] = {nullptr, nullptr}, xs = nullptr;
    hipEvent_t ev_fork = nullptr, ev_k1[8] = {nullptr}, ev_k1_join[2] = {nullptr, nullptr}, ev_x = nullptr;
    struct ItemList { reo::DevBuf<uint32_t> buf; size_t n = 0; uint64_t key[5] = {0, 0, 0, 0, 0}; };
    ItemList k1_wave_items[8];          // the item lists of the waves (kept until the geometry changes)
    reo::DevBuf<uint32_t> k1_park[2];   // range items of the pipelined upload: parked counts of a side, [items][kParkSlot] (kernels.hip)
    reo::DevBuf<int32_t> k1_park_ge[2]; // ... and which slots hold n_ge of their own
    int64_t eager_range_launches = 0;   // launches of the pair kernel that the last pipelined upload made over RANGES of a side's blocks (reo_get_info 20)
    int eager_ranges = -1;              // REO_EAGER_RANGES: ranges of sample blocks per side in the pipelined upload (1: whole sides, as in round 5; -1: by shape)
    reo::DevBuf<int32_t> check_flag;    // [1] verdict of k_check_table (kernels.hip, launch_check_table)

    // iteration state
    reo::DevBuf<uint32_t> refbits[2];   // [Wp]
    reo::DevBuf<uint8_t> refbytes[2];   // [Gp]
    reo::DevBuf<int32_t> raw;           // [G][8]
    reo::DevBuf<uint32_t> delta_list;   // [2][Gp] changed genes (gene << 1 | added) per pass parity
    reo::DevBuf<int32_t> cont;          // [G][9]
    reo::DevBuf<double> result;         // [15][G]
    reo::DevBuf<double> sorted_d;       // [G]
    reo::DevBuf<double> sorted_p;       // [G]
    reo::DevBuf<uint32_t> rank_s, rank_a;  // [G]
    reo::DevBuf<double> scal;           // [8] device scalars (se, ...)
    reo::DevBuf<double> blockmin;       // [<= 64]
    reo::DevBuf<double> chunk_v;        // [nchunk][kSortChunk] chunk-sorted delta1
    reo::DevBuf<uint16_t> chunk_i;      // [nchunk][kSortChunk] gene offsets inside the chunk
    reo::DevBuf<double> part;           // [<= 256][3] slice moments per block
    reo::DevBuf<reo::IterState> state;  // [1]
    reo::DevBuf<int32_t> trace;         // [n_iter][2]
    reo::DevBuf<int32_t> modes;         // [K2 launches] 1 = the launch scanned the whole table, 0 = incremental update or skipped
    reo::DevBuf<double> cand;           // [2 parities][2 windows][64] light passes: values inside the quantile windows
    reo::DevBuf<unsigned> gridbar;      // [1] arrival counter of the persistent light kernel's grid barrier
    reo::DevBuf<reo::LightState> lstate;  // [1] batch log of the two-launch light passes
    reo::DevBuf<int32_t> clist;         // [2][256 + 256 * kListCap * 2] genes near the BH cut, by workgroup (kernels.hip, kl_rank)
    int light_band = 32;                // REO_LIGHT_BAND (tests)
    bool cycle_watch = true;            // REO_CYCLE=0 switches the cycle watch of the light passes off (every pass is then executed)
    reo::DevBuf<uint8_t> snap;          // [2][Gp] cycle watch: mask snapshots
    int it_cycle_period = 0, it_cycle_at = 0, it_cycle_skipped = 0;  // the last reo_identify_degs: period found (0: none), in front of which pass, passes skipped
    int hist_below = 256;               // REO_HIST_BELOW (tests): ranks under the last cut that keep a histogram bin (kernels.hip, hist_first)
    int xcc_local = 0;                  // the per-XCD histogram atomics may stay in the XCD's L2 (checked once per context: kernels.hip, xcc_selftest)
    int light_window = 24, light_min_g = 4096;  // set from kernels.hip's constants in reo_create (REO_LIGHT_WINDOW, REO_LIGHT_MIN_G)
    int light_mode = 1;                 // 0 sorting passes only, 1 light passes as two launches each (the default), 2 as one persistent launch,
                                        // 3 as ONE launch each (round 4: measured slower, 24.0 against 23.1 us per pass; opt-in) (REO_LIGHT)
    reo::DevBuf<int32_t> olist;         // [2][kOneStride] one-launch form: the genes near the BH cut with their delta1, by workgroup (kernels.hip, kl_one)
    reo::DevBuf<int32_t> hist, mrank;   // [2][G padded to whole 32768-bin rounds], [Gp] light passes: histogram of the BH ranks (by launch parity), the ranks
    // parameters of the running reo_identify_degs call (kernels.hip, iter_args)
    double it_pval_deg = 1.0, it_padj_deg = 0.05;
    int it_n_iter = 0, it_n_conv = 0, it_a0 = 0, it_b0 = 0;
    int k2_idx = 0;                     // K2 launches of the running call
    int it_light_form = 1;              // the form of light pass the running call uses (light_mode, or 1 above 65 535 genes)
    bool it_no_light = false;           // the running call has given up on light passes (two light batches in a row completed no pass)
    reo::IterState *host_state = nullptr;  // pinned
    bool state_mirror_wanted = true;       // REO_STATE_MIRROR=0 (read once, in reo_create, like every other switch)
    bool debug_passes = false, debug_stamps = false, k1_stamps = false;  // REO_DEBUG_PASSES, REO_DEBUG_STAMPS, REO_K1_STAMPS (diagnostics)
    bool state_mirror = false;             // the kernels write what the host reads of IterState straight into host_state (REO_STATE_MIRROR=0: a copy per batch)
    uint8_t *host_ref = nullptr;           // pinned: the caller's reference mask on its way to the device (reo_identify_degs)
    size_t host_ref_cap = 0;

    // timing
    bool profiling = false;
    double t_ms[REO_NTIMINGS] = {0};
    std::vector<std::pair<int, reo::StageTimer>> pending;  // (slot, events)
    std::vector<reo::StageTimer> pool;
    std::vector<size_t> open;  // indices into pending of timers not yet closed (tic/toc nest)
    std::vector<int32_t> k2_modes;  // per pass of the last identify_degs call: 1 = full table scan in K2
    size_t k2_seen = 0;
};

namespace reo {

// transform.hip
int32_t run_transform(reo_ctx *c);
int32_t eager_upload(reo_ctx *c, const void *hX, int64_t hld, bool with_k1);  // host matrix -> HBM in chunks, ranked (and paired) as they arrive
int32_t upload_columns(reo_ctx *c, const void *hX, int64_t hld, int64_t G, int64_t ncols, void *dX, int dtype);  // a host matrix into a device matrix (ld = G), chunked, Int64 narrowed
int32_t ensure_upload_streams(reo_ctx *c);                                      // c->up, c->rk and their events (created on first use)
int32_t ensure_staging(reo_ctx *c, size_t slot_bytes);                          // three pinned + device staging slots of at least that size, their events
void host_parallel(int nthreads, int ntasks, const std::function<void(int)> &fn);   // fn(0 .. ntasks - 1) on the process-wide pool of host threads (and the caller)

// kernels.hip
// range: the launch covers sample blocks [b0, b1) of its ONE side (counted from the side's first block); the counts wait in the context's park
// buffer between the ranges of a side and the launch of the last range classifies (kernels.hip, K1Args::park).  prepare: only make
// (and upload) the unit map and the item list of `sides` -- the pipelined upload does that before it queues any wait on the stream.
struct K1Range { int b0, b1; bool first, last; };
int32_t launch_k1(reo_ctx *c, int k, int sides = 3, bool keep_table = false, const int32_t *gate = nullptr, const K1Range *range = nullptr,
                  bool prepare = false);  // sides: bit 0 = the comparison's own group, bit 1 = the rest (wave form; eager_upload)
int64_t exchange_unit_words(const reo_ctx *c);   // uint32 per packed work unit
int32_t exchange_units_per_rank(const reo_ctx *c);
int32_t launch_pack_units(reo_ctx *c, int m0 = 0, int mcnt = -1, uint32_t *send = nullptr, hipStream_t st = nullptr);    // this shard's units (all, or slots m0 .. m0 + mcnt - 1) -> c->xsend / send
int32_t launch_expand_units(reo_ctx *c, int m0 = 0, int mcnt = -1, const uint32_t *recv = nullptr, hipStream_t st = nullptr);  // every shard's pack -> the table: the others' words and their mirrors
int32_t launch_counts(reo_ctx *c, int64_t i0, int64_t i1, int64_t j0, int64_t j1, uint16_t *d_gt, uint16_t *d_eq);
int32_t launch_check_table(reo_ctx *c, int *bad);  // consistency of an exchanged class table (kernels.hip, k_check_table)
int32_t launch_decode(reo_ctx *c, int64_t i0, int64_t i1, int64_t j0, int64_t j1, uint8_t *d_code);
int32_t launch_pack_ref(reo_ctx *c, const uint8_t *d_bytes, uint32_t *d_bits);
int32_t launch_cycle_skip(reo_ctx *c, int skip, int period);  // cycle watch: move the loop state `skip` passes on (whole periods), fill their trace, end the watch
int32_t launch_iter_init(reo_ctx *c, const uint8_t *host_ref, const IterState *host_state);  // mask + loop state from pinned host memory, clears
int32_t launch_tally(reo_ctx *c, int nref);
int32_t launch_full_pass(reo_ctx *c, bool replay);
int32_t launch_light_persistent(reo_ctx *c);
int32_t launch_light_batch(reo_ctx *c, int nlight);
int32_t xcc_selftest(reo_ctx *c, int *ok);
int32_t light_min_genes();
int32_t light_window();
int32_t launch_mccullagh(reo_ctx *c, const int32_t *d_cont, int64_t n, double *d_out);

// comm.hip: in-library RCCL.  Returns REO_OK after enqueueing the sum on c->stream, 1 when no communicator is attached
int32_t comm_allgather(reo_ctx *c, const void *send, void *recv, int64_t bytes_per_rank, hipStream_t st = nullptr);
void comm_release(reo_ctx *c);
void comm_abort(reo_ctx *c);                     // abort the communicator, mark the context unusable for exchanges
int32_t comm_wait(reo_ctx *c);                   // stream wait that watches the communicator (async errors, time limit)
int32_t multi_build_pairs(reo_ctx *lead, int32_t k, int32_t (*build_local)(reo_ctx *, int32_t));

// timing helpers (api.hip)
void tic(reo_ctx *c, int slot);
void toc(reo_ctx *c);
void collect_timings(reo_ctx *c);

}  // namespace reo
