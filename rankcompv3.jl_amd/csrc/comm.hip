// In-library RCCL path for multi-GPU runs (include/reo_hip.h: reo_comm_unique_id, reo_comm_init_rank,
// reo_create_multi).  The reference has no multi-device path (/root/reference/src/RankCompV3.jl uses
// Threads.@threads only, :368,402); what is exchanged here is this build's own intermediate, the class table.
//
// Protocol: the pair tiles of the G x G triangle are dealt to the shards (launch_k1, `unit % world`); every shard
// writes the table words of its tiles (forward and mirror bits) into a zeroed table.  One process per GPU
// (reo_comm_init_rank): every shard packs the forward words of its own units (upper triangle: 107 MB in all at 20 000
// genes), ONE ncclAllGather per class table hands every pack to every shard, and each shard unpacks the others' words
// and derives their mirror words itself (api.hip, exchange_table; kernels.hip, x_pack / x_expand_*) -- a quarter of the
// bytes that an in-place sum of the whole 205 MB table moves.  One process, all GPUs (reo_create_multi): the peers'
// packs go to the leader with grouped ncclSend / ncclRecv and are unpacked there.  The iteration passes then run with
// no collective at all.  Failure handling: comm_abort / comm_wait below -- a rank that cannot go on aborts its
// communicator, a waiting rank watches ncclCommGetAsyncError and a time limit; nobody blocks for ever.
// No run with more than one rank has happened on hardware (the pool offers one GPU per box).
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "reo_internal.h"

namespace reo {

#define REO_NCCL_CHECK(expr)                                                                  \
    do {                                                                                      \
        ncclResult_t r_ = (expr);                                                             \
        if (r_ != ncclSuccess) {                                                              \
            ::reo::set_error("%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__, __LINE__); \
            return REO_ECOMM;                                                                 \
        }                                                                                     \
    } while (0)

int32_t comm_allgather(reo_ctx *c, const void *send, void *recv, int64_t bytes_per_rank, hipStream_t st)
{
    if (!c->comm) return 1;
    const ncclResult_t r = ncclAllGather(send, recv, static_cast<size_t>(bytes_per_rank), ncclUint8, static_cast<ncclComm_t>(c->comm), st ? st : c->stream);
    if (r != ncclSuccess) { set_error("ncclAllGather failed: %s", ncclGetErrorString(r)); return REO_ECOMM; }
    return REO_OK;
}

void comm_release(reo_ctx *c)
{
    if (c->comm) (void)ncclCommDestroy(static_cast<ncclComm_t>(c->comm));
    c->comm = nullptr;
}

// A rank that cannot go on (allocation, launch or hook failure, an RCCL error) must not leave its peers blocked inside
// the collective: it aborts its communicator, which makes the peers' collective fail (comm_wait sees the error or runs
// into its limit), and every later call on this context answers REO_ECOMM.
void comm_abort(reo_ctx *c)
{
    if (c->comm) (void)ncclCommAbort(static_cast<ncclComm_t>(c->comm));
    c->comm = nullptr;
    c->comm_dead = true;
}

// Wait for the stream with the communicator watched: an asynchronous RCCL error or REO_COMM_TIMEOUT_S seconds (default
// 300) without the stream finishing aborts the communicator and returns REO_ECOMM instead of blocking for ever.
int32_t comm_wait(reo_ctx *c)
{
    if (!c->comm) { REO_HIP_CHECK(hipStreamSynchronize(c->stream)); return REO_OK; }
    double limit = 300.0;
    if (const char *e = getenv("REO_COMM_TIMEOUT_S")) limit = std::max(1.0, atof(e));
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; ++spin) {
        const hipError_t q = hipStreamQuery(c->stream);
        if (q == hipSuccess) return REO_OK;
        if (q != hipErrorNotReady) { set_error("stream failed while waiting for the exchange: %s", hipGetErrorString(q)); comm_abort(c); return REO_EHIP; }
        ncclResult_t ar = ncclSuccess;
        const ncclResult_t r = ncclCommGetAsyncError(static_cast<ncclComm_t>(c->comm), &ar);
        if (r != ncclSuccess || (ar != ncclSuccess && ar != ncclInProgress)) {
            set_error("RCCL reported an asynchronous error during the exchange: %s", ncclGetErrorString(r != ncclSuccess ? r : ar));
            comm_abort(c);
            return REO_ECOMM;
        }
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (el > limit) {
            set_error("the exchange did not finish within %.0f s (REO_COMM_TIMEOUT_S): a peer has probably failed; communicator aborted", limit);
            comm_abort(c);
            return REO_ECOMM;
        }
        if (spin > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));  // (an exchange takes well under a millisecond: spin first)
    }
}

}  // namespace reo

using namespace reo;

extern "C" {

int32_t reo_comm_unique_id(void *id)
{
    static_assert(sizeof(ncclUniqueId) == REO_UNIQUE_ID_BYTES, "REO_UNIQUE_ID_BYTES must match ncclUniqueId");
    if (!id) { set_error("id is null"); return REO_EINVAL; }
    ncclUniqueId u;
    REO_NCCL_CHECK(ncclGetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return REO_OK;
}

int32_t reo_comm_init_rank(reo_ctx *c, const void *id, int32_t rank, int32_t world)
{
    if (!c || !id) { set_error("null argument"); return REO_EINVAL; }
    if (!c->peers.empty()) { set_error("a multi-GPU context (reo_create_multi) manages its own communicator"); return REO_EINVAL; }
    int32_t rc = reo_set_shard(c, rank, world);
    if (rc) return rc;
    REO_HIP_CHECK(hipSetDevice(c->device));
    comm_release(c);
    c->comm_dead = false;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclComm_t comm = nullptr;
    REO_NCCL_CHECK(ncclCommInitRank(&comm, world, u, rank));
    c->comm = comm;
    // Every rank must issue the same sequence of collectives.  How many all-gathers a build makes (the waves of the pipelined
    // exchange, kernels.hip launch_k1) follows from per-process switches -- REO_EXCHANGE_WAVES, REO_K1_STAMPS, REO_K1_WAVE -- so
    // ranks started with different environments would wait for each other until the watchdog fires.  The switches are compared
    // here, once, with the communicator's first collective (which also shows that the ranks reach each other at all).
    const int32_t mine = c->x_waves | (c->k1_stamps ? 0x100 : 0) | (c->k1_wave ? 0x200 : 0);
    DevBuf<int32_t> cfg;
    if ((rc = cfg.ensure(static_cast<size_t>(world) + 1))) { comm_abort(c); return rc; }
    std::vector<int32_t> all(static_cast<size_t>(world), 0);
    hipError_t e = hipMemcpyAsync(cfg.p + world, &mine, sizeof mine, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        const ncclResult_t r = ncclAllGather(cfg.p + world, cfg.p, 1, ncclInt32, comm, c->stream);
        if (r != ncclSuccess) { set_error("ncclAllGather (agreement on the exchange switches) failed: %s", ncclGetErrorString(r)); (void)hipStreamSynchronize(c->stream); comm_abort(c); return REO_ECOMM; }
        e = hipMemcpyAsync(all.data(), cfg.p, sizeof(int32_t) * world, hipMemcpyDeviceToHost, c->stream);
    }
    if (e != hipSuccess) { set_error("reo_comm_init_rank: %s", hipGetErrorString(e)); (void)hipStreamSynchronize(c->stream); comm_abort(c); return REO_EHIP; }
    if ((rc = comm_wait(c))) return rc;   // (watches the communicator: a peer that never arrives ends in REO_ECOMM, not in a hang; `mine` and `all` are locals)
    for (int r = 0; r < world; ++r)
        if (all[r] != mine) {
            set_error("rank %d runs with exchange switches %#x, rank %d with %#x (REO_EXCHANGE_WAVES | REO_K1_STAMPS << 8 | REO_K1_WAVE << 9): every "
                      "rank must be started with the same environment", rank, mine, r, all[r]);
            comm_abort(c);
            return REO_EINVAL;
        }
    return REO_OK;
}

// Single-process form: one context per visible GPU behind one handle.  The returned (leader) context lives on
// device 0 and owns the others; every entry point of reo_hip.h may be called on it exactly as on a one-GPU
// context.  reo_build_pairs runs the transform and each device's share of the pair tiles on all devices at once
// (one host thread per device), the peers' packed table words are handed to the leader (multi_build_pairs below), and
// the passes run on the leader.
int32_t reo_create_multi(reo_ctx **out, int32_t n_gpus, uint64_t seed)
{
    if (!out) { set_error("out is null"); return REO_EINVAL; }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("no HIP device visible: libreo_hip has no CPU fallback");
        return REO_EHIP;
    }
    if (n_gpus <= 0) n_gpus = ndev;  // 0 = all visible
    // Test seam (REO_MULTI_ONE_DEVICE=1): the shard contexts share device 0 and their packs travel by device copies, so
    // the whole orchestration below -- one host thread per shard, pack, hand-over, unpack, propagation of the settings --
    // runs on a one-GPU box.  RCCL itself is not involved then.
    const char *seam = getenv("REO_MULTI_ONE_DEVICE");
    const bool one_device = seam && seam[0] == '1';
    if (n_gpus > ndev && !one_device) { set_error("%d GPUs requested, %d visible", n_gpus, ndev); return REO_EINVAL; }
    if (n_gpus > 64) { set_error("%d shards requested", n_gpus); return REO_EINVAL; }
    std::vector<reo_ctx *> all(n_gpus, nullptr);
    int32_t rc = REO_OK;
    for (int d = 0; d < n_gpus && !rc; ++d) rc = reo_create(&all[d], one_device ? 0 : d, seed);
    std::vector<ncclComm_t> comms(n_gpus, nullptr);
    if (!rc && n_gpus > 1 && !one_device) {
        std::vector<int> devs(n_gpus);
        for (int d = 0; d < n_gpus; ++d) devs[d] = d;
        const ncclResult_t r = ncclCommInitAll(comms.data(), n_gpus, devs.data());
        if (r != ncclSuccess) { set_error("ncclCommInitAll failed: %s", ncclGetErrorString(r)); rc = REO_ECOMM; }
    }
    if (rc) {
        for (ncclComm_t cm : comms) if (cm) (void)ncclCommDestroy(cm);
        for (reo_ctx *c : all) if (c) reo_destroy(c);
        return rc;
    }
    for (int d = 0; d < n_gpus; ++d) {
        all[d]->rank = d; all[d]->world = n_gpus;
        all[d]->comm = comms[d];
        all[d]->multi_one_device = one_device;
        all[d]->in_multi = true;
    }
    all[0]->peers.assign(all.begin() + 1, all.end());
    *out = all[0];
    return REO_OK;
}

}  // extern "C"

namespace reo {

// reo_build_pairs of a multi-GPU context: every device builds its tiles; the peers pack the forward words of their own
// units (kernels.hip, x_pack) and send the packs to the leader (grouped ncclSend / ncclRecv: 107 MB in all at 20 000 genes,
// against 205 MB per peer for a reduce of the whole tables), which unpacks them and derives the mirror words.
int32_t multi_build_pairs(reo_ctx *lead, int32_t k, int32_t (*build_local)(reo_ctx *, int32_t))
{
    std::vector<reo_ctx *> all{lead};
    all.insert(all.end(), lead->peers.begin(), lead->peers.end());
    const int world = static_cast<int>(all.size());
    // after comm_abort every communicator is gone while world stays above 1: answer before any work or NCCL group starts
    if (world > 1 && !lead->multi_one_device)
        for (reo_ctx *c : all)
            if (c->comm_dead || !c->comm) {
                set_error("the communicators of this multi-GPU context were aborted after an earlier failure: destroy it and create a new one");
                return REO_ECOMM;
            }
    std::vector<int32_t> rcs(all.size(), REO_OK);
    std::vector<std::string> errs(all.size());
    std::vector<std::thread> th;
    // every shard: its tiles, then (peers) its pack.  Everything that can fail on a shard happens before any transfer starts.
    auto shard = [&](size_t d) {
        reo_ctx *c = all[d];
        int32_t rc = build_local(c, k);
        if (!rc && d > 0) {
            const int64_t words = exchange_unit_words(c) * exchange_units_per_rank(c);
            if (!(rc = c->xsend.ensure(static_cast<size_t>(words))) && !(rc = launch_pack_units(c)))
                if (hipStreamSynchronize(c->stream) != hipSuccess) { set_error("pack failed on shard %zu", d); rc = REO_EHIP; }
        }
        rcs[d] = rc;
        if (rc) errs[d] = reo_last_error();
    };
    for (size_t d = 1; d < all.size(); ++d) th.emplace_back(shard, d);
    shard(0);
    for (auto &t : th) t.join();
    for (size_t d = 0; d < all.size(); ++d)
        if (rcs[d]) { if (d) set_error("device %zu: %s", d, errs[d].c_str()); return rcs[d]; }
    lead->table_complete = world <= 1;
    if (world > 1) {
        const int64_t words = exchange_unit_words(lead) * exchange_units_per_rank(lead);
        const size_t bytes = static_cast<size_t>(words) * sizeof(uint32_t);
        int32_t rc;
        REO_HIP_CHECK(hipSetDevice(lead->device));
        if ((rc = lead->xrecv.ensure(static_cast<size_t>(words) * world))) return rc;
        tic(lead, 6);
        if (lead->multi_one_device) {
            for (int d = 1; d < world; ++d)
                REO_HIP_CHECK(hipMemcpyAsync(lead->xrecv.p + static_cast<size_t>(d) * words, all[d]->xsend.p, bytes, hipMemcpyDeviceToDevice, lead->stream));
        } else {
            // the group is always closed, whatever a call inside it returned
            ncclResult_t first = ncclGroupStart();
            for (int d = 1; d < world && first == ncclSuccess; ++d) {
                if (hipSetDevice(all[d]->device) != hipSuccess) { first = ncclUnhandledCudaError; break; }
                first = ncclSend(all[d]->xsend.p, bytes, ncclUint8, 0, static_cast<ncclComm_t>(all[d]->comm), all[d]->stream);
                if (first != ncclSuccess) break;
                if (hipSetDevice(lead->device) != hipSuccess) { first = ncclUnhandledCudaError; break; }
                first = ncclRecv(lead->xrecv.p + static_cast<size_t>(d) * words, bytes, ncclUint8, d, static_cast<ncclComm_t>(lead->comm), lead->stream);
            }
            const ncclResult_t end = ncclGroupEnd();
            (void)hipSetDevice(lead->device);
            if (first != ncclSuccess || end != ncclSuccess) {
                set_error("hand-over of the class-table words failed: %s", ncclGetErrorString(first != ncclSuccess ? first : end));
                for (reo_ctx *c : all) comm_abort(c);
                return REO_ECOMM;
            }
        }
        if ((rc = launch_expand_units(lead))) return rc;
        toc(lead);
        for (reo_ctx *c : all) {
            REO_HIP_CHECK(hipSetDevice(c->device));
            if ((rc = comm_wait(c))) { for (reo_ctx *o : all) comm_abort(o); (void)hipSetDevice(lead->device); return rc; }
        }
        REO_HIP_CHECK(hipSetDevice(lead->device));
        lead->table_complete = true;
    }
    return REO_OK;
}

}  // namespace reo
