// In-library RCCL path for multi-GPU runs (include/reo_hip.h: reo_comm_unique_id, reo_comm_init_rank,
// reo_create_multi).  The reference has no multi-device path (/root/reference/src/RankCompV3.jl uses
// Threads.@threads only, :368,402); what is exchanged here is this build's own intermediate, the class table.
//
// Protocol: the pair tiles of the G x G triangle are dealt to the shards (launch_k1, `unit % world`); every shard
// writes the table words of its tiles (forward and mirror bits) into a zeroed table.  One process per GPU
// (reo_comm_init_rank): every shard packs the forward words of its own units (upper triangle: 107 MB in all at 20 000
// genes), ONE ncclAllGather per class table hands every pack to every shard, and each shard unpacks the others' words
// and derives their mirror words itself (api.hip, exchange_table; kernels.hip, x_pack / x_expand_*) -- a quarter of the
// bytes that an in-place sum of the whole 205 MB table moves.  One process, all GPUs (reo_create_multi): the tables,
// whose bits are disjoint, are summed onto the leader with one ncclReduce.  The iteration passes then run with no
// collective at all.
#include <rccl/rccl.h>

#include <cstring>
#include <thread>

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

int32_t comm_allgather(reo_ctx *c, const void *send, void *recv, int64_t bytes_per_rank)
{
    if (!c->comm) return 1;
    const ncclResult_t r = ncclAllGather(send, recv, static_cast<size_t>(bytes_per_rank), ncclUint8, static_cast<ncclComm_t>(c->comm), c->stream);
    if (r != ncclSuccess) { set_error("ncclAllGather failed: %s", ncclGetErrorString(r)); return REO_ECOMM; }
    return REO_OK;
}

void comm_release(reo_ctx *c)
{
    if (c->comm) (void)ncclCommDestroy(static_cast<ncclComm_t>(c->comm));
    c->comm = nullptr;
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
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclComm_t comm = nullptr;
    REO_NCCL_CHECK(ncclCommInitRank(&comm, world, u, rank));
    c->comm = comm;
    return REO_OK;
}

// Single-process form: one context per visible GPU behind one handle.  The returned (leader) context lives on
// device 0 and owns the others; every entry point of reo_hip.h may be called on it exactly as on a one-GPU
// context.  reo_build_pairs runs the transform and each device's share of the pair tiles on all devices at once
// (one host thread per device), the tables are summed onto the leader with one ncclReduce, and the passes run on
// the leader.
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
    if (n_gpus > ndev) { set_error("%d GPUs requested, %d visible", n_gpus, ndev); return REO_EINVAL; }
    std::vector<reo_ctx *> all(n_gpus, nullptr);
    int32_t rc = REO_OK;
    for (int d = 0; d < n_gpus && !rc; ++d) rc = reo_create(&all[d], d, seed);
    std::vector<ncclComm_t> comms(n_gpus, nullptr);
    if (!rc && n_gpus > 1) {
        std::vector<int> devs(n_gpus);
        for (int d = 0; d < n_gpus; ++d) devs[d] = d;
        const ncclResult_t r = ncclCommInitAll(comms.data(), n_gpus, devs.data());
        if (r != ncclSuccess) { set_error("ncclCommInitAll failed: %s", ncclGetErrorString(r)); rc = REO_ECOMM; }
    }
    if (rc) {
        for (reo_ctx *c : all) if (c) reo_destroy(c);
        return rc;
    }
    for (int d = 0; d < n_gpus; ++d) {
        all[d]->rank = d; all[d]->world = n_gpus;
        all[d]->comm = comms[d];
    }
    all[0]->peers.assign(all.begin() + 1, all.end());
    *out = all[0];
    return REO_OK;
}

}  // extern "C"

namespace reo {

// reo_build_pairs of a multi-GPU context: every device builds its tiles, then one reduce onto the leader.
int32_t multi_build_pairs(reo_ctx *lead, int32_t k, int32_t (*build_local)(reo_ctx *, int32_t))
{
    std::vector<reo_ctx *> all{lead};
    all.insert(all.end(), lead->peers.begin(), lead->peers.end());
    std::vector<int32_t> rcs(all.size(), REO_OK);
    std::vector<std::string> errs(all.size());
    std::vector<std::thread> th;
    for (size_t d = 1; d < all.size(); ++d)
        th.emplace_back([&, d] { rcs[d] = build_local(all[d], k); if (rcs[d]) errs[d] = reo_last_error(); });
    rcs[0] = build_local(lead, k);
    for (auto &t : th) t.join();
    for (size_t d = 0; d < all.size(); ++d)
        if (rcs[d]) { if (d) set_error("device %zu: %s", d, errs[d].c_str()); return rcs[d]; }
    const size_t count = static_cast<size_t>(lead->G) * kPlanes * lead->Wp;
    tic(lead, 6);
    REO_NCCL_CHECK(ncclGroupStart());
    for (reo_ctx *c : all) {
        REO_HIP_CHECK(hipSetDevice(c->device));
        REO_NCCL_CHECK(ncclReduce(c->table.p, c->table.p, count, ncclUint32, ncclSum, 0, static_cast<ncclComm_t>(c->comm), c->stream));
    }
    REO_NCCL_CHECK(ncclGroupEnd());
    REO_HIP_CHECK(hipSetDevice(lead->device));
    toc(lead);
    for (reo_ctx *c : all) {
        REO_HIP_CHECK(hipSetDevice(c->device));
        REO_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    REO_HIP_CHECK(hipSetDevice(lead->device));
    lead->table_complete = true;
    return REO_OK;
}

}  // namespace reo
