// C ABI of libreo_hip.so (include/reo_hip.h): context, host-side driver of the
// iteration loop of /root/reference/src/RankCompV3.jl:396-425, error plumbing.
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include <ctime>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "reo_internal.h"

#include <mutex>
#include <unordered_map>

namespace reo {

static thread_local std::string g_err;
thread_local bool tl_release_synced = false;

// ---- the block cache (reo_internal.h) -------------------------------------------------------------------------------------------
namespace {
struct Block { void *p; size_t bytes; int device; bool pinned; };
struct Pool {
    std::mutex mu;
    std::vector<Block> free_blocks;   // oldest first
    std::unordered_map<void *, size_t> live;   // blocks handed out -> their TRUE size (a cached block may be a quarter larger than the
                                               // request it serves: what comes back is booked with the size it has, so that `cached`
                                               // is what the cache holds and REO_DEVICE_CACHE_MB a real bound)
    size_t cached = 0, cap = size_t(16384) << 20;
    bool read_env = false;
};
Pool &pool() { static Pool *p = new Pool(); return *p; }   // (never destroyed: the HIP runtime may be gone when statics are)
size_t pool_round(size_t bytes) { return bytes <= 4096 ? 4096 : (bytes + 255) & ~size_t(255); }
hipError_t raw_alloc(void **p, size_t bytes, bool pinned) { return pinned ? hipHostMalloc(p, bytes) : hipMalloc(p, bytes); }
void raw_free(void *p, bool pinned) { if (pinned) (void)hipHostFree(p); else (void)hipFree(p); }
}  // namespace

namespace {
struct Handles {
    std::mutex mu;
    std::vector<std::pair<int, hipStream_t>> streams[2];   // (device, stream) by kind
    std::vector<std::pair<int, hipEvent_t>> events[2];
};
Handles &handles() { static Handles *h = new Handles(); return *h; }
}  // namespace

hipError_t handle_stream(hipStream_t *s, int kind)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        Handles &H = handles();
        std::lock_guard<std::mutex> lk(H.mu);
        auto &v = H.streams[kind];
        for (size_t i = v.size(); i-- > 0;)
            if (v[i].first == dev) { *s = v[i].second; v.erase(v.begin() + static_cast<long>(i)); return hipSuccess; }
    }
    if (kind == 0) return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    int lo_pri = 0, hi_pri = 0;
    const hipError_t e = hipDeviceGetStreamPriorityRange(&lo_pri, &hi_pri);
    return e != hipSuccess ? e : hipStreamCreateWithPriority(s, hipStreamNonBlocking, hi_pri);
}

hipError_t handle_event(hipEvent_t *ev, int kind)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        Handles &H = handles();
        std::lock_guard<std::mutex> lk(H.mu);
        auto &v = H.events[kind];
        for (size_t i = v.size(); i-- > 0;)
            if (v[i].first == dev) { *ev = v[i].second; v.erase(v.begin() + static_cast<long>(i)); return hipSuccess; }
    }
    return kind ? hipEventCreate(ev) : hipEventCreateWithFlags(ev, hipEventDisableTiming);
}

void release_stream(hipStream_t s, int kind)
{
    if (!s) return;
    int dev = 0;
    (void)hipGetDevice(&dev);
    Handles &H = handles();
    {
        std::lock_guard<std::mutex> lk(H.mu);
        // (kept only if idle and healthy: a stream that reports an error is destroyed, not handed to the next context)
        if (H.streams[kind].size() < 64 && pool().cap > 0 && hipStreamQuery(s) == hipSuccess) { H.streams[kind].emplace_back(dev, s); return; }
    }
    (void)hipStreamDestroy(s);
}

void release_event(hipEvent_t ev, int kind)
{
    if (!ev) return;
    int dev = 0;
    (void)hipGetDevice(&dev);
    Handles &H = handles();
    {
        std::lock_guard<std::mutex> lk(H.mu);
        if (H.events[kind].size() < 1024 && pool().cap > 0) { H.events[kind].emplace_back(dev, ev); return; }
    }
    (void)hipEventDestroy(ev);
}

hipError_t pool_alloc(void **out, size_t bytes, bool pinned)
{
    Pool &P = pool();
    bytes = pool_round(bytes);
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lk(P.mu);
        if (!P.read_env) {
            P.read_env = true;
            if (const char *e = getenv("REO_DEVICE_CACHE_MB")) P.cap = static_cast<size_t>(std::max(0L, atol(e))) << 20;
        }
        // best fit among blocks that waste at most a quarter (contexts of one problem size ask for the same sizes again: exact hits)
        int best = -1;
        for (int i = 0; i < static_cast<int>(P.free_blocks.size()); ++i) {
            const Block &b = P.free_blocks[i];
            if (b.pinned != pinned || (!pinned && b.device != dev) || b.bytes < bytes || b.bytes > bytes + bytes / 4 + 4096) continue;
            if (best < 0 || b.bytes < P.free_blocks[best].bytes) best = i;
        }
        if (best >= 0) {
            *out = P.free_blocks[best].p;
            P.cached -= P.free_blocks[best].bytes;
            P.live[*out] = P.free_blocks[best].bytes;
            P.free_blocks.erase(P.free_blocks.begin() + best);
            return hipSuccess;
        }
    }
    hipError_t e = raw_alloc(out, bytes, pinned);
    if (e == hipErrorOutOfMemory) {   // give the cache back and try once more
        (void)hipGetLastError();
        (void)reo_trim_memory();
        e = raw_alloc(out, bytes, pinned);
    }
    if (e == hipSuccess) { std::lock_guard<std::mutex> lk(P.mu); P.live[*out] = bytes; }
    return e;
}

void pool_free(void *p, size_t bytes, bool pinned)
{
    if (!p) return;
    Pool &P = pool();
    bytes = pool_round(bytes);
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::vector<Block> evict;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        const auto it = P.live.find(p);
        if (it != P.live.end()) { bytes = it->second; P.live.erase(it); }   // the block's own size, not the size of the request it served
        if (bytes > P.cap) { evict.push_back({p, bytes, dev, pinned}); }
        else {
            P.free_blocks.push_back({p, bytes, dev, pinned});
            P.cached += bytes;
            while (P.cached > P.cap && !P.free_blocks.empty()) {   // the oldest go first
                evict.push_back(P.free_blocks.front());
                P.cached -= P.free_blocks.front().bytes;
                P.free_blocks.erase(P.free_blocks.begin());
            }
        }
    }
    for (const Block &b : evict) {
        if (!b.pinned && b.device != dev) (void)hipSetDevice(b.device);
        raw_free(b.p, b.pinned);
        if (!b.pinned && b.device != dev) (void)hipSetDevice(dev);
    }
}

void set_error(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

void tic(reo_ctx *c, int slot)
{
    if (!c->profiling) return;
    StageTimer t;
    if (!c->pool.empty()) { t = c->pool.back(); c->pool.pop_back(); }
    else { (void)handle_event(&t.a, 1); (void)handle_event(&t.b, 1); }
    (void)hipEventRecord(t.a, c->stream);
    c->pending.emplace_back(slot, t);
    c->open.push_back(c->pending.size() - 1);
}

void toc(reo_ctx *c)
{
    if (!c->profiling || c->open.empty()) return;
    (void)hipEventRecord(c->pending[c->open.back()].second.b, c->stream);
    c->open.pop_back();
}

void collect_timings(reo_ctx *c)
{
    for (auto &pr : c->pending) {
        float ms = 0.f;
        (void)hipEventSynchronize(pr.second.b);
        if (hipEventElapsedTime(&ms, pr.second.a, pr.second.b) == hipSuccess) {
            c->t_ms[pr.first] += ms;
            if (pr.first == 2 && !c->k2_modes.empty()) {  // split the K2 stage time by what the pass did on the device
                if (c->k2_seen < c->k2_modes.size()) {
                    if (c->k2_modes[c->k2_seen]) { c->t_ms[8] += ms; c->t_ms[9] += 1.0; }
                    else c->t_ms[10] += ms;
                }
                ++c->k2_seen;
            }
        }
        c->pool.push_back(pr.second);
    }
    c->pending.clear();
    c->open.clear();
    c->k2_modes.clear();
    c->k2_seen = 0;
}

// get_major_reo_lower_count, src/RankCompV3.jl:81-92, with
// pvalue(Binomial(n, 1/2), x; tail = :both) = min(1, 2 min(ccdf(x-1), cdf(x)))
// (HypothesisTests, call sites :83,85).  The cdf is accumulated term by term in
// long double, pmf(t) = exp(log C(n,t) - n ln 2), so the scan over x is O(n).
static int32_t major_reo_lower_count(int32_t n, double thr)
{
    const long double ln2 = 0.693147180559945309417232121458L;
    long double logc = 0.0L, cdf_prev = 0.0L;  // cdf(x-1)
    int32_t first_above = -1;
    double pmin = 1.0;
    for (int x = 0; x <= n / 2; ++x) {
        if (x > 0) logc += std::log(static_cast<long double>(n - x + 1)) - std::log(static_cast<long double>(x));
        long double cdf = x >= n ? 1.0L : cdf_prev + std::exp(logc - static_cast<long double>(n) * ln2);
        if (cdf > 1.0L) cdf = 1.0L;
        const long double hi = 1.0L - cdf_prev;
        long double p = 2.0L * (cdf < hi ? cdf : hi);
        if (p > 1.0L) p = 1.0L;
        if (x == 0) {
            pmin = static_cast<double>(p);
            if (!(pmin < thr)) return n;  // the WARN branch (:87-90)
        }
        if (static_cast<double>(p) > thr) { first_above = x; break; }
        cdf_prev = cdf;
    }
    return first_above < 0 ? -1 : n - first_above + 1;  // -idx + 2 + n with idx = x + 1
}

// Host wait for the stream on the hot path.  REO_SPIN_WAIT=1 polls the stream instead of the blocking wait; measured on
// the bench step (tools/step_breakdown.py, round 3): 6.05 against 6.10 ms -- the host-side gaps of a step are not the
// wake-up, so the blocking wait stays the default (a polling thread is a bad neighbour in a threaded host program).
static hipError_t stream_wait(reo_ctx *c)
{
    if (!c->spin_wait) return hipStreamSynchronize(c->stream);
    for (unsigned spin = 0;; ++spin) {
        const hipError_t q = hipStreamQuery(c->stream);
        if (q != hipErrorNotReady) return q;
        if (spin > (1u << 22)) return hipStreamSynchronize(c->stream);  // a long wait after all: stop burning the core
    }
}

// REO_DEBUG_SEGV=1 (diagnostics; tools/fuzz_*.py set it): a native backtrace on SIGSEGV / SIGBUS / SIGABRT before the process dies --
// Python's faulthandler shows the Python frames only, and the one host crash of this project (tools/fuzz_gpu.py, rounds 1 and 4:
// profiles/faults/) came and went without a native stack.  Async-signal-safe calls only; the default action follows.
static struct sigaction g_old_action[3];   // SIGSEGV, SIGBUS, SIGABRT: what was installed before (Python's faulthandler, usually)

static void segv_backtrace(int sig)
{
    static const char head[] = "libreo_hip: fatal signal, native backtrace of the faulting thread:\n";
    (void)!write(2, head, sizeof head - 1);
    void *frames[64];
    const int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    sigaction(sig, &g_old_action[sig == SIGSEGV ? 0 : (sig == SIGBUS ? 1 : 2)], nullptr);   // then whoever was there before (or the default action)
    raise(sig);
}

static void install_segv_backtrace()
{
    static bool done = false;
    if (done) return;
    done = true;
    void *warm[2];
    (void)backtrace(warm, 2);   // (loads libgcc now: not from inside the handler)
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = segv_backtrace;
    sa.sa_flags = SA_NODEFER;
    sigaction(SIGSEGV, &sa, &g_old_action[0]);
    sigaction(SIGBUS, &sa, &g_old_action[1]);
    sigaction(SIGABRT, &sa, &g_old_action[2]);
}

// REO_DEBUG_PASSES: host wall clock of the calls that can block inside reo_identify_degs (microseconds, stderr)
static double wall_us()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

// The wait of a call that may be the first to see an asynchronous failure of the pair kernel (reo_build_pairs returns with
// it in flight on one GPU): the table cannot be trusted then, and a retry must rebuild it.
static int32_t wait_or_drop_table(reo_ctx *c)
{
    const hipError_t e = stream_wait(c);
    if (e == hipSuccess) return REO_OK;
    c->built_k = -1;
    c->table_complete = false;
    set_error("a kernel queued on this context failed: %s (the class table was dropped: call reo_build_pairs again)", hipGetErrorString(e));
    return REO_EHIP;
}

// Every call that enqueues an asynchronous copy from or into CALLER memory (or a local) holds one of these: whatever way the call
// ends -- a failing REO_HIP_CHECK between the enqueue and the wait included -- the stream has been waited for before the caller
// gets its arrays back (include/reo_hip.h: "no host pointer is retained after return").  A call that has done its own wait
// dismisses the guard; the extra wait of an error path costs nothing that matters.
struct DrainOnExit {
    reo_ctx *c;
    bool armed = true;
    explicit DrainOnExit(reo_ctx *ctx) : c(ctx) {}
    DrainOnExit(const DrainOnExit &) = delete;
    DrainOnExit &operator=(const DrainOnExit &) = delete;
    void dismiss() { armed = false; }
    ~DrainOnExit()
    {
        if (!armed || !c || !c->stream) return;
        (void)hipStreamSynchronize(c->stream);
    }
};

static int32_t use(reo_ctx *c)
{
    if (!c) { set_error("null context"); return REO_EINVAL; }
    REO_HIP_CHECK(hipSetDevice(c->device));
    return REO_OK;
}

static void invalidate(reo_ctx *c)
{
    c->transformed = false;
    c->built_k = -1;
    c->gc_valid = false;
    c->eager_k1 = false;
}

static int32_t set_matrix(reo_ctx *c, const void *X, int64_t G, int64_t S, int64_t ld, int dtype, bool on_device)
{
    int32_t rc = use(c);
    if (rc) return rc;
    if (!X) { set_error("matrix pointer is null"); return REO_EINVAL; }
    if (G < 2 || G > kMaxGenes || S < 2 || S > (1 << 20)) {
        set_error("matrix is %lld x %lld; G must be in [2, %d] and S in [2, 1048576]", (long long)G, (long long)S, kMaxGenes);
        return REO_EINVAL;
    }
    if (ld < G) { set_error("leading dimension %lld < G = %lld", (long long)ld, (long long)G); return REO_EINVAL; }
    invalidate(c);
    c->G = G; c->S = S; c->dtype = dtype;
    if (on_device) {
        c->dX = X; c->ld = ld;
        c->dX_owned.release();
    } else {
        if ((rc = c->dX_owned.ensure(static_cast<size_t>(G) * S * 8))) return rc;
        c->dX = c->dX_owned.p; c->ld = G;
        // Groups already set (the order the Julia shim and hotpath.py use): the upload is pipelined with the per-sample transform
        // and -- two groups, thresholds set, one GPU -- with the pair kernel's group-0 side (transform.hip, eager_upload).
        const bool lds_ranking = G <= 65535 && !(getenv("REO_TRANSFORM") && getenv("REO_TRANSFORM")[0] == 's');
        if (c->eager_mode > 0 && lds_ranking && static_cast<int64_t>(c->group_id.size()) == S && !c->in_multi) {
            const bool k1 = c->eager_mode > 1 && c->ngroups == 2 && c->thr_set && c->world <= 1 && !c->comm && !c->ag && !c->ar && c->k1_wave &&
                            S <= 65535 && !c->k1_stamps;
            c->Gp = static_cast<int>((c->G + kGenePad - 1) / kGenePad) * kGenePad;
            c->Wp = c->Gp / 32;
            if ((rc = c->table.ensure(static_cast<size_t>(c->G) * kPlanes * c->Wp))) return rc;
            const double w0 = wall_us();
            struct DrainUp {   // no exit leaves a copy from the caller's array in flight (the upload stream; c->stream has its own guard below)
                reo_ctx *c;
                ~DrainUp() { if (c->up) (void)hipStreamSynchronize(c->up); if (c->rk) (void)hipStreamSynchronize(c->rk); }
            } drain_up{c};
            DrainOnExit drain(c);
            rc = eager_upload(c, X, ld, k1);
            if (rc) { invalidate(c); return rc; }
            drain.dismiss();   // (the pair kernel may still be running, as after reo_build_pairs on one GPU: it reads device memory only)
            c->t_ms[11] += (wall_us() - w0) * 1e-3;
            if (c->eager_k1) { c->built_k = 0; c->table_complete = true; }
            return REO_OK;
        }
        // groups not known yet (or the pipelining switched off): the upload alone, chunked, Int64 narrowed
        struct DrainUp2 {
            reo_ctx *c;
            ~DrainUp2() { if (c->up) (void)hipStreamSynchronize(c->up); }
        } drain_up{c};
        DrainOnExit drain(c);
        const double w0 = wall_us();
        if ((rc = upload_columns(c, X, ld, G, S, c->dX_owned.p, dtype))) { invalidate(c); return rc; }
        drain.dismiss();   // (upload_columns has waited for the upload stream: the caller's array has been read)
        c->t_ms[11] += (wall_us() - w0) * 1e-3;
    }
    return REO_OK;
}

static int32_t ensure_transform(reo_ctx *c)
{
    if (c->dtype == 0) { set_error("no expression matrix set"); return REO_EINVAL; }
    if (c->group_id.empty()) { set_error("no groups set"); return REO_EINVAL; }
    if (static_cast<int64_t>(c->group_id.size()) != c->S) {
        // DimensionMismatch, src/RankCompV3.jl:355
        set_error("'data' and 'group' do not have compatible sizes (%lld columns, %zu group labels)",
                  (long long)c->S, c->group_id.size());
        return REO_EINVAL;
    }
    if (c->transformed) return REO_OK;
    c->Gp = static_cast<int>((c->G + kGenePad - 1) / kGenePad) * kGenePad;  // every lane's genes exist
    c->Wp = c->Gp / 32;
    c->table_prezeroed = false;
    int32_t rc = c->table.ensure(static_cast<size_t>(c->G) * kPlanes * c->Wp);  // (the transform queues its clearing, see transform.hip)
    if (rc) return rc;
    return run_transform(c);
}

static int32_t ensure_iter_buffers(reo_ctx *c)
{
    int32_t rc;
    const size_t G = c->G;
    for (int t = 0; t < 2; ++t) {
        if ((rc = c->refbits[t].ensure(c->Wp))) return rc;
        if ((rc = c->refbytes[t].ensure(c->Gp))) return rc;
    }
    if ((rc = c->raw.ensure(G * kRaw)) ||
        (rc = c->delta_list.ensure(2 * static_cast<size_t>(c->Gp))) || (rc = c->cont.ensure(G * 9)) || (rc = c->result.ensure(G * 15)) ||
        (rc = c->sorted_d.ensure((G + 63) / 64 * 64 + (G + 63) / 64)) || (rc = c->sorted_p.ensure(G)) || (rc = c->rank_s.ensure(G)) ||
        (rc = c->rank_a.ensure(G)) || (rc = c->scal.ensure(8)) || (rc = c->blockmin.ensure(std::max<size_t>(64, (G + 1023) / 1024))) ||
        (rc = c->state.ensure(1)) ||
        (rc = c->chunk_v.ensure(((G + kSortChunk - 1) / kSortChunk) * (kSortChunk + kSortChunk / 32))) ||
        (rc = c->chunk_i.ensure(((G + kSortChunk - 1) / kSortChunk) * kSortChunk)) || (rc = c->part.ensure(3 * (std::max<size_t>(65536, (G + kSortChunk - 1) / kSortChunk * kSortChunk) / 16 + 8))) ||
        (rc = c->cand.ensure(2 * 1024)) || (rc = c->gridbar.ensure(4)) || (rc = c->hist.ensure(3 * kHistParts * ((G + 32767) / 32768 * 32768))) || (rc = c->olist.ensure(2 * kOneStride)) || (rc = c->mrank.ensure(c->Gp)) || (rc = c->lstate.ensure(1)) || (rc = c->clist.ensure(2 * kListStride)) || (rc = c->scal.ensure(64)))
        return rc;
    if (!c->host_state) REO_HIP_CHECK(pool_alloc(reinterpret_cast<void **>(&c->host_state), sizeof(IterState), true));
    return REO_OK;
}

static int32_t upload_ref(reo_ctx *c, const uint8_t *ref, int slot, int32_t *nref)
{
    int32_t n = 0;
    for (int64_t i = 0; i < c->G; ++i) n += ref[i] != 0;
    *nref = n;
    REO_HIP_CHECK(hipMemsetAsync(c->refbytes[slot].p, 0, c->Gp, c->stream));
    REO_HIP_CHECK(hipMemcpyAsync(c->refbytes[slot].p, ref, c->G, hipMemcpyHostToDevice, c->stream));
    return launch_pack_ref(c, c->refbytes[slot].p, c->refbits[slot].p);
}

static int32_t init_state(reo_ctx *c, int32_t nref)
{
    IterState st;
    memset(&st, 0, sizeof st);
    st.nref = nref;
    st.nref_prev = nref;
    st.delta_cnt[0] = st.delta_cnt[1] = 0x7FFFFFFF;  // the first pass counts from scratch
    st.need_full = 1;
    st.raw_pass = -1;
    st.kstar = -1;
    *c->host_state = st;
    REO_HIP_CHECK(hipMemcpyAsync(c->state.p, c->host_state, sizeof st, hipMemcpyHostToDevice, c->stream));
    return REO_OK;
}

// Several shards (world > 1): every shard has built the class-table words of its own pair tiles and left the rest
// zero; the shards' bits are disjoint.  One exchange per class table -- an all-gather of the shards' packed forward
// words (in-library RCCL when a communicator is attached, else the caller's all-gather hook), or, with only a sum hook
// set, an in-place sum of the whole tables (disjoint bits: the sum IS the table); after it every shard holds the
// complete table and runs the iteration passes on its own, with no further collective.
static int32_t exchange_table(reo_ctx *c)
{
    if (c->x_pipelined) {  // launch_k1 has exchanged the table wave by wave, beside the pair kernel (kernels.hip)
        c->table_complete = true;
        return REO_OK;
    }
    c->table_complete = c->world <= 1;
    if (c->world <= 1 && !c->comm) return REO_OK;  // (a communicator of one rank still makes its call: the path stays testable on one GPU)
    int32_t rc;
    if (c->comm || c->ag) {
        // gather form: pack the forward words of this shard's units, all-gather the packs, unpack the others' words and
        // derive their mirror words here (kernels.hip, x_pack / x_expand_fwd / x_expand_mirror)
        const int64_t bytes = exchange_unit_words(c) * exchange_units_per_rank(c) * static_cast<int64_t>(sizeof(uint32_t));
        if ((rc = c->xsend.ensure(static_cast<size_t>(bytes / 4))) || (rc = c->xrecv.ensure(static_cast<size_t>(bytes / 4) * c->world))) return rc;
        tic(c, 6);  // the exchange stage: pack + collective + unpack
        if ((rc = launch_pack_units(c))) return rc;
        if (c->comm) {
            if ((rc = comm_allgather(c, c->xsend.p, c->xrecv.p, bytes)) < 0) return rc;
        } else {
            rc = c->ag(c->xsend.p, c->xrecv.p, bytes, c->stream, c->ag_user);  // stream-ordered, no host sync here
            if (rc) { set_error("all-gather hook failed with %d", rc); return REO_ECOMM; }
        }
        if ((rc = launch_expand_units(c))) return rc;
        toc(c);
    } else {
        if (!c->ar) return REO_OK;  // no exchange configured: the partial table can still be inspected (reo_get_codes)
        const int64_t count = static_cast<int64_t>(c->G) * kPlanes * c->Wp;
        tic(c, 6);
        rc = c->ar(c->table.p, count, c->stream, c->ar_user);  // stream-ordered, no host sync here
        toc(c);
        if (rc) { set_error("all-reduce hook failed with %d", rc); return REO_ECOMM; }
    }
    c->table_complete = true;
    return REO_OK;
}

// what the pass kernels report through IterState.fault (kernels.hip: kFaultBarrier, kFaultTallies)
static int32_t pass_fault(reo_ctx *c)
{
    const int32_t f = c->host_state ? c->host_state->fault : 0;
    if (!f) return REO_OK;
    if (f == kFaultTallies) {
        set_error("the iteration passes met tallies that no class table can produce (a gene pair in two states at once): the table in "
                  "this context is not what reo_build_pairs makes -- an exchange hook delivered wrong words and the scan of its "
                  "table was switched off (REO_CHECK_HOOK_TABLE=0)");
        c->built_k = -1;  // nothing may run on this table again
        c->table_complete = false;
    } else {
        set_error("the persistent iteration kernel gave up at a grid barrier (a workgroup did not arrive within its bound); "
                  "REO_LIGHT=1 runs the same passes as separate launches");
    }
    return REO_EHIP;
}

static int32_t need_complete_table(reo_ctx *c)
{
    if (c->built_k < 0) { set_error("no class table: call reo_build_pairs first"); return REO_EINVAL; }
    if (!c->table_complete) {
        set_error("shard %d of %d holds only its own part of the class table: attach a communicator (reo_comm_init_rank) or an "
                  "exchange hook (reo_set_allgather, reo_set_allreduce) before reo_build_pairs", c->rank, c->world);
        return REO_ECOMM;
    }
    return REO_OK;
}

}  // namespace reo

using namespace reo;

extern "C" {

int32_t reo_version(void) { return 100; }

int32_t reo_trim_memory(void)
{
    std::vector<Block> all;
    {
        Pool &P = pool();
        std::lock_guard<std::mutex> lk(P.mu);
        all.swap(P.free_blocks);
        P.cached = 0;
    }
    int dev = 0;
    const bool have_dev = hipGetDevice(&dev) == hipSuccess;
    for (const Block &b : all) {
        if (!b.pinned) (void)hipSetDevice(b.device);
        raw_free(b.p, b.pinned);
    }
    {
        Handles &H = handles();
        std::lock_guard<std::mutex> lk(H.mu);
        for (int k = 0; k < 2; ++k) {
            for (auto &st : H.streams[k]) { (void)hipSetDevice(st.first); (void)hipStreamDestroy(st.second); }
            for (auto &ev : H.events[k]) { (void)hipSetDevice(ev.first); (void)hipEventDestroy(ev.second); }
            H.streams[k].clear(); H.events[k].clear();
        }
    }
    if (have_dev) (void)hipSetDevice(dev);
    return REO_OK;
}

const char *reo_last_error(void) { return g_err.c_str(); }

int32_t reo_create(reo_ctx **out, int32_t device, uint64_t seed)
{
    if (!out) { set_error("out is null"); return REO_EINVAL; }
    *out = nullptr;
    if (getenv("REO_DEBUG_SEGV")) install_segv_backtrace();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("no HIP device visible: libreo_hip has no CPU fallback");
        return REO_EHIP;
    }
    if (device < 0) REO_HIP_CHECK(hipGetDevice(&device));
    if (device >= ndev) { set_error("device %d out of range (%d visible)", device, ndev); return REO_EINVAL; }
    hipDeviceProp_t prop;
    REO_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; this library carries gfx950 (MI355X) code objects only", device, prop.gcnArchName);
        return REO_EHIP;
    }
    REO_HIP_CHECK(hipSetDevice(device));
    reo_ctx *c = new (std::nothrow) reo_ctx();
    if (!c) { set_error("out of host memory"); return REO_ENOMEM; }
    c->device = device;
    c->seed = seed;
    if (const char *e = getenv("REO_K1_WAVE")) c->k1_wave = (e[0] != '0');
    if (const char *e = getenv("REO_K1_HALF")) c->k1_half = (e[0] != '0');
    if (const char *e = getenv("REO_K1_ORDER")) c->k1_order = std::max(0, std::min(2, atoi(e)));
    {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && n > 0) c->n_cus = n;
    }
    if (const char *e = getenv("REO_SPIN_WAIT")) c->spin_wait = (e[0] != '0');
    if (const char *e = getenv("REO_CHECK_HOOK_TABLE")) c->check_hook_table = (e[0] != '0');
    if (const char *e = getenv("REO_SHARE_GROUP_COUNTS")) c->share_counts = (e[0] != '0');
    if (const char *e = getenv("REO_LIGHT_BAND")) c->light_band = std::max(0, atoi(e));
    if (const char *e = getenv("REO_HIST_BELOW")) c->hist_below = std::max(0, atoi(e));
    if (const char *e = getenv("REO_CYCLE")) c->cycle_watch = atoi(e) != 0;
    if (const char *e = getenv("REO_LIGHT")) c->light_mode = e[0] == '0' ? 0 : (e[0] == '2' ? 2 : (e[0] == '3' ? 3 : 1));
    c->light_window = light_window(); c->light_min_g = light_min_genes();
    if (const char *e = getenv("REO_LIGHT_WINDOW")) c->light_window = std::max(1, std::min(31, atoi(e)));  // 2 W + 1 <= 64 window members
    if (const char *e = getenv("REO_LIGHT_MIN_G")) c->light_min_g = std::max(64, atoi(e));
    // (every switch is read here, once: no getenv on the paths a step takes)
    if (const char *e = getenv("REO_STATE_MIRROR")) c->state_mirror_wanted = (e[0] != '0');
    if (const char *e = getenv("REO_EXCHANGE_WAVES")) c->x_waves = std::max(1, std::min(8, atoi(e)));
    if (const char *e = getenv("REO_EAGER_UPLOAD")) c->eager_mode = std::max(0, std::min(2, atoi(e)));
    if (const char *e = getenv("REO_EAGER_CHUNK")) c->eager_chunk = std::max(1, atoi(e));
    if (const char *e = getenv("REO_EAGER_GATE")) c->eager_gate = atoi(e) != 0;
    if (const char *e = getenv("REO_EAGER_RANGES")) c->eager_ranges = std::max(1, std::min(6, atoi(e)));
    if (const char *e = getenv("REO_UPLOAD_THREADS")) c->upload_threads = std::max(0, std::min(64, atoi(e)));
    c->debug_passes = getenv("REO_DEBUG_PASSES") != nullptr;
    c->debug_stamps = getenv("REO_DEBUG_STAMPS") != nullptr;
    c->k1_stamps = getenv("REO_K1_STAMPS") != nullptr;
    hipError_t e = handle_stream(&c->stream, 0);
    if (e != hipSuccess) { delete c; set_error("hipStreamCreate failed: %s", hipGetErrorString(e)); return REO_EHIP; }
    // The light passes keep their BH-rank histogram as one partial per XCD, updated by atomics that stay in that XCD's L2.
    // That rests on two properties of the part, checked here once (a 20 us kernel): the XCC id register tells workgroups
    // of different XCDs apart, and such atomics from different workgroups of one XCD do not lose updates.  If the check
    // fails (or REO_XCC_LOCAL=0) the atomics are made coherent across the device instead -- slower, same results.
    int ok = 0;
    const char *xe = getenv("REO_XCC_LOCAL");
    if (!(xe && xe[0] == '0')) {
        const int32_t rc = xcc_selftest(c, &ok);
        if (rc) { reo_destroy(c); return rc; }
    }
    c->xcc_local = ok;
    *out = c;
    return REO_OK;
}

void reo_destroy(reo_ctx *c)
{
    if (!c) return;
    for (reo_ctx *p : c->peers) reo_destroy(p);
    c->peers.clear();
    (void)hipSetDevice(c->device);
    // Nothing of this context may still be queued or running when its buffers, its pinned mirrors (host_state, host_flags, host_ref:
    // kernels write the first two) and its communicator go: ALL of its streams are waited for -- the pair-kernel streams and the
    // exchange stream of the pipelined exchange as well (an error exit of launch_k1 joins them into c->stream, but a context can
    // also be destroyed right after a failure of the join itself).
    for (int q = 0; q < 2; ++q) if (c->k1s[q]) (void)hipStreamSynchronize(c->k1s[q]);
    if (c->xs) (void)hipStreamSynchronize(c->xs);
    if (c->up) (void)hipStreamSynchronize(c->up);
    if (c->rk) (void)hipStreamSynchronize(c->rk);
    (void)hipStreamSynchronize(c->stream);
    comm_release(c);
    collect_timings(c);
    tl_release_synced = true;   // every stream of this context has been waited for: its blocks go back to the cache without further waits
    for (auto &t : c->pool) { release_event(t.a, 1); release_event(t.b, 1); }
    c->dX_owned.release(); c->pos.release(); c->lo.release(); c->hi.release(); c->goff_dev.release();
    c->table.release();
    c->t_kin.release(); c->t_kout.release(); c->t_vin.release(); c->t_vout.release(); c->t_temp.release();
    c->t_order.release(); c->t_flags.release(); c->t_slots.release(); c->unit_map.release();
    for (auto &il : c->k1_wave_items) il.buf.release();
    for (int q = 0; q < 2; ++q) { release_stream(c->k1s[q], 0); release_event(c->ev_k1_join[q], 0); }
    release_stream(c->xs, 0);
    release_stream(c->up, 0);
    release_stream(c->rk, 1);
    for (auto &e : c->ev_rk) release_event(e, 0);
    for (auto &e : c->ev_up) release_event(e, 0);
    c->e_lists.release();
    for (int q = 0; q < 3; ++q) {
        if (c->stage_h[q]) pool_free(c->stage_h[q], c->stage_cap, true);
        c->stage_d[q].release();
        release_event(c->ev_stage[q], 0);
        release_event(c->ev_widen[q], 0);
    }
    release_event(c->ev_fork, 0);
    release_event(c->ev_x, 0);
    for (auto &e : c->ev_k1) release_event(e, 0);
    c->t_pos16.release(); c->t_lo16.release(); c->t_hi16.release(); c->gcounts.release();
    c->t_pos32.release(); c->t_lo32.release(); c->t_hi32.release(); c->t_vin32.release(); c->t_vout32.release();
    for (int t = 0; t < 2; ++t) { c->refbits[t].release(); c->refbytes[t].release(); }
    c->raw.release(); c->delta_list.release(); c->cont.release(); c->result.release(); c->sorted_d.release(); c->sorted_p.release();
    c->rank_s.release(); c->rank_a.release(); c->scal.release(); c->blockmin.release();
    c->state.release(); c->trace.release(); c->modes.release(); c->cand.release(); c->hist.release(); c->mrank.release(); c->lstate.release(); c->clist.release(); c->olist.release(); c->units_all.release(); c->xsend.release(); c->xrecv.release(); c->check_flag.release(); c->gridbar.release(); c->chunk_v.release(); c->chunk_i.release(); c->part.release();
    if (c->host_state) pool_free(c->host_state, sizeof(IterState), true);
    if (c->host_flags) pool_free(c->host_flags, 8 * sizeof(int32_t), true);
    release_event(c->ev_flags, 0);
    if (c->host_ref) pool_free(c->host_ref, c->host_ref_cap, true);
    release_stream(c->stream, 0);
    delete c;   // (its remaining DevBuf members are empty by now)
    tl_release_synced = false;
}

int32_t reo_set_shard(reo_ctx *c, int32_t rank, int32_t world)
{
    if (!c) { set_error("null context"); return REO_EINVAL; }
    if (world < 1 || rank < 0 || rank >= world) { set_error("bad shard %d of %d", rank, world); return REO_EINVAL; }
    if (!c->peers.empty()) { set_error("a multi-GPU context (reo_create_multi) shards by itself"); return REO_EINVAL; }
    c->rank = rank; c->world = world;
    c->built_k = -1;
    c->eager_k1 = false;
    c->gc_valid = false;
    c->table_complete = false;
    return REO_OK;
}

int32_t reo_set_allreduce(reo_ctx *c, reo_allreduce_fn fn, void *user)
{
    if (!c) { set_error("null context"); return REO_EINVAL; }
    c->ar = fn; c->ar_user = user;
    return REO_OK;
}

int32_t reo_set_allgather(reo_ctx *c, reo_allgather_fn fn, void *user)
{
    if (!c) { set_error("null context"); return REO_EINVAL; }
    c->ag = fn; c->ag_user = user;
    return REO_OK;
}

// multi-GPU context: every device gets its own copy of the matrix (host source: one upload each; device source:
// a peer copy from the leader's buffer)
static int32_t set_matrix_all(reo_ctx *c, const void *X, int64_t G, int64_t S, int64_t ld, int dtype, bool on_device)
{
    int32_t rc = set_matrix(c, X, G, S, ld, dtype, on_device);
    for (size_t d = 0; d < (c ? c->peers.size() : 0) && !rc; ++d) {
        reo_ctx *p = c->peers[d];
        if (!on_device) { rc = set_matrix(p, X, G, S, ld, dtype, false); continue; }
        if ((rc = use(p)) || (rc = p->dX_owned.ensure(static_cast<size_t>(G) * S * 8))) break;
        // one 2-D copy (the source may have a leading dimension); unified addressing finds the source device
        if (hipMemcpy2DAsync(p->dX_owned.p, static_cast<size_t>(G) * 8, X, static_cast<size_t>(ld) * 8, static_cast<size_t>(G) * 8, static_cast<size_t>(S),
                             hipMemcpyDefault, p->stream) != hipSuccess) {
            set_error("peer copy of the matrix to device %d failed", p->device); rc = REO_EHIP;
        }
        if (!rc && hipStreamSynchronize(p->stream) != hipSuccess) { set_error("peer copy failed"); rc = REO_EHIP; }
        if (!rc) { invalidate(p); p->G = G; p->S = S; p->dtype = dtype; p->dX = p->dX_owned.p; p->ld = G; }
    }
    if (c && !c->peers.empty()) (void)hipSetDevice(c->device);
    return rc;
}

int32_t reo_set_matrix_f64(reo_ctx *c, const double *X, int64_t G, int64_t S, int64_t ld) { return set_matrix_all(c, X, G, S, ld, 1, false); }
int32_t reo_set_matrix_i64(reo_ctx *c, const int64_t *X, int64_t G, int64_t S, int64_t ld) { return set_matrix_all(c, X, G, S, ld, 2, false); }
int32_t reo_set_matrix_dev_f64(reo_ctx *c, const void *dX, int64_t G, int64_t S, int64_t ld) { return set_matrix_all(c, dX, G, S, ld, 1, true); }
int32_t reo_set_matrix_dev_i64(reo_ctx *c, const void *dX, int64_t G, int64_t S, int64_t ld) { return set_matrix_all(c, dX, G, S, ld, 2, true); }

int32_t reo_set_groups(reo_ctx *c, const int32_t *group_id, int64_t len, int32_t ngroups)
{
    if (!c || !group_id) { set_error("null argument"); return REO_EINVAL; }
    if (ngroups < 2) {  // DimensionMismatch, src/RankCompV3.jl:356
        set_error("only %d level in 'group', at least 2 levels are needed", ngroups);
        return REO_EINVAL;
    }
    if (ngroups > len) { set_error("%d groups declared for %lld samples", ngroups, (long long)len); return REO_EINVAL; }  // (any number the samples allow, like :353)
    std::vector<int32_t> cnt(ngroups, 0);
    int next = 0;
    for (int64_t s = 0; s < len; ++s) {
        const int32_t g = group_id[s];
        if (g < 0 || g >= ngroups) { set_error("group id %d of sample %lld outside [0,%d)", g, (long long)s, ngroups); return REO_EINVAL; }
        if (cnt[g] == 0) {
            if (g != next) { set_error("group ids must be numbered in order of first appearance (unique(), :353)"); return REO_EINVAL; }
            ++next;
        }
        cnt[g]++;
    }
    if (next != ngroups) { set_error("%d groups declared but %d appear", ngroups, next); return REO_EINVAL; }
    c->group_id.assign(group_id, group_id + len);
    c->ngroups = ngroups;
    c->thr_set = false;
    invalidate(c);
    for (reo_ctx *p : c->peers) { p->group_id = c->group_id; p->ngroups = ngroups; p->thr_set = false; invalidate(p); }
    return REO_OK;
}

int32_t reo_threshold(int32_t n, double pval_reo) { return major_reo_lower_count(n, pval_reo); }

int32_t reo_compute_thresholds(reo_ctx *c, double pval_reo)
{
    if (!c || c->group_id.empty()) { set_error("groups must be set before thresholds"); return REO_EINVAL; }
    const int32_t S = static_cast<int32_t>(c->group_id.size());
    std::vector<int32_t> cnt(c->ngroups, 0);
    for (int32_t g : c->group_id) cnt[g]++;
    c->thr.assign(2 * c->ngroups, 0);
    std::vector<std::pair<int32_t, int32_t>> seen;  // (n, threshold) of this call: group sizes repeat (two balanced groups: one n, four uses)
    auto threshold_of = [&](int32_t n) {
        for (const auto &s : seen) if (s.first == n) return s.second;
        seen.emplace_back(n, major_reo_lower_count(n, pval_reo));
        return seen.back().second;
    };
    for (int k = 0; k < c->ngroups; ++k) {  // threshold = f.(hcat(gsi1,gsi2)'), :362
        c->thr[2 * k] = threshold_of(cnt[k]);
        c->thr[2 * k + 1] = threshold_of(S - cnt[k]);
        if (c->thr[2 * k] < 0 || c->thr[2 * k + 1] < 0) {
            set_error("no count in 0..n/2 has a two-sided binomial p above %g (the reference's findfirst returns nothing)", pval_reo);
            return REO_EINVAL;
        }
    }
    c->thr_set = true;
    c->built_k = -1;
    c->eager_k1 = false;
    for (reo_ctx *p : c->peers) { p->thr = c->thr; p->thr_set = true; p->built_k = -1; }
    return REO_OK;
}

int32_t reo_set_thresholds(reo_ctx *c, const int32_t *m)
{
    if (!c || !m || c->ngroups < 2) { set_error("groups must be set before thresholds"); return REO_EINVAL; }
    c->thr.assign(m, m + 2 * c->ngroups);
    c->thr_set = true;
    c->built_k = -1;
    c->eager_k1 = false;
    for (reo_ctx *p : c->peers) { p->thr = c->thr; p->thr_set = true; p->built_k = -1; }
    return REO_OK;
}

int32_t reo_get_thresholds(reo_ctx *c, int32_t *m)
{
    if (!c || !m || !c->thr_set) { set_error("thresholds not set"); return REO_EINVAL; }
    memcpy(m, c->thr.data(), sizeof(int32_t) * c->thr.size());
    return REO_OK;
}

// transform + this context's share of the pair tiles; no exchange
static int32_t enqueue_local(reo_ctx *c, int32_t k)
{
    int32_t rc = use(c);
    if (rc) return rc;
    if ((rc = ensure_transform(c))) return rc;
    if (!c->thr_set) { set_error("thresholds not set (reo_compute_thresholds)"); return REO_EINVAL; }
    if (k < 0 || k >= c->ngroups) { set_error("comparison %d outside [0,%d)", k, c->ngroups); return REO_EINVAL; }
    if ((rc = c->table.ensure(static_cast<size_t>(c->G) * kPlanes * c->Wp))) return rc;
    c->built_k = -1;
    return launch_k1(c, k);
}

// ... finished on return (shards: the pack and the exchange follow)
static int32_t build_local(reo_ctx *c, int32_t k)
{
    const int32_t rc = enqueue_local(c, k);
    if (rc) return rc;
    REO_HIP_CHECK(stream_wait(c));
    return REO_OK;
}

int32_t reo_build_pairs(reo_ctx *c, int32_t k)
{
    int32_t rc;
    if (c && !c->peers.empty()) {
        if ((rc = multi_build_pairs(c, k, build_local))) return rc;
    } else {
        if (c && c->comm_dead && c->world > 1) { set_error("the communicator of this context was aborted after an earlier failure: attach a new one (reo_comm_init_rank)"); return REO_ECOMM; }
        if (c && c->eager_k1 && k == 0 && c->built_k == 0 && c->transformed && !c->comm && c->world <= 1 && !c->ag && !c->ar) {
            // reo_set_matrix (host matrix, groups and thresholds known) has launched this comparison's pair kernel already
            c->eager_k1 = false;   // (once: a second reo_build_pairs rebuilds, as it always did)
            c->t_ms[5] += 1.0;
            return REO_OK;
        }
        if (c && !c->comm && c->world <= 1 && !c->ag && !c->ar) {
            // one GPU, nothing to exchange: the pair kernel is left running.  Every later call works on the same stream, so
            // reo_identify_degs queues its first passes behind it without a host round trip in between; an asynchronous
            // failure of the kernel surfaces at the next wait (REO_EHIP); the stage timers are collected by the next call
            // that reads them.
            if ((rc = enqueue_local(c, k))) return rc;
            c->table_complete = true;
            c->t_ms[5] += 1.0;
            c->built_k = k;
            return REO_OK;
        }
        // a rank that fails here must not leave its peers waiting inside the collective: it aborts its communicator
        // (a failing caller-supplied hook has to do the same with its own)
        if ((rc = build_local(c, k))) { if (c && c->comm && c->world > 1) comm_abort(c); return rc; }
        if ((rc = exchange_table(c))) { if (c->comm) comm_abort(c); return rc; }
        if ((rc = comm_wait(c))) return rc;
        if (c->table_complete && (c->ag || c->ar) && c->check_hook_table) {
            // a caller-supplied collective: the table it delivered must be a class table (one 35 us scan), else the
            // passes would run on tallies that break their invariants
            int bad = 0;
            if ((rc = launch_check_table(c, &bad))) return rc;
            if (bad) {
                c->table_complete = false;
                set_error("the exchange hook delivered an inconsistent class table (a pair in two states, or bits outside the table)");
                return REO_ECOMM;
            }
        }
    }
    c->t_ms[5] += 1.0;
    collect_timings(c);
    c->built_k = k;
    return REO_OK;
}

int32_t reo_pair_counts(reo_ctx *c, int64_t i0, int64_t i1, int64_t j0, int64_t j1, uint16_t *n_gt, uint16_t *n_eq)
{
    int32_t rc = use(c);
    if (rc) return rc;
    if ((rc = ensure_transform(c))) return rc;
    if (c->S > 65535) { set_error("reo_pair_counts returns 16-bit counts: not available with more than 65535 samples"); return REO_EINVAL; }
    if (!n_gt || !n_eq || i0 < 0 || j0 < 0 || i1 > c->G || j1 > c->G || i0 >= i1 || j0 >= j1) {
        set_error("bad pair block [%lld,%lld) x [%lld,%lld)", (long long)i0, (long long)i1, (long long)j0, (long long)j1);
        return REO_EINVAL;
    }
    const size_t n = static_cast<size_t>(i1 - i0) * (j1 - j0) * c->ngroups;
    DevBuf<uint16_t> dg, de;
    if ((rc = dg.ensure(n)) || (rc = de.ensure(n))) return rc;
    DrainOnExit drain(c);   // (declared after the buffers: the wait comes before their release)
    rc = launch_counts(c, i0, i1, j0, j1, dg.p, de.p);
    if (!rc) {
        hipError_t e = hipMemcpyAsync(n_gt, dg.p, n * 2, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(n_eq, de.p, n * 2, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { set_error("copy-back failed: %s", hipGetErrorString(e)); rc = REO_EHIP; }
    }
    dg.release(); de.release();
    return rc;
}

int32_t reo_get_codes(reo_ctx *c, int64_t i0, int64_t i1, int64_t j0, int64_t j1, uint8_t *code)
{
    int32_t rc = use(c);
    if (rc) return rc;
    if (c->built_k < 0) { set_error("no class table: call reo_build_pairs first"); return REO_EINVAL; }
    if (!code || i0 < 0 || j0 < 0 || i1 > c->G || j1 > c->G || i0 >= i1 || j0 >= j1) {
        set_error("bad pair block"); return REO_EINVAL;
    }
    const size_t n = static_cast<size_t>(i1 - i0) * (j1 - j0);
    DevBuf<uint8_t> d;
    if ((rc = d.ensure(n))) return rc;
    DrainOnExit drain(c);
    rc = launch_decode(c, i0, i1, j0, j1, d.p);
    if (!rc) {
        hipError_t e = hipMemcpyAsync(code, d.p, n, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { set_error("copy-back failed: %s", hipGetErrorString(e)); rc = REO_EHIP; }
    }
    d.release();
    return rc;
}

int32_t reo_tally(reo_ctx *c, const uint8_t *ref_mask, int32_t *cont)
{
    int32_t rc = use(c);
    if (rc) return rc;
    if ((rc = need_complete_table(c))) return rc;
    if (!ref_mask || !cont) { set_error("null argument"); return REO_EINVAL; }
    if ((rc = ensure_iter_buffers(c))) return rc;
    DrainOnExit drain(c);   // ref_mask in, cont out
    int32_t nref = 0;
    if ((rc = upload_ref(c, ref_mask, 0, &nref))) return rc;
    if ((rc = init_state(c, nref))) return rc;
    if ((rc = launch_tally(c, nref))) return rc;
    REO_HIP_CHECK(hipMemcpyAsync(cont, c->cont.p, sizeof(int32_t) * 9 * c->G, hipMemcpyDeviceToHost, c->stream));
    REO_HIP_CHECK(hipStreamSynchronize(c->stream));
    drain.dismiss();
    collect_timings(c);
    return REO_OK;
}

int32_t reo_identify_degs(reo_ctx *c, const uint8_t *ref0, double pval_deg, double padj_deg, int32_t n_iter,
                          int32_t n_conv, double *result, int32_t *iters_run, int32_t *trace)
{
    int32_t rc = use(c);
    if (rc) return rc;
    if ((rc = need_complete_table(c))) return rc;
    if (!ref0 || !result) { set_error("null argument"); return REO_EINVAL; }
    if ((rc = ensure_iter_buffers(c))) return rc;
    const int64_t G = c->G;
    // slice bounds of :411, round(Int, x) = round-half-even
    const int64_t a = static_cast<int64_t>(std::nearbyint(static_cast<double>(G) * 0.05));
    const int64_t b = static_cast<int64_t>(std::nearbyint(static_cast<double>(G) * 0.95));
    if (n_iter > 0 && (a < 1 || b > G || b - a + 1 < 2)) {
        set_error("G = %lld: the slice %lld:%lld of the sorted delta1 vector is out of bounds or too short "
                  "(the reference throws at src/RankCompV3.jl:411-413)", (long long)G, (long long)a, (long long)b);
        return REO_EINVAL;
    }
    if ((rc = c->trace.ensure(2 * static_cast<size_t>(n_iter > 0 ? n_iter : 1)))) return rc;
    if ((rc = c->modes.ensure(static_cast<size_t>(n_iter > 0 ? n_iter : 1) + 64))) return rc;
    // flush K2 timers of earlier calls (reo_tally): the ones below are matched to launches by order.  Timers of the pair
    // stage stay pending -- waiting for them here would put a host round trip between the pair kernel and the first pass.
    for (const auto &pr : c->pending)
        if (pr.first == 2) { collect_timings(c); break; }
    // mask and initial loop state through pinned host memory, and zeros(r,15) (:398), the K2 mode log and the histograms
    // cleared, all in one launch (kernels.hip, k_iter_init)
    if (c->host_ref_cap < static_cast<size_t>(G)) {
        if (c->host_ref) pool_free(c->host_ref, c->host_ref_cap, true);   // (the last call that read it has been waited for)
        c->host_ref = nullptr; c->host_ref_cap = 0;
        REO_HIP_CHECK(pool_alloc(reinterpret_cast<void **>(&c->host_ref), static_cast<size_t>(c->Gp), true));
        c->host_ref_cap = static_cast<size_t>(c->Gp);
    }
    int32_t nref = 0;
    for (int64_t i = 0; i < G; ++i) { c->host_ref[i] = ref0[i]; nref += ref0[i] != 0; }
    {
        IterState st;
        memset(&st, 0, sizeof st);
        st.nref = nref;
        st.nref_prev = nref;
        st.delta_cnt[0] = st.delta_cnt[1] = 0x7FFFFFFF;  // the first pass counts from scratch
        st.need_full = 1;
        st.raw_pass = -1;
        st.kstar = -1;
        st.cyc_period = -1;  // (set to "watching" below, once the form of the light passes is known)
        *c->host_state = st;
    }
    // cycle watch (kernels.hip, kl_head): the two-launch light passes notice when the reference set of a pass equals that of an
    // earlier pass; the loop below then skips whole periods.  Off: REO_CYCLE=0, the other forms of light pass, sorting passes only.
    const bool watch = c->cycle_watch && c->light_mode != 0 && (c->light_mode == 1 || G > 65535) && G >= c->light_min_g && n_iter > 0;
    if (watch) {
        if ((rc = c->snap.ensure(2 * static_cast<size_t>(c->Gp)))) return rc;
        c->host_state->cyc_period = 0;
    }
    c->it_cycle_period = 0; c->it_cycle_at = 0; c->it_cycle_skipped = 0;
    // from here on launches are queued that write the pinned mirrors, and -- at the end -- copies into the caller's trace and result:
    // no exit of this call leaves any of them in flight
    DrainOnExit drain(c);
    if ((rc = launch_iter_init(c, c->host_ref, c->host_state))) return rc;
    c->it_pval_deg = pval_deg; c->it_padj_deg = padj_deg; c->it_n_iter = n_iter; c->it_n_conv = n_conv;
    c->it_a0 = static_cast<int>(a - 1); c->it_b0 = static_cast<int>(b - 1);
    c->k2_idx = 0;
    // The loop control of :400,418-424 lives in device memory (IterState): passes are enqueued in batches, every
    // kernel looks at the state and returns at once when its pass is not wanted (convergence, n_iter reached, the
    // other kind of pass is due), and the host reads the state once per batch.  Two kinds of pass (kernels.hip):
    // the sorting path -- needed for the first pass, whenever the reference set changed by more genes than a tally
    // update takes, and when a quantile window lost its order statistic -- and the light path.  A batch = two
    // sorting passes, or a run of light passes; small problems sort every pass.
    // light passes above 65 535 genes: the two-launch form only (its lists and block moments are sized for 1 024 workgroups of 256
    // genes; the persistent form needs every workgroup resident, the one-launch form packs 16-bit state)
    const int light_form = (G > 65535 && c->light_mode != 0) ? 1 : c->light_mode;
    c->it_light_form = light_form;
    c->state_mirror = light_form != 2 && c->state_mirror_wanted;
    bool small = G < c->light_min_g || c->light_mode == 0;
    c->it_no_light = false;  // (the device must know when nobody enqueues light passes: a sorting pass that left need_full clear would wait for them)
    int passes = 0, seen_need_full = 1, light_batches = 0, idle_light = 0, idle_any = 0;  // idle_light: light batches in a row that completed no pass
    while (n_iter > 0) {  // :400
        const int remaining = n_iter - passes;
        const double w_loop = c->debug_passes ? wall_us() : 0.0;
        // the state read after the last batch says which kind of pass is due: sorting launches are enqueued only then
        // (they would return at once otherwise: seven idle launches each), light launches only behind a pass that left
        // windows.  The first light batch is short, because a run that converges does so within a few passes and every
        // launch after that is idle; later ones are longer.
        // (the first batch: four -- the reference set settles over the first three or four passes, and a sorting launch that is not
        //  wanted returns at once for the price of a batch boundary)
        const int nfull = small ? std::min(8, remaining) : (seen_need_full ? std::min(passes == 0 ? 4 : 2, remaining) : 0);
        const int nlight = (small || seen_need_full) ? 0 : std::min(light_batches == 0 ? 32 : kLightBatch, remaining);
        if (nlight > 0) ++light_batches;
        tic(c, 3);
        for (int t = 0; t < nfull; ++t)
            if ((rc = launch_full_pass(c, false))) return rc;
        if (nlight > 0 && light_form == 2) {
            if ((rc = launch_light_persistent(c))) return rc;  // runs light passes until the state stops them
        } else if (nlight > 0) {
            if ((rc = launch_light_batch(c, nlight))) return rc;
        }
        toc(c);
        if (!c->state_mirror) REO_HIP_CHECK(hipMemcpyAsync(c->host_state, c->state.p, sizeof(IterState), hipMemcpyDeviceToHost, c->stream));
        const double w0 = c->debug_passes ? wall_us() : 0.0;
        if ((rc = wait_or_drop_table(c))) return rc;
        if (c->debug_passes) fprintf(stderr, "  waited %.0f us for the batch (enqueueing it took %.0f us)\n", wall_us() - w0, w0 - w_loop);
        if (c->debug_passes && nlight > 0 && light_form == 3) {  // which check of the one-launch form ended the batch
            static LightState hs;  // (half a megabyte: not on the stack)
            if (hipMemcpy(&hs, c->lstate.p, sizeof hs, hipMemcpyDeviceToHost) == hipSuccess)
                for (int q = 1; q <= nlight; ++q)
                    if (hs.slot[q].pad0[0] || q <= 3)
                        fprintf(stderr, "  launch %d: why %d (1 window, 2 se left the bracket, 4 the histogram's cut left the band, 8 a list overflowed, 16 too many listed, 32 cut below the band), "
                                "cut of the m_lo histogram %d, listed %d, surely inside %d, cut %d (band around %d), eta %.3g, se %.10g\n", q, hs.slot[q].pad0[0], hs.slot[q].pad0[1],
                                hs.slot[q].pad0[2], hs.slot[q].pad0[3], hs.slot[q].pad0[4], hs.slot[q - 1].rec.kstar, hs.slot[q - 1].eta, hs.slot[q].se_base);
        }
        if (c->debug_passes)
            fprintf(stderr, "batch: %d sorting + %d light launches, passes %d -> %d, need_full %d, done %d, last_full %d, changed genes in front of the next pass %d\n", nfull, nlight,
                    passes, c->host_state->passes, c->host_state->need_full, c->host_state->done, c->host_state->last_full,
                    c->host_state->delta_cnt[c->host_state->passes & 1]);
        if ((rc = pass_fault(c))) return rc;
        idle_any = c->host_state->passes == passes ? idle_any + 1 : 0;
        if (idle_any >= 4) {  // no kind of batch completes a pass any more: an error of the loop control, never a reason to spin
            set_error("the iteration made no progress in four batches of launches (passes %d, need_full %d): loop control fault", passes, c->host_state->need_full);
            return REO_EHIP;
        }
        if (nlight > 0) idle_light = c->host_state->passes == passes ? idle_light + 1 : 0;
        if (idle_light >= 2 && !small) {
            // two light batches in a row completed no pass (each ended by handing its first pass to the sorting path, which
            // leaves need_full set): sorting passes for the rest of the call
            small = true;
            c->it_no_light = true;
        }
        if (c->host_state->passes < passes || c->host_state->passes > n_iter) {
            // (the device never counts past n_iter -- every pass kernel checks -- but this number sizes a copy into the CALLER's trace
            //  array below: it is not taken on trust)
            set_error("the iteration's pass counter reads %d after %d of %d passes: loop control fault", c->host_state->passes, passes, n_iter);
            return REO_EHIP;
        }
        passes = c->host_state->passes;
        seen_need_full = c->host_state->need_full;
        if (c->host_state->done || passes >= n_iter) break;  // :419-422
        if (c->host_state->cyc_period > 0) {
            // The reference set in front of pass `passes` equals the one p passes earlier, and a pass is a function of its reference
            // set alone (:401-424: the table is fixed, the convergence test compares consecutive sets): from here on the loop repeats
            // its last p passes for ever -- it has not converged within them, so it never will.  Whole periods are therefore skipped:
            // the pass counter moves on, the trace of the skipped passes is the last period's, and the passes that remain run as usual.
            // (An even number of passes: mask, delta list and tally state are double-buffered by the parity of the pass index.  At
            //  least one pass is left to execute: the batch stopped with the tallies of pass `passes` already made, and the replay
            //  below wants the loop to end the way it always does, with the tallies of the LAST executed pass in place.)
            const int p = c->host_state->cyc_period, P = (p & 1) ? 2 * p : p;
            const int skip = (n_iter - passes - 1) / P * P;
            if ((rc = launch_cycle_skip(c, skip, p))) return rc;   // (also ends the watch)
            c->it_cycle_period = p; c->it_cycle_at = passes; c->it_cycle_skipped = skip;
            if (c->debug_passes) fprintf(stderr, "cycle: the reference set in front of pass %d equals that of pass %d; %d passes skipped\n", passes, passes - p, skip);
            passes += skip;
        }
    }
    if (passes > 0 && !c->host_state->last_full) {
        // the loop ended on a light pass: delta2, se, z1, the tallies and padj of that pass come from the sorting path
        tic(c, 3);
        if ((rc = launch_full_pass(c, true))) return rc;
        toc(c);
    }
    if (trace && passes > 0)
        REO_HIP_CHECK(hipMemcpyAsync(trace, c->trace.p, sizeof(int32_t) * 2 * passes, hipMemcpyDeviceToHost, c->stream));
    const int nk2 = std::min<int>(c->k2_idx, static_cast<int>(c->modes.n));
    if (c->profiling && nk2 > 0) {
        // (sized before the copy is enqueued and not touched again until collect_timings, behind the wait below; an error exit in
        //  between waits through `drain`)
        c->k2_modes.assign(c->k2_idx, 0);
        REO_HIP_CHECK(hipMemcpyAsync(c->k2_modes.data(), c->modes.p, sizeof(int32_t) * nk2, hipMemcpyDeviceToHost, c->stream));
    }
    if (c->debug_stamps) {  // diagnostic builds (-DREO_STAMPS): marks of workgroup 0 in the last light launches, 10 ns units
        unsigned long long st[24];
        REO_HIP_CHECK(hipMemcpy(st, c->scal.p + 32, sizeof st, hipMemcpyDeviceToHost));
        if (light_form == 3) {
            fprintf(stderr, "stamps kl_one (inputs back, se, cut, lists + mask step, changed rows, delta1, sums + windows, end):");
            for (int k = 0; k <= 7; ++k) fprintf(stderr, " %lld", (long long)(st[k] - st[8]));
        } else if (light_form == 2) {
            fprintf(stderr, "stamps kl_persist, one pass (phase 1, barrier, phase 2, barrier, loads asked, cut, mask step):");
            for (int k = 1; k <= 7; ++k) fprintf(stderr, " %lld", (long long)(st[k] - st[0]));
        } else {
            fprintf(stderr, "stamps kl_head (inputs back, cut, mask step, changed rows, delta1, window flags, block sums, end):");
            for (int k = 0; k <= 7; ++k) fprintf(stderr, " %lld", (long long)(st[k] - st[8]));
            fprintf(stderr, "\nstamps kl_rank (inputs back, selection + se, p + BH rank, end):");
            for (int k = 20; k <= 23; ++k) fprintf(stderr, " %lld", (long long)(st[k] - st[19]));
        }
        fprintf(stderr, "  (x 10 ns)\n");
    }
    if (iters_run) *iters_run = passes;
    const double w_copy = c->debug_passes ? wall_us() : 0.0;
    REO_HIP_CHECK(hipMemcpyAsync(result, c->result.p, sizeof(double) * 15 * G, hipMemcpyDeviceToHost, c->stream));
    const double w_copied = c->debug_passes ? wall_us() : 0.0;
    if ((rc = wait_or_drop_table(c))) return rc;
    drain.dismiss();
    if (c->debug_passes) fprintf(stderr, "  result copy: enqueued in %.0f us, waited %.0f us\n", w_copied - w_copy, wall_us() - w_copied);
    collect_timings(c);
    return pass_fault(c);  // (the replay, too, derives every gene's tallies)
}

int32_t reo_mccullagh(reo_ctx *c, const int32_t *cont, int64_t n, double *out)
{
    int32_t rc = use(c);
    if (rc) return rc;
    if (!cont || !out || n < 1) { set_error("bad argument"); return REO_EINVAL; }
    DevBuf<int32_t> dc;
    DevBuf<double> dout;
    if ((rc = dc.ensure(n * 9)) || (rc = dout.ensure(n * 5))) return rc;
    DrainOnExit drain(c);
    hipError_t e = hipMemcpyAsync(dc.p, cont, sizeof(int32_t) * 9 * n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        rc = launch_mccullagh(c, dc.p, n, dout.p);
        if (!rc) {
            e = hipMemcpyAsync(out, dout.p, sizeof(double) * 5 * n, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        }
    }
    if (e != hipSuccess) { set_error("HIP failure: %s", hipGetErrorString(e)); rc = REO_EHIP; }
    dc.release(); dout.release();
    return rc;
}

int32_t reo_set_profiling(reo_ctx *c, int32_t on)
{
    if (!c) { set_error("null context"); return REO_EINVAL; }
    c->profiling = on != 0;
    for (reo_ctx *p : c->peers) p->profiling = c->profiling;
    return REO_OK;
}

int32_t reo_reset_timings(reo_ctx *c)
{
    if (!c) { set_error("null context"); return REO_EINVAL; }
    collect_timings(c);
    for (double &v : c->t_ms) v = 0.0;
    return REO_OK;
}

int32_t reo_get_timings(reo_ctx *c, double *ms, int32_t n)
{
    if (!c || !ms) { set_error("null argument"); return REO_EINVAL; }
    collect_timings(c);
    for (int i = 0; i < n && i < REO_NTIMINGS; ++i) ms[i] = c->t_ms[i];
    return REO_OK;
}

int32_t reo_get_info(reo_ctx *c, int64_t *info, int32_t n)
{
    if (!c || !info) { set_error("null argument"); return REO_EINVAL; }
    const int64_t v[21] = {c->G, c->S, c->Gp, static_cast<int64_t>(c->table.n * sizeof(uint32_t)), c->has_ties,
                           c->tiles_owned, c->tiles_total, kTileI, c->k1_cj, c->k1_q, kUnitH,
                           c->goff32.empty() ? 0 : c->goff32.back(), c->last_k1_shared,
                           static_cast<int64_t>(c->gcounts.n * sizeof(uint16_t)), c->transform_in_lds, c->xcc_local,
                           c->it_cycle_period, c->it_cycle_at, c->it_cycle_skipped, c->narrowed_bytes, c->eager_range_launches};
    for (int i = 0; i < n && i < 21; ++i) info[i] = v[i];
    return REO_OK;
}

}  // extern "C"
