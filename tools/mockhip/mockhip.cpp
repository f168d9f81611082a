// A HOST-ONLY stand-in for libamdhip64 + librccl, for ONE purpose: running the host side of libreo_hip.so under AddressSanitizer on a
// machine without a GPU (tools/asan_host_mock.sh).  GPU ASan is not available on this project's pool, and the instrumented library
// cannot create a context on a GPU box (AMD's ASan runtime wants xnack) -- so the host logic (work lists, transform bookkeeping, the
// iteration driver's copies into caller memory, the exchange orchestration) was never run under a sanitizer with real shapes.  Here
// "device memory" is malloc'd host memory (so every hipMemcpy* is a memcpy that ASan checks at both ends), kernel launches do
// nothing, streams and events are tokens.  Results are garbage by construction: this checks memory discipline, not numerics.
// NOT part of the product: nothing under rankcompv3.jl_amd/ knows about it; the shipped library links the real runtime only.
//
// The one piece of behaviour: the iteration driver reads IterState from pinned memory after each batch of launches.  With
// MOCKHIP_N_ITER=n in the environment, a stream wait marks the most recent 100-byte-or-so pinned allocation (IterState) as
// "n passes executed, the last one on the sorting path", so reo_identify_degs leaves its loop and copies trace and result out.
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace {
struct Call { dim3 g, b; size_t shmem; hipStream_t st; };
thread_local std::vector<Call> g_calls;
std::mutex g_mu;
std::map<void *, size_t> g_pinned;   // live pinned allocations
long g_launches = 0, g_copies = 0;
int tok;                          // address used for opaque handles
void advance_state()
{
    // IterState lives in a small pinned block; since the library takes its pinned blocks from a cache (4 KB classes) the mock cannot
    // tell it from the two other small pinned blocks of a context (the transform's flags, the reference mask on its way to the
    // device), so every small pinned block gets the two words.  In the flags block that reads as "the data has ties" -- harmless.
    const char *e = getenv("MOCKHIP_N_ITER");
    if (!e) return;
    const int n = atoi(e);
    if (n <= 0) return;
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &kv : g_pinned) {
        if (kv.second < 64 || kv.second > 16384) continue;   // (a 4 KB request may be served by a cached block a little larger)
        int32_t *w = static_cast<int32_t *>(kv.first);
        w[1] = n; w[11] = 1;   // IterState.passes, .last_full (reo_internal.h)
    }
}
}  // namespace

extern "C" {

long mockhip_launches() { return g_launches; }
long mockhip_copies() { return g_copies; }

void **__hipRegisterFatBinary(const void *) { static void *h = &tok; return &h; }
void __hipRegisterFunction(void **, const void *, char *, const char *, unsigned, void *, void *, void *, void *, int *) {}
void __hipRegisterVar(void **, void *, char *, const char *, int, size_t, int, int) {}
void __hipUnregisterFatBinary(void **) {}
hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t shmem, hipStream_t st) { g_calls.push_back({g, b, shmem, st}); return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3 *g, dim3 *b, size_t *shmem, hipStream_t *st)
{
    if (g_calls.empty()) return hipErrorInvalidValue;
    const Call c = g_calls.back(); g_calls.pop_back();
    *g = c.g; *b = c.b; *shmem = c.shmem; *st = c.st;
    return hipSuccess;
}
hipError_t hipLaunchKernel(const void *, dim3 g, dim3 b, void **, size_t shmem, hipStream_t)
{
    // what a real launch would refuse
    if (g.x == 0 || g.y == 0 || g.z == 0 || b.x * b.y * b.z == 0 || b.x * b.y * b.z > 1024 || g.y > 65535 || g.z > 65535 || shmem > 160 * 1024) {
        fprintf(stderr, "mockhip: launch with grid (%u,%u,%u) block (%u,%u,%u) shmem %zu would fail\n", g.x, g.y, g.z, b.x, b.y, b.z, shmem);
        return hipErrorInvalidConfiguration;
    }
    ++g_launches;
    return hipSuccess;
}
hipError_t hipGetDeviceCount(int *n) { const char *e = getenv("MOCKHIP_NDEV"); *n = e ? atoi(e) : 1; return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
hipError_t hipDeviceGetPCIBusId(char *bdf, int len, int dev) { snprintf(bdf, static_cast<size_t>(len), "0000:%02x:00.0", 0xf0 + dev); return hipSuccess; }   // (no such device under /sys: the NUMA binding finds nothing to do)
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_t *p, int)
{
    memset(p, 0, sizeof *p);
    strcpy(p->gcnArchName, "gfx950:sramecc+:xnack-");
    p->multiProcessorCount = 256;
    return hipSuccess;
}
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t a, int) { *v = a == hipDeviceAttributeMultiprocessorCount ? 256 : 0; return hipSuccess; }
hipError_t hipMemGetInfo(size_t *f, size_t *t) { *f = size_t(200) << 30; *t = size_t(288) << 30; return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "mock error"; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
int hipGetStreamDeviceId(hipStream_t) { return 0; }

hipError_t hipMalloc(void **p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }   // (zeros: the copy of the loop state that REO_STATE_MIRROR=0 reads back must not carry a fault code)
hipError_t hipFree(void *p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned)
{
    *p = calloc(1, n ? n : 1);
    std::lock_guard<std::mutex> lk(g_mu);
    if (*p) g_pinned[*p] = n;
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void *p)
{
    { std::lock_guard<std::mutex> lk(g_mu); g_pinned.erase(p); }
    free(p);
    return hipSuccess;
}
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { ++g_copies; memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { ++g_copies; memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyWithStream(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { ++g_copies; memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t)
{
    ++g_copies;
    for (size_t r = 0; r < h; ++r) memmove(static_cast<char *>(d) + r * dp, static_cast<const char *>(s) + r * sp, w);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }

hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = reinterpret_cast<hipStream_t>(malloc(8)); return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = reinterpret_cast<hipStream_t>(malloc(8)); return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0; *hi = -1; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { advance_state(); return hipSuccess; }
hipError_t hipDeviceSynchronize() { advance_state(); return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { advance_state(); return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = reinterpret_cast<hipEvent_t>(malloc(8)); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = reinterpret_cast<hipEvent_t>(malloc(8)); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { advance_state(); return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

// ---- librccl: communicators are (rank, world) records; collectives copy in process ------------------------------------------------
struct MockComm { int rank, world; };
struct Pending { const void *send; void *recv; size_t bytes; int from, to; };
static std::vector<Pending> g_sends, g_recvs;
static int g_group = 0;
static void match()
{
    for (auto &r : g_recvs)
        for (auto &s : g_sends)
            if (s.send && s.from == r.from && s.to == r.to) { memmove(r.recv, s.send, r.bytes < s.bytes ? r.bytes : s.bytes); s.send = nullptr; break; }
    g_sends.clear(); g_recvs.clear();
}
typedef int ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) { memset(id, 0x5a, sizeof *id); return 0; }
ncclResult_t ncclCommInitRank(void **comm, int world, ncclUniqueId, int rank) { *comm = new MockComm{rank, world}; return 0; }
ncclResult_t ncclCommInitAll(void **comms, int n, const int *) { for (int i = 0; i < n; ++i) comms[i] = new MockComm{i, n}; return 0; }
ncclResult_t ncclCommDestroy(void *c) { delete static_cast<MockComm *>(c); return 0; }
ncclResult_t ncclCommAbort(void *c) { delete static_cast<MockComm *>(c); return 0; }
ncclResult_t ncclCommGetAsyncError(void *, ncclResult_t *e) { *e = 0; return 0; }
const char *ncclGetErrorString(ncclResult_t) { return "mock nccl"; }
ncclResult_t ncclGroupStart() { ++g_group; return 0; }
ncclResult_t ncclGroupEnd() { if (--g_group == 0) match(); return 0; }
static size_t tsize(int dt) { return (dt == 0 || dt == 1) ? 1 : (dt == 2 || dt == 3 || dt == 7) ? 4 : (dt == 6 || dt == 9) ? 2 : 8; }
ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, int dt, void *comm, hipStream_t)
{
    const MockComm *c = static_cast<MockComm *>(comm);
    const size_t b = count * tsize(dt);
    memmove(static_cast<char *>(recv) + b * c->rank, send, b);   // (a one-rank communicator is all this process can have: its own pack comes back)
    return 0;
}
ncclResult_t ncclSend(const void *buf, size_t count, int dt, int peer, void *comm, hipStream_t)
{
    g_sends.push_back({buf, nullptr, count * tsize(dt), static_cast<MockComm *>(comm)->rank, peer});
    if (!g_group) match();
    return 0;
}
ncclResult_t ncclRecv(void *buf, size_t count, int dt, int peer, void *comm, hipStream_t)
{
    g_recvs.push_back({nullptr, buf, count * tsize(dt), peer, static_cast<MockComm *>(comm)->rank});
    if (!g_group) match();
    return 0;
}

}  // extern "C"
