// Pseudo-bulk front end: pseudobulk_group of /root/reference/src/RankCompV3.jl:56-67 (call site
// :608-612) as a segmented column sum on the GPU.  The caller supplies the shuffled cell order
// and the chunk boundaries (the reference's sample(1:c, c) + Iterators.partition, :62), so that
// every output profile sums its cells in exactly that order: integer data is exact, Float64
// data reproduces the reference's left-to-right row sums (:63) bit for bit.
//
//   dense  X f64/i64 [G x C] column-major (what the reference holds after CSV.read)
//   CSC    colptr i64 [C+1], rowidx i32 [nnz], val f64/i64 [nnz]  (this build's own container for
//          sparse single-cell counts; the reference has no sparse input format)
// Both are HBM-bound streams: every needed input byte is read once.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <type_traits>

#include "reo_internal.h"

namespace reo {

namespace {

// dense: one thread per gene row, one workgroup column per output profile; the threads of a wave
// read consecutive rows of the same cell column (coalesced), cells in the given order.
template <class T>
__global__ __launch_bounds__(256) void pb_dense(const T *__restrict__ X, int64_t ld, int G,
                                                const int32_t *__restrict__ order,
                                                const int32_t *__restrict__ chunk_ptr, T *__restrict__ out)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    const int o = blockIdx.y;
    if (g >= G) return;
    T acc = 0;
    for (int t = chunk_ptr[o]; t < chunk_ptr[o + 1]; ++t) acc += X[static_cast<int64_t>(order[t]) * ld + g];
    out[static_cast<int64_t>(o) * G + g] = acc;
}

// CSC: one workgroup (1024 threads) per (output profile, tile of kPbRows gene rows); a dense accumulator
// for the tile lives in LDS; the cells of the profile are visited in order, a barrier between cells keeps
// the per-row summation order (Float64 sums reproduce the reference's left-to-right order bit for bit);
// inside one cell every row occurs at most once, so plain LDS read-modify-writes by distinct threads need
// no atomics.  The kernel is latency-bound (a cell has ~1000 entries), so the entries of the next kPbDepth
// cells are fetched into registers while the current one is accumulated.
constexpr int kPbRows = 16384;
constexpr int kPbThreads = 1024;
constexpr int kPbPer = 2;    // prefetched entries per thread and cell; longer cells finish through the slow loop
constexpr int kPbDepth = 4;  // cells in flight (measured: 2 -> 0.48 ms, 4 -> 0.30 ms, 8 -> 0.46 ms at config 5)

template <class T>
struct PbCell {
    int64_t e0, e1;
    int r[kPbPer];
    T v[kPbPer];
};

template <class T>
__device__ __forceinline__ void pb_fetch(PbCell<T> &cell, const int64_t *__restrict__ colptr, const int32_t *__restrict__ rowidx,
                                         const T *__restrict__ val, const int32_t *__restrict__ order, int t, int t_end)
{
    cell.e0 = cell.e1 = 0;
#pragma unroll
    for (int i = 0; i < kPbPer; ++i) cell.r[i] = -1;
    if (t >= t_end) return;
    const int c = order[t];
    cell.e0 = colptr[c]; cell.e1 = colptr[c + 1];
#pragma unroll
    for (int i = 0; i < kPbPer; ++i) {
        const int64_t e = cell.e0 + static_cast<int64_t>(i) * kPbThreads + threadIdx.x;
        if (e < cell.e1) { cell.r[i] = rowidx[e]; cell.v[i] = val[e]; }
    }
}

template <class T>
__global__ __launch_bounds__(kPbThreads) void pb_csc(const int64_t *__restrict__ colptr, const int32_t *__restrict__ rowidx,
                                                     const T *__restrict__ val, int G, const int32_t *__restrict__ order,
                                                     const int32_t *__restrict__ chunk_ptr, T *__restrict__ out)
{
    __shared__ T acc[kPbRows];
    const int o = blockIdx.x;
    const int r0 = blockIdx.y * kPbRows, r1 = min(G, r0 + kPbRows);
    for (int t = threadIdx.x; t < kPbRows; t += kPbThreads) acc[t] = 0;
    const int t0 = chunk_ptr[o], t1 = chunk_ptr[o + 1];
    PbCell<T> q[kPbDepth];  // the next kPbDepth cells, in flight while the current one is accumulated
#pragma unroll
    for (int d = 0; d < kPbDepth; ++d) pb_fetch(q[d], colptr, rowidx, val, order, t0 + d, t1);
    __syncthreads();
    for (int t = t0; t < t1; t += kPbDepth) {
#pragma unroll
        for (int d = 0; d < kPbDepth; ++d) {  // slot d is used and refilled in place: registers of loads in flight never move
            if (t + d >= t1) break;  // workgroup-uniform
            PbCell<T> &a = q[d];
#pragma unroll
            for (int i = 0; i < kPbPer; ++i)
                if (a.r[i] >= r0 && a.r[i] < r1) acc[a.r[i] - r0] += a.v[i];
            for (int64_t e = a.e0 + static_cast<int64_t>(kPbPer) * kPbThreads + threadIdx.x; e < a.e1; e += kPbThreads) {
                const int r = rowidx[e];
                if (r >= r0 && r < r1) acc[r - r0] += val[e];
            }
            pb_fetch(a, colptr, rowidx, val, order, t + d + kPbDepth, t1);
            __syncthreads();
        }
    }
    for (int t = threadIdx.x; t < r1 - r0; t += kPbThreads) out[static_cast<int64_t>(o) * G + r0 + t] = acc[t];
}

template <class T>
int32_t run_dense(reo_ctx *c, const T *X, int64_t G, int64_t C, int64_t ld, const int32_t *order, int64_t n_order,
                  const int32_t *chunk_ptr, int32_t n_out, T *out)
{
    DevBuf<T> dX, dOut;
    DevBuf<int32_t> dOrd, dPtr;
    int32_t rc;
    if ((rc = dX.ensure(static_cast<size_t>(G) * C)) || (rc = dOut.ensure(static_cast<size_t>(G) * n_out)) ||
        (rc = dOrd.ensure(std::max<int64_t>(n_order, 1))) || (rc = dPtr.ensure(n_out + 1)))
        return rc;
    struct Drain {   // no exit leaves a copy from or into the caller's arrays in flight
        reo_ctx *c;
        ~Drain() { if (c->up) (void)hipStreamSynchronize(c->up); (void)hipStreamSynchronize(c->stream); }
    } drain{c};
    // the cell matrix in chunks of columns, Int64 narrowed on the way (transform.hip, upload_columns)
    if ((rc = upload_columns(c, X, ld, G, C, dX.p, std::is_same<T, double>::value ? 1 : 2))) return rc;
    hipError_t e = hipSuccess;
    if (e == hipSuccess && n_order) e = hipMemcpyAsync(dOrd.p, order, n_order * sizeof(int32_t), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dPtr.p, chunk_ptr, (n_out + 1) * sizeof(int32_t), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        tic(c, 7);
        pb_dense<T><<<dim3(static_cast<unsigned>((G + 255) / 256), n_out), 256, 0, c->stream>>>(dX.p, G, static_cast<int>(G), dOrd.p, dPtr.p, dOut.p);
        toc(c);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, dOut.p, static_cast<size_t>(G) * n_out * sizeof(T), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) (void)hipStreamSynchronize(c->stream);   // nothing queued reads or writes the caller's arrays (or these buffers) after return
    dX.release(); dOut.release(); dOrd.release(); dPtr.release();
    if (e != hipSuccess) { set_error("pseudobulk (dense) failed: %s", hipGetErrorString(e)); return e == hipErrorOutOfMemory ? REO_ENOMEM : REO_EHIP; }
    collect_timings(c);
    return REO_OK;
}

// widening kernels of the narrowed CSC upload (consecutive lanes, consecutive entries)
template <class N, class W>
__global__ __launch_bounds__(256) void pb_widen(const N *__restrict__ src, W *__restrict__ dst, size_t n)
{
    const size_t base = static_cast<size_t>(blockIdx.x) * 2048 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const size_t i = base + static_cast<size_t>(k) * 256;
        if (i < n) dst[i] = static_cast<W>(src[i]);
    }
}

// CSC input from host memory (round 5).  At config 5 (60 M entries: 720 MB) the call used to spend 13 ms checking the row indices in
// one host thread and 12.6 ms uploading at the link's 57 GB/s, for a 0.3 ms kernel.  Now the entries travel in chunks: a pool of
// host threads checks a chunk's row indices and narrows it into a pinned staging slot -- row indices to 16 bits when G <= 65 536,
// Int64 values to 16 (else 32) bits when they fit, chunk by chunk, the widths only growing -- the link carries 4 to 8 bytes per
// entry instead of 12, and kernels widen the chunk into the arrays that pb_csc reads.  Exact: what arrives is the caller's data.
// Float64 values go over the link as they are (from the caller's array), beside the narrowed row indices.
template <class T>
int32_t run_csc(reo_ctx *c, int64_t G, int64_t C, const int64_t *colptr, const int32_t *rowidx, const T *val,
                const int32_t *order, int64_t n_order, const int32_t *chunk_ptr, int32_t n_out, T *out)
{
    const int64_t nnz = colptr[C];
    DevBuf<int64_t> dCp;
    DevBuf<int32_t> dRi, dOrd, dPtr;
    DevBuf<T> dVal, dOut;
    int32_t rc;
    if ((rc = dCp.ensure(C + 1)) || (rc = dRi.ensure(std::max<int64_t>(nnz, 1))) || (rc = dVal.ensure(std::max<int64_t>(nnz, 1))) ||
        (rc = dOut.ensure(static_cast<size_t>(G) * n_out)) || (rc = dOrd.ensure(std::max<int64_t>(n_order, 1))) ||
        (rc = dPtr.ensure(n_out + 1)))
        return rc;
    struct Drain {   // no exit leaves a copy from or into the caller's arrays in flight
        reo_ctx *c;
        ~Drain() { if (c->up) (void)hipStreamSynchronize(c->up); (void)hipStreamSynchronize(c->stream); }
    } drain{c};
    hipError_t e = hipMemcpyAsync(dCp.p, colptr, (C + 1) * sizeof(int64_t), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && n_order) e = hipMemcpyAsync(dOrd.p, order, n_order * sizeof(int32_t), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dPtr.p, chunk_ptr, (n_out + 1) * sizeof(int32_t), hipMemcpyHostToDevice, c->stream);
    bool bad_row = false;
    if (e == hipSuccess && nnz > 0 && c->upload_threads > 0) {
        if ((rc = ensure_upload_streams(c))) return rc;
        constexpr bool kIntVal = std::is_same<T, long long>::value;
        const bool r16 = G <= 65536;
        const int64_t CHN = int64_t(4) << 20;   // entries per chunk: 8-16 MB of staging for the row indices, as much for the values
        const size_t rbytes = static_cast<size_t>(CHN) * (r16 ? 2 : 4), slot_bytes = rbytes + (kIntVal ? static_cast<size_t>(CHN) * 4 : 0);
        if ((rc = ensure_staging(c, slot_bytes))) return rc;
        const int nthreads = std::max(1, std::min<int>(c->upload_threads, static_cast<int>(std::thread::hardware_concurrency())));
        int width = kIntVal ? 2 : 8;   // bytes of a value on the link (8: the caller's array itself)
        int64_t k = 0;
        for (int64_t e0 = 0; e0 < nnz && e == hipSuccess; e0 += CHN, ++k) {
            const int64_t n = std::min(CHN, nnz - e0);
            const int sl = static_cast<int>(k % 3);
            if (k >= 3) e = hipEventSynchronize(c->ev_stage[sl]);   // the copy that read this slot's pinned half is done
            if (e != hipSuccess) break;
            unsigned char *hs = c->stage_h[sl], *ds = c->stage_d[sl].p;
            std::atomic<int> rows_ok{1}, fits{1};
            const int64_t per = (n + nthreads - 1) / nthreads;
            bool first_try = true;
            for (;;) {   // (again, one width up, when a value of this chunk does not fit)
                fits.store(1);
                const int w = width;
                host_parallel(nthreads, nthreads, [&](int t) {
                    const int64_t a = std::min(n, t * per), b = std::min(n, a + per);
                    if (a >= b) return;
                    if (first_try) {   // row indices: range check + 16-bit form (or a plain copy into the staging slot)
                        const int32_t *r = rowidx + e0;
                        uint32_t bad = 0;
                        if (r16) { uint16_t *d = reinterpret_cast<uint16_t *>(hs); for (int64_t i = a; i < b; ++i) { const uint32_t v = static_cast<uint32_t>(r[i]); bad |= v >= static_cast<uint32_t>(G) ? 1u : 0u; d[i] = static_cast<uint16_t>(v); } }
                        else { int32_t *d = reinterpret_cast<int32_t *>(hs); for (int64_t i = a; i < b; ++i) { const uint32_t v = static_cast<uint32_t>(r[i]); bad |= v >= static_cast<uint32_t>(G) ? 1u : 0u; d[i] = r[i]; } }
                        if (bad) rows_ok.store(0);
                    }
                    if (kIntVal && w < 8) {
                        const long long *v = reinterpret_cast<const long long *>(val) + e0;
                        long long lost = 0;
                        if (w == 2) { int16_t *d = reinterpret_cast<int16_t *>(hs + rbytes); for (int64_t i = a; i < b; ++i) { const int16_t q = static_cast<int16_t>(v[i]); d[i] = q; lost |= v[i] ^ static_cast<long long>(q); } }
                        else { int32_t *d = reinterpret_cast<int32_t *>(hs + rbytes); for (int64_t i = a; i < b; ++i) { const int32_t q = static_cast<int32_t>(v[i]); d[i] = q; lost |= v[i] ^ static_cast<long long>(q); } }
                        if (lost) fits.store(0);
                    }
                });
                first_try = false;
                if (fits.load() || width >= 8) break;
                width *= 2;
            }
            if (!rows_ok.load()) { bad_row = true; break; }
            // the link: row indices (+ narrowed values) from the pinned slot, then widened on the upload stream; values that are not
            // narrowed go straight from the caller's array
            const unsigned grid = static_cast<unsigned>((n + 2047) / 2048);
            e = hipMemcpyAsync(ds, hs, static_cast<size_t>(n) * (r16 ? 2 : 4), hipMemcpyHostToDevice, c->up);
            if (e == hipSuccess && kIntVal && width < 8) e = hipMemcpyAsync(ds + rbytes, hs + rbytes, static_cast<size_t>(n) * width, hipMemcpyHostToDevice, c->up);
            if (e == hipSuccess) e = hipEventRecord(c->ev_stage[sl], c->up);
            if (e != hipSuccess) break;
            if (r16) pb_widen<uint16_t, int32_t><<<grid, 256, 0, c->up>>>(reinterpret_cast<const uint16_t *>(ds), dRi.p + e0, static_cast<size_t>(n));
            else pb_widen<int32_t, int32_t><<<grid, 256, 0, c->up>>>(reinterpret_cast<const int32_t *>(ds), dRi.p + e0, static_cast<size_t>(n));
            if (kIntVal && width == 2) pb_widen<int16_t, T><<<grid, 256, 0, c->up>>>(reinterpret_cast<const int16_t *>(ds + rbytes), dVal.p + e0, static_cast<size_t>(n));
            else if (kIntVal && width == 4) pb_widen<int32_t, T><<<grid, 256, 0, c->up>>>(reinterpret_cast<const int32_t *>(ds + rbytes), dVal.p + e0, static_cast<size_t>(n));
            else e = hipMemcpyAsync(dVal.p + e0, val + e0, static_cast<size_t>(n) * sizeof(T), hipMemcpyHostToDevice, c->up);
            if (e == hipSuccess) e = hipGetLastError();
        }
        if (e == hipSuccess && !bad_row) {   // the kernel on the context's stream follows the upload stream
            e = hipEventRecord(c->ev_up[0], c->up);
            if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ev_up[0], 0);
        }
    } else if (e == hipSuccess && nnz > 0) {   // REO_UPLOAD_THREADS=0: one copy each, the row indices checked by the calling thread
        for (int64_t q = 0; q < nnz; ++q)
            if (rowidx[q] < 0 || rowidx[q] >= G) { bad_row = true; break; }
        if (!bad_row) e = hipMemcpyAsync(dRi.p, rowidx, nnz * sizeof(int32_t), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess && !bad_row) e = hipMemcpyAsync(dVal.p, val, nnz * sizeof(T), hipMemcpyHostToDevice, c->stream);
    }
    if (bad_row) { set_error("a row index is outside [0,%lld)", (long long)G); return REO_EINVAL; }
    if (e == hipSuccess) {
        tic(c, 7);
        pb_csc<T><<<dim3(n_out, static_cast<unsigned>((G + kPbRows - 1) / kPbRows)), kPbThreads, 0, c->stream>>>(
            dCp.p, dRi.p, dVal.p, static_cast<int>(G), dOrd.p, dPtr.p, dOut.p);
        toc(c);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, dOut.p, static_cast<size_t>(G) * n_out * sizeof(T), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { set_error("pseudobulk (CSC) failed: %s", hipGetErrorString(e)); return e == hipErrorOutOfMemory ? REO_ENOMEM : REO_EHIP; }
    collect_timings(c);
    return REO_OK;
}

int32_t check_args(reo_ctx *c, int64_t G, int64_t C, const int32_t *order, int64_t n_order, const int32_t *chunk_ptr,
                   int32_t n_out, const void *in, const void *out)
{
    if (!c || !in || !out || !chunk_ptr || (n_order > 0 && !order)) { set_error("null argument"); return REO_EINVAL; }
    if (G < 1 || C < 1 || n_out < 1 || n_order < 0) { set_error("bad pseudobulk shape"); return REO_EINVAL; }
    if (chunk_ptr[0] != 0 || chunk_ptr[n_out] != n_order) { set_error("chunk_ptr must run from 0 to n_order"); return REO_EINVAL; }
    for (int o = 0; o < n_out; ++o)
        if (chunk_ptr[o + 1] < chunk_ptr[o]) { set_error("chunk_ptr must be non-decreasing"); return REO_EINVAL; }
    for (int64_t t = 0; t < n_order; ++t)
        if (order[t] < 0 || order[t] >= C) { set_error("cell index %d outside [0,%lld)", order[t], (long long)C); return REO_EINVAL; }
    REO_HIP_CHECK(hipSetDevice(c->device));
    return REO_OK;
}

}  // namespace

}  // namespace reo

using namespace reo;

extern "C" {

int32_t reo_pseudobulk_dense_f64(reo_ctx *c, const double *X, int64_t G, int64_t C, int64_t ld, const int32_t *order,
                                 int64_t n_order, const int32_t *chunk_ptr, int32_t n_out, double *out)
{
    int32_t rc = check_args(c, G, C, order, n_order, chunk_ptr, n_out, X, out);
    if (rc) return rc;
    if (ld < G) { set_error("leading dimension < G"); return REO_EINVAL; }
    return run_dense<double>(c, X, G, C, ld, order, n_order, chunk_ptr, n_out, out);
}

int32_t reo_pseudobulk_dense_i64(reo_ctx *c, const int64_t *X, int64_t G, int64_t C, int64_t ld, const int32_t *order,
                                 int64_t n_order, const int32_t *chunk_ptr, int32_t n_out, int64_t *out)
{
    int32_t rc = check_args(c, G, C, order, n_order, chunk_ptr, n_out, X, out);
    if (rc) return rc;
    if (ld < G) { set_error("leading dimension < G"); return REO_EINVAL; }
    return run_dense<long long>(c, reinterpret_cast<const long long *>(X), G, C, ld, order, n_order, chunk_ptr, n_out,
                                reinterpret_cast<long long *>(out));
}

static int32_t check_csc(int64_t G, int64_t C, const int64_t *colptr, const int32_t *rowidx)
{
    if (!colptr || colptr[0] != 0) { set_error("colptr must start at 0"); return REO_EINVAL; }
    for (int64_t k = 0; k < C; ++k)
        if (colptr[k + 1] < colptr[k]) { set_error("colptr must be non-decreasing"); return REO_EINVAL; }
    if (colptr[C] > 0 && !rowidx) { set_error("rowidx is null"); return REO_EINVAL; }
    (void)G;   // (the row indices are checked chunk by chunk while they are narrowed for the upload: run_csc)
    return REO_OK;
}

int32_t reo_pseudobulk_csc_f64(reo_ctx *c, int64_t G, int64_t C, const int64_t *colptr, const int32_t *rowidx,
                               const double *val, const int32_t *order, int64_t n_order, const int32_t *chunk_ptr,
                               int32_t n_out, double *out)
{
    const auto w0 = std::chrono::steady_clock::now();
    int32_t rc = check_args(c, G, C, order, n_order, chunk_ptr, n_out, colptr, out);
    if (rc || (rc = check_csc(G, C, colptr, rowidx))) return rc;
    if (c->debug_passes) fprintf(stderr, "  pseudobulk csc: arguments checked after %.0f us\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count());
    rc = run_csc<double>(c, G, C, colptr, rowidx, val, order, n_order, chunk_ptr, n_out, out);
    if (c->debug_passes) fprintf(stderr, "  pseudobulk csc: done after %.0f us\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count());
    return rc;
}

int32_t reo_pseudobulk_csc_i64(reo_ctx *c, int64_t G, int64_t C, const int64_t *colptr, const int32_t *rowidx,
                               const int64_t *val, const int32_t *order, int64_t n_order, const int32_t *chunk_ptr,
                               int32_t n_out, int64_t *out)
{
    const auto w0 = std::chrono::steady_clock::now();
    int32_t rc = check_args(c, G, C, order, n_order, chunk_ptr, n_out, colptr, out);
    if (rc || (rc = check_csc(G, C, colptr, rowidx))) return rc;
    if (c->debug_passes) fprintf(stderr, "  pseudobulk csc: arguments checked after %.0f us\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count());
    rc = run_csc<long long>(c, G, C, colptr, rowidx, reinterpret_cast<const long long *>(val), order, n_order,
                            chunk_ptr, n_out, reinterpret_cast<long long *>(out));
    if (c->debug_passes) fprintf(stderr, "  pseudobulk csc: done after %.0f us\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count());
    return rc;
}

}  // extern "C"
