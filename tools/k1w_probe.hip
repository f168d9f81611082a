// Timing probe of the generated K1 count loop (round 3): the production item order of config 3 (20 000 genes, one side
// of 16 blocks per item), loop variants with parts switched off, 1-3 waves per SIMD (dynamic LDS), item lists with and
// without the empty items.  Results are NOT checked (the variants compute garbage); parity lives in tests/.
//   python3 tools/k1w_probe_gen.py > tools/bin/k1w_probe_gen.inc
//   hipcc -O3 --offload-arch=gfx950 -Itools/bin tools/k1w_probe.hip -o tools/bin/k1w_probe && tools/bin/k1w_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "k1w_probe_gen.inc"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int V>
__global__ __launch_bounds__(64, 3) void probe(const uint4 *P, const uint4 *AL, int Gp, int nblk, const uint32_t *items, uint32_t *out)
{
    extern __shared__ uint4 ring[];
    const uint32_t it = items[blockIdx.x];
    if (it == 0xFFFFFFFFu) return;
    const int i0 = __builtin_amdgcn_readfirstlane(static_cast<int>(it & 0xFFFFu) * 32);
    const int jw = __builtin_amdgcn_readfirstlane(static_cast<int>((it >> 16) & 0x7FFFu) * 256);
    const int side = __builtin_amdgcn_readfirstlane(static_cast<int>(it >> 31));
    const int bb = side * nblk;
    u32x16 c0, c1, c2, c3;
    const char *pb = reinterpret_cast<const char *>(P) + static_cast<size_t>(bb) * 4 * Gp * 16;
    const char *al = reinterpret_cast<const char *>(AL) + (static_cast<size_t>(bb) * Gp + i0) * 64;
    const uint32_t lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(&ring[0]));
    const uint32_t poff = static_cast<uint32_t>(jw + threadIdx.x) * 16u, aoff = threadIdx.x * 16u;
#define CALL(i, fn, nm) if (V == i) fn(c0, c1, c2, c3, pb, static_cast<uint32_t>(Gp) * 16u, al, al, static_cast<uint32_t>(Gp) * 64u, static_cast<uint32_t>(nblk), poff, aoff, lds);
    PROBE_VARIANTS(CALL)
#undef CALL
    uint32_t s = 0;
#pragma unroll
    for (int h = 0; h < 16; ++h) s += c0[h] + c1[h] + c2[h] + c3[h];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

// the four-waves-per-SIMD experiment: items of 16 rows (128 registers)
template <int V>
__global__ __launch_bounds__(64, 4) void probe16(const uint4 *P, const uint4 *AL, int Gp, int nblk, const uint32_t *items, uint32_t *out)
{
    extern __shared__ uint4 ring[];
    const uint32_t it = items[blockIdx.x];
    if (it == 0xFFFFFFFFu) return;
    const int i0 = __builtin_amdgcn_readfirstlane(static_cast<int>(it & 0xFFFFu) * 16);
    const int jw = __builtin_amdgcn_readfirstlane(static_cast<int>((it >> 16) & 0x7FFFu) * 256);
    const int side = __builtin_amdgcn_readfirstlane(static_cast<int>(it >> 31));
    const int bb = side * nblk;
    u32x8 c0, c1, c2, c3;
    const char *pb = reinterpret_cast<const char *>(P) + static_cast<size_t>(bb) * 4 * Gp * 16;
    const char *al = reinterpret_cast<const char *>(AL) + (static_cast<size_t>(bb) * Gp + i0) * 64;
    const uint32_t lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(&ring[0]));
    const uint32_t poff = static_cast<uint32_t>(jw + threadIdx.x) * 16u, aoff = threadIdx.x * 16u;
    if (V == 0) probe16_half(c0, c1, c2, c3, pb, static_cast<uint32_t>(Gp) * 16u, al, al, static_cast<uint32_t>(Gp) * 64u, static_cast<uint32_t>(nblk), poff, aoff, lds);
    else probe16_half_noreload(c0, c1, c2, c3, pb, static_cast<uint32_t>(Gp) * 16u, al, al, static_cast<uint32_t>(Gp) * 64u, static_cast<uint32_t>(nblk), poff, aoff, lds);
    uint32_t s = 0;
#pragma unroll
    for (int h = 0; h < 8; ++h) s += c0[h] + c1[h] + c2[h] + c3[h];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

int main()
{
    const int G = 20000, Gp = 21504, nblk = 16, nblk_all = 32;
    // production item order (kernels.hip, wave_item): units = panels of Q = 4 chunks of 1024 columns x 32 i-tiles;
    // unit u belongs to XCD slot u & 7; inside a unit: side, i-tile, wave chunk
    const int Q = 4, QW = 16, NJ = Gp / 1024, NIT = Gp / 32, NP = (NJ + Q - 1) / Q;
    std::vector<std::vector<uint32_t>> slot_full(8), slot_compact(8);
    int nunits = 0;
    long active = 0;
    for (int p = 0; p < NP; ++p) {
        const int ni = std::min(NIT, 32 * Q * (p + 1));
        for (int r = 0; r * 32 < ni; ++r, ++nunits) {
            const int s = nunits & 7;
            for (int side = 0; side < 2; ++side)
                for (int t = 0; t < 32; ++t)
                    for (int w = 0; w < QW; ++w) {
                        const int i0 = (r * 32 + t) * 32, jw = (p * QW + w) * 256;
                        const bool ok = i0 < G && jw < Gp && !(jw >= G || ((jw + 255) >> 6) < (i0 >> 6));
                        const uint32_t it = ok ? (static_cast<uint32_t>(side) << 31 | static_cast<uint32_t>(jw / 256) << 16 | static_cast<uint32_t>(i0 / 32)) : 0xFFFFFFFFu;
                        slot_full[s].push_back(it);
                        if (ok) { slot_compact[s].push_back(it); ++active; }
                    }
        }
    }
    auto interleave = [](const std::vector<std::vector<uint32_t>> &sl) {
        size_t m = 0;
        for (auto &v : sl) m = std::max(m, v.size());
        std::vector<uint32_t> out(m * 8, 0xFFFFFFFFu);
        for (int s = 0; s < 8; ++s)
            for (size_t k = 0; k < sl[s].size(); ++k) out[k * 8 + s] = sl[s][k];
        return out;
    };
    std::vector<uint32_t> full = interleave(slot_full), compact = interleave(slot_compact);
    // balanced: all active items dealt round-robin to the 8 slots in production order
    std::vector<std::vector<uint32_t>> bal(8);
    { size_t k = 0; for (int s = 0; s < 8; ++s) for (uint32_t it : slot_compact[s]) bal[k++ & 7].push_back(it); }
    std::vector<uint32_t> balanced = interleave(bal);
    printf("units %d, items launched %zu, active %ld, compact grid %zu, balanced grid %zu\n", nunits, full.size(), active, compact.size(), balanced.size());

    uint4 *dP, *dA; uint32_t *dItems, *dOut;
    const size_t pbytes = static_cast<size_t>(nblk_all) * 4 * Gp * 16, abytes = static_cast<size_t>(nblk_all) * Gp * 64;
    CHECK(hipMalloc(&dP, pbytes)); CHECK(hipMalloc(&dA, abytes));
    {   // random planes: the clock under load depends on how many bits toggle
        std::vector<uint32_t> h(std::max(pbytes, abytes) / 4);
        uint64_t z = 0x9E3779B97F4A7C15ULL;
        for (auto &w : h) { z = z * 6364136223846793005ULL + 1442695040888963407ULL; w = static_cast<uint32_t>(z >> 32); }
        CHECK(hipMemcpy(dP, h.data(), pbytes, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dA, h.data(), abytes, hipMemcpyHostToDevice));
    }
    CHECK(hipMalloc(&dItems, full.size() * 4)); CHECK(hipMalloc(&dOut, full.size() * 64 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const double cmp = 0.5 * G * (G - 1.0) * 1000;

    auto run = [&](const char *name, auto kern, const std::vector<uint32_t> &items, int waves_per_simd) -> int {
        CHECK(hipMemcpy(dItems, items.data(), items.size() * 4, hipMemcpyHostToDevice));
        const size_t lds = waves_per_simd >= 3 ? 4096 : (160 * 1024 / (4 * waves_per_simd)) & ~size_t(1023);
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        float best = 1e30f;
        for (int rep = 0; rep < 8; ++rep) {
            CHECK(hipEventRecord(e0));
            kern<<<static_cast<unsigned>(items.size()), 64, lds>>>(dP, dA, Gp, nblk, dItems, dOut);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
        }
        printf("%-12s %-9s %dw  %7.3f ms  %6.2f Tcmp/s\n", name, &items == &full ? "full" : (&items == &compact ? "compact" : "balanced"), waves_per_simd, best, cmp / best / 1e9);
        return 0;
    };
    for (int round = 0; round < 3; ++round) {   // (the first launches run on a cold, slowly clocking chip: compare within a round)
        printf("-- round %d\n", round);
#define RUN(i, fn, nm) if (run(nm, probe<i>, balanced, 3)) return 1;
        PROBE_VARIANTS(RUN)
#undef RUN
    }
    {   // 16-row items: every item of the balanced list as two halves (same XCD), 4 waves per SIMD
        std::vector<uint32_t> half;
        for (size_t k = 0; k < balanced.size(); k += 8)
            for (int h = 0; h < 2; ++h)
                for (int x = 0; x < 8; ++x) {
                    const uint32_t it = balanced[k + x];
                    half.push_back(it == 0xFFFFFFFFu ? it : ((it & 0xFFFF0000u) | ((it & 0xFFFFu) * 2 + h)));
                }
        CHECK(hipFree(dItems)); CHECK(hipFree(dOut));
        CHECK(hipMalloc(&dItems, half.size() * 4)); CHECK(hipMalloc(&dOut, half.size() * 64 * 4));
        CHECK(hipMemcpy(dItems, half.data(), half.size() * 4, hipMemcpyHostToDevice));
        for (int v = 0; v < 2; ++v)
            for (int rep = 0; rep < 3; ++rep) {
                float best = 1e30f;
                for (int r2 = 0; r2 < 6; ++r2) {
                    CHECK(hipEventRecord(e0));
                    if (v == 0) probe16<0><<<static_cast<unsigned>(half.size()), 64, 2048>>>(dP, dA, Gp, nblk, dItems, dOut);
                    else probe16<1><<<static_cast<unsigned>(half.size()), 64, 2048>>>(dP, dA, Gp, nblk, dItems, dOut);
                    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                    best = std::min(best, ms);
                }
                printf("%-12s half-items 4w  %7.3f ms  %6.2f Tcmp/s\n", v == 0 ? "half" : "half_norel", best, cmp / best / 1e9);
            }
        CHECK(hipFree(dItems)); CHECK(hipFree(dOut));
        CHECK(hipMalloc(&dItems, full.size() * 4)); CHECK(hipMalloc(&dOut, full.size() * 64 * 4));
    }
    if (run("base", probe<0>, full, 3)) return 1;
    if (run("base", probe<0>, compact, 3)) return 1;
    if (run("base", probe<0>, balanced, 2)) return 1;
    return 0;
}
