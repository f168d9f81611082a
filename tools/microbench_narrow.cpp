// How fast can host threads narrow an Int64 matrix (pageable memory) to 32- or 16-bit numbers?  If PCIe (57 GB/s measured from
// pageable memory) is the floor of the drop-in call's upload, sending fewer bytes is the only way under it -- provided the host can
// produce them faster than the link takes the wide ones.  g++ -O3 -march=native -pthread; run on the GPU box's host.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
template <class N>
static void narrow(const int64_t *src, N *dst, size_t n, int64_t *bad)
{
    int64_t b = 0;
    for (size_t i = 0; i < n; ++i) { const int64_t v = src[i]; dst[i] = static_cast<N>(v); b |= v ^ static_cast<int64_t>(static_cast<N>(v)); }
    *bad = b;
}
int main(int argc, char **argv)
{
    const size_t G = 20000, S = argc > 1 ? atoi(argv[1]) : 1000, n = G * S;
    std::vector<int64_t> X(n);
    for (size_t i = 0; i < n; ++i) X[i] = static_cast<int64_t>((i * 2654435761u) % 20000);
    std::vector<int32_t> o32(n);
    std::vector<int16_t> o16(n);
    for (int T : {1, 4, 8, 16, 32, 64}) {
        for (int w = 0; w < 2; ++w) {
            double best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                std::vector<int64_t> bad(T);
                const auto t0 = std::chrono::steady_clock::now();
                std::vector<std::thread> th;
                for (int t = 0; t < T; ++t)
                    th.emplace_back([&, t] {
                        const size_t a = n * t / T, b = n * (t + 1) / T;
                        if (w == 0) narrow<int32_t>(X.data() + a, o32.data() + a, b - a, &bad[t]);
                        else narrow<int16_t>(X.data() + a, o16.data() + a, b - a, &bad[t]);
                    });
                for (auto &x : th) x.join();
                best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
            }
            printf("%2d threads, int64 -> int%d: %.2f ms for %.0f MB in (%.1f GB/s read)\n", T, w ? 16 : 32, best * 1e3, n * 8 / 1e6, n * 8 / best / 1e9);
        }
    }
    return 0;
}
