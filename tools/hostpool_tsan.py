"""ThreadSanitizer stress of the host thread pool of the narrowed upload (HostPool, csrc/transform.hip): the class is cut out of the
source verbatim, four caller threads run 20 000 short jobs each with task counts around the worker count, every task must run
exactly once, TSan must stay silent.  CPU only.    python tools/hostpool_tsan.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = open(os.path.join(ROOT, "rankcompv3.jl_amd", "csrc", "transform.hip")).read()
a = s.index("class HostPool {"); b = s.index("};", s.index("    uint64_t gen_ = 0;", a)) + 2
src = '''#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
''' + s[a:b] + '''
int main()
{
    std::atomic<long> bad{0};
    std::vector<std::thread> callers;
    for (int c = 0; c < 4; ++c)
        callers.emplace_back([&, c] {
            for (int rep = 0; rep < 20000; ++rep) {
                const int n = 1 + (rep * 7 + c) % 13;
                std::vector<int> hits(n, 0);
                HostPool::get(2 + rep % 7).run(n, [&](int t) { hits[t]++; });
                for (int t = 0; t < n; ++t) if (hits[t] != 1) bad++;
            }
        });
    for (auto &t : callers) t.join();
    printf("HostPool stress: %ld tasks ran a number of times other than once\\n", bad.load());
    return bad.load() != 0;
}
'''
open("/tmp/hostpool_stress.cpp", "w").write(src)
for flags, name in ((["-O1", "-g", "-fsanitize=thread"], "tsan"), (["-O2"], "plain")):
    subprocess.check_call(["g++", *flags, "-pthread", "-o", "/tmp/hostpool_" + name, "/tmp/hostpool_stress.cpp"])
    out = subprocess.run(["/tmp/hostpool_" + name], capture_output=True, text=True)
    print(name + ":", out.stdout.strip(), "| stderr lines:", len(out.stderr.strip().splitlines()), "| exit", out.returncode)
    if out.returncode or out.stderr.strip():
        print(out.stderr[:4000]); sys.exit(1)
print("hostpool_tsan: clean")
