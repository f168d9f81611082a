#!/bin/bash
# Host-side AddressSanitizer audit of libreo_hip.so (round 3; the fuzz SIGSEGV of round 1 is still open): the host
# code of every translation unit is built with -fsanitize=address (device code untouched: GPU ASan is not available
# on this pool), and everything that can run WITHOUT a device is run against it -- symbol table, reo_threshold over
# n = 2..1500, reo_create / reo_create_multi error paths, null and out-of-range arguments of every entry point that
# checks before it touches the GPU.  (On a GPU box the instrumented library cannot create a context: AMD's ASan runtime
# intercepts hsa_amd_memory_pool_allocate and refuses without xnack, which this pool does not offer --
# gpurun_out/r3_asan_host.log, round 3.  The audit is therefore CPU-only.)
set -e
cd "$(dirname "$0")/.."
B=${TMPDIR:-/tmp}/reo_asan
mkdir -p $B
SRC=rankcompv3.jl_amd/csrc
for f in api kernels transform pseudobulk comm; do
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fsanitize=address -fno-gpu-sanitize \
      -Iinclude -I$SRC -c -o $B/$f.o $SRC/$f.hip &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address -fno-gpu-sanitize -o $B/libreo_hip_asan.so \
    $B/api.o $B/kernels.o $B/transform.o $B/pseudobulk.o $B/comm.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
RT=$(find /opt/rocm/lib/llvm/lib/clang -name "libclang_rt.asan-x86_64.so" | head -1)
export LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 REO_LIB_PATH=$B/libreo_hip_asan.so
python3 tools/asan_host_calls.py
python3 -m pytest tests/test_library_cpu.py -x -q -p no:cacheprovider
echo "asan_host: clean"
