#!/bin/bash
# The HOST side of libreo_hip.so under AddressSanitizer with real call sequences, on a machine WITHOUT a GPU (round 5).
# tools/asan_host.sh (round 3) could only run what fails before the first HIP call.  Here every translation unit is built with
# -fsanitize=address for the host (device code untouched) and linked against tools/mockhip -- a stand-in runtime in which device
# memory is malloc'd host memory, copies are memcpy (checked by ASan at both ends, caller buffers included), launches do nothing.
# tools/asan_host_mock_calls.py then drives: the library half of tools/fuzz_gpu.py on the cases of the crash log
# (profiles/faults/r4_fuzz_gpu_host_crash.log: seed 2026), every other entry point, sharded builds through both hooks and the
# in-library communicator, the pipelined exchange, the multi-GPU context, error paths.  Numerics are garbage by construction.
set -e
cd "$(dirname "$0")/.."
B=${TMPDIR:-/tmp}/reo_asan_mock
mkdir -p $B
SRC=rankcompv3.jl_amd/csrc
CLANG=/opt/rocm/lib/llvm/bin/clang++
g++ -O1 -g -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -o $B/libmockhip.so tools/mockhip/mockhip.cpp
for f in api kernels transform pseudobulk comm; do
  if [ ! -f $B/$f.o ] || [ $SRC/$f.hip -nt $B/$f.o ] || [ $SRC/reo_internal.h -nt $B/$f.o ] || [ include/reo_hip.h -nt $B/$f.o ]; then
    /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fsanitize=address -fno-gpu-sanitize \
        -Iinclude -I$SRC -c -o $B/$f.o $SRC/$f.hip &
  fi
done
wait
$CLANG -shared -fPIC -fsanitize=address -shared-libasan -o $B/libreo_hip_asan_mock.so \
    $B/api.o $B/kernels.o $B/transform.o $B/pseudobulk.o $B/comm.o -L$B -lmockhip -Wl,-rpath,$B
RT=$(find /opt/rocm/lib/llvm/lib/clang -name "libclang_rt.asan-x86_64.so" | head -1)
export LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1 REO_LIB_PATH=$B/libreo_hip_asan_mock.so REO_MOCK_LIB=$B/libmockhip.so
REO_DEVICE_CACHE_MB=0 python3 tools/asan_host_mock_calls.py "$@"     # every release is a free: ASan sees use-after-release at once
python3 tools/asan_host_mock_calls.py "$@"                           # the block cache on (the default): blocks travel between contexts
echo "asan_host_mock: clean"
