set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5f
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1
python tools/fuzz_gpu.py 600 2026 > $O/fuzz_gpu_600_seed2026.txt 2>&1
python tools/host_breakdown.py > $O/host_breakdown.txt 2>&1
