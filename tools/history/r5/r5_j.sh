set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5j
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1
python tools/host_breakdown.py > $O/host_breakdown.txt 2>&1
python tools/fuzz_gpu.py 300 4711 > $O/fuzz_gpu_300.txt 2>&1
