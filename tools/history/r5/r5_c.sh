set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-tie-rich --no-cycle-watch > $O/bench.json 2> $O/bench.err
python tools/host_breakdown.py > $O/host_breakdown.txt 2>&1
