set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5e
mkdir -p $O
python tools/host_breakdown.py > $O/host_breakdown.txt 2>&1
REO_DEBUG_PASSES=1 python tools/host_breakdown.py > $O/host_breakdown_dbg.txt 2>&1 || true
python bench.py --gpus 2 --debug-gloo-one-gpu --steps 4 --warmup 1 --no-cpu-baseline --no-tie-rich --no-float64 --no-cycle-watch > $O/bench_2ranks_one_gpu.json 2> $O/bench_2ranks.err
