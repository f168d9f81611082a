# Round-5 opening run on the GPU box: the GPU suite on the tree as round 4 left it, the default bench line, the host-buffer breakdown.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5a
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1
python bench.py > $O/bench.json 2> $O/bench.err
python tools/host_breakdown.py > $O/host_breakdown.txt 2>&1
lscpu > $O/lscpu.txt
