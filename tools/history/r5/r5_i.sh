set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5i
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1
python tools/upload_sweep.py t0 > $O/sweep_c3.txt 2>&1
python tools/host_breakdown.py > $O/host_breakdown.txt 2>&1
