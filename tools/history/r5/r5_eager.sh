# Round 5: the pipelined upload -- its parity tests, then the from_host block of the bench line.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined_upload or error_paths or host_matrix_view or plain_c_client or reoa_bundled or golden" > $O/eager_tests.log 2>&1
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-tie-rich --no-cycle-watch > $O/bench.json 2> $O/bench.err
