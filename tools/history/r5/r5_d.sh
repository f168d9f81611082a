set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5d
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined_upload or error_paths or host_matrix_view or plain_c_client" > $O/eager_tests.log 2>&1
python tools/from_host_breakdown.py t0 > $O/fhb_t0.txt 2>&1
python tools/from_host_breakdown.py t1 > $O/fhb_t1.txt 2>&1
python tools/from_host_breakdown.py t0 30000 4000 > $O/fhb_c4.txt 2>&1
