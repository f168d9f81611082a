set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5k
mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "pipelined_upload or config3 or concurrent_host or error_paths or reoa_bundled or plain_c_client" > $O/tests.log 2>&1
python tools/fuzz_upload.py 100 31 > $O/fuzz_upload_100.txt 2>&1
python tools/from_host_breakdown.py t0 > $O/fhb_t0.txt 2>&1
python tools/from_host_breakdown.py t1 > $O/fhb_t1.txt 2>&1
python tools/from_host_breakdown.py float > $O/fhb_float.txt 2>&1
python tools/from_host_breakdown.py t0 30000 4000 > $O/fhb_c4.txt 2>&1
