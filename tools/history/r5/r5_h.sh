set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5h
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1
python tools/fuzz_transform.py 300 777 > $O/fuzz_transform_300.txt 2>&1
python tools/fuzz_transform.py 60 4243 big > $O/fuzz_transform_big_60.txt 2>&1
python tools/transform_big_time.py big > $O/transform_big_time.txt 2>&1
