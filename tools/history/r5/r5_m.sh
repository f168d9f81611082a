set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5m
mkdir -p $O
python tools/fuzz_gpu.py 600 97531 > $O/fuzz_gpu_600_seed97531.txt 2>&1
python tools/fuzz_upload.py 250 2468 > $O/fuzz_upload_250.txt 2>&1
python tools/fuzz_transform.py 600 1357 > $O/fuzz_transform_600.txt 2>&1
python tools/fuzz_transform.py 80 8642 big > $O/fuzz_transform_big_80.txt 2>&1
python tools/fuzz_light.py 1200 > $O/fuzz_light_1200.txt 2>&1
