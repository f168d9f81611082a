# Round-4 final evidence (run on the GPU box through gpurun): the GPU suite, the bench line, and rocprofv3 kernel stats of (a) the
# headline legs alone -- so that the average of k1w_pairs<15,false> is config 3's -- and (b) the whole default bench.
# Summaries land in gpurun_out/r4z; what is kept is copied to profiles/r4_z_*.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4z
mkdir -p $O
if [ "$1" != "stats-only" ]; then
  python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1
  python bench.py > $O/bench.json 2> $O/bench.err
fi
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_h -o s -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-config4 --no-float64 --no-tie-rich --no-cycle-watch > $O/bench_headline_under_rocprof.json 2> $O/stats_h.log
cp $(find $O/stats_h -name "s_kernel_stats.csv" | head -1) $O/kernel_stats_headline.csv
rm -rf $O/stats_h
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.log
cp $(find $O/stats -name "s_kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/stats
