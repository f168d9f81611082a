# Round-5 final evidence (run on the GPU box through gpurun): the GPU suite, the bench line, rocprofv3 kernel stats of (a) the headline
# legs alone -- so that the average of k1w_pairs<15,false> is config 3's -- and (b) the whole default bench, PMC passes (FETCH_SIZE /
# WRITE_SIZE, separately) for K1 on both families and for the tally scan, and the randomized runs.  Summaries land in gpurun_out/r5z; what
# is kept is copied to profiles/r5_z_*.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5z
mkdir -p $O
if [ "$1" != "profiles-only" ]; then
  python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1
  python bench.py > $O/bench.json 2> $O/bench.err
fi
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_h -o s -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-config4 --no-float64 --no-tie-rich --no-cycle-watch --no-from-host > $O/bench_headline_under_rocprof.json 2> $O/stats_h.log
cp $(find $O/stats_h -name "s_kernel_stats.csv" | head -1) $O/kernel_stats_headline.csv
rm -rf $O/stats_h
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.log
cp $(find $O/stats -name "s_kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/stats
for fam in t0 t1; do
  B="python3 tools/k1_only.py $fam"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${fam}_fetch -o p -- $B > /dev/null 2> $O/${fam}_fetch.log
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${fam}_write -o p -- $B > /dev/null 2> $O/${fam}_write.log
  python3 profiles/summarize_pmc.py $O/${fam}_pmc_per_launch.csv $O/${fam}_fetch $O/${fam}_write
  rm -rf $O/${fam}_fetch $O/${fam}_write
done
B="python3 tools/k2_only.py"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/k2_fetch -o p -- $B > /dev/null 2> $O/k2_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/k2_write -o p -- $B > /dev/null 2> $O/k2_write.log
python3 profiles/summarize_pmc.py $O/k2_pmc_per_launch.csv $O/k2_fetch $O/k2_write
rm -rf $O/k2_fetch $O/k2_write
if [ "$1" != "profiles-only" ]; then
  python tools/fuzz_light.py 600 > $O/fuzz_light_600.txt 2>&1
  python tools/fuzz_shards.py 30 > $O/fuzz_shards_30.txt 2>&1
fi
