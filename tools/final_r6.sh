# Round-6 final evidence (run on the GPU box through gpurun, in two calls: `tests` and `profiles`).
#   tests:    the plain `-m gpu` suite as the driver runs it (timed, with the slowest tests listed), then the FULL set including the
#             gpu_slow cases (REO_RUN_SLOW=1), the randomized runs, the two-rank rehearsals on one GPU (gloo) incl. the failure path
#   profiles: the bench line, rocprofv3 kernel stats of the headline legs alone and of the whole default bench, PMC passes (FETCH_SIZE /
#             WRITE_SIZE / SQ counters, each in its own pass) for K1 at config 3 (both families) and config 4, and for the tally scan
# Summaries land in gpurun_out/r6z; what is kept is copied to profiles/r6_z_*.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6z
mkdir -p $O
if [ "$1" = "tests" ]; then
  python -m pytest tests -m gpu -x -q --durations=25 > $O/gpu_tests.log 2>&1
  REO_RUN_SLOW=1 python -m pytest tests -m gpu -x -q -k "largest_gene_count or 140000 or (config4 and t1)" > $O/gpu_tests_slow_cases.log 2>&1
  python tools/fuzz_gpu.py 400 60606 > $O/fuzz_gpu_400.txt 2>&1
  python tools/fuzz_transform.py 300 60607 > $O/fuzz_transform_300.txt 2>&1
  python tools/fuzz_transform.py 40 60608 big > $O/fuzz_transform_big_40.txt 2>&1
  python tools/fuzz_upload.py 150 60609 > $O/fuzz_upload_150.txt 2>&1
  python tools/fuzz_light.py 600 > $O/fuzz_light_600.txt 2>&1
  python tools/fuzz_shards.py 30 > $O/fuzz_shards_30.txt 2>&1
  python bench.py --gpus 2 --debug-gloo-one-gpu --steps 4 --warmup 1 --no-cpu-baseline --no-tie-rich --no-float64 > $O/bench_2ranks_rehearsal_one_gpu_gloo.json 2> $O/bench_2ranks.err
  REO_BENCH_FAIL_RANK=1 python bench.py --gpus 2 --debug-gloo-one-gpu --steps 2 --warmup 1 --no-cpu-baseline --no-tie-rich --no-float64 --no-config4 > $O/bench_2ranks_rank1_fails.json 2> $O/bench_2ranks_fail.err || echo "exit code $?" >> $O/bench_2ranks_rank1_fails.json
  exit 0
fi
python bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_h -o s -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-config4 --no-float64 --no-tie-rich --no-cycle-watch --no-from-host > $O/bench_headline_under_rocprof.json 2> $O/stats_h.log
cp $(find $O/stats_h -name "s_kernel_stats.csv" | head -1) $O/kernel_stats_headline.csv
rm -rf $O/stats_h
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.log
cp $(find $O/stats -name "s_kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/stats
for job in "t0:tools/k1_only.py t0" "t1:tools/k1_only.py t1" "c4:tools/k1_shape.py 30000 4000 t0"; do
  name=${job%%:*}; B="python3 ${job#*:}"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${name}_fetch -o p -- $B > /dev/null 2> $O/${name}_fetch.log
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${name}_write -o p -- $B > /dev/null 2> $O/${name}_write.log
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/${name}_sq -o p -- $B > /dev/null 2> $O/${name}_sq.log
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/${name}_sq2 -o p -- $B > /dev/null 2> $O/${name}_sq2.log
  python3 profiles/summarize_pmc.py $O/${name}_pmc_per_launch.csv $O/${name}_fetch $O/${name}_write $O/${name}_sq $O/${name}_sq2
  rm -rf $O/${name}_fetch $O/${name}_write $O/${name}_sq $O/${name}_sq2
done
B="python3 tools/k2_only.py"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/k2_fetch -o p -- $B > /dev/null 2> $O/k2_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/k2_write -o p -- $B > /dev/null 2> $O/k2_write.log
python3 profiles/summarize_pmc.py $O/k2_pmc_per_launch.csv $O/k2_fetch $O/k2_write
rm -rf $O/k2_fetch $O/k2_write
grep -E "k1w_pairs" $O/t0_pmc_per_launch.csv $O/c4_pmc_per_launch.csv | head -40
