# Round-2 profiles of the hot path (run on the GPU box through gpurun): kernel stats of the default bench, and PMC
# passes (FETCH_SIZE / WRITE_SIZE / SQ counters in separate passes: they exceed the counter hardware together) for
# the tie-free and the tie-rich family separately.  Summaries land in gpurun_out/r2c; copy what is kept to profiles/.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-tie-rich > $O/bench_under_rocprof.json 2> $O/stats.log
for fam in t0 t1; do
  B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-tie-rich --n-iter 4 --family $fam"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${fam}_fetch -o p -- $B > /dev/null 2> $O/${fam}_fetch.log
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${fam}_write -o p -- $B > /dev/null 2> $O/${fam}_write.log
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/${fam}_sq -o p -- $B > /dev/null 2> $O/${fam}_sq.log
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/${fam}_sq2 -o p -- $B > /dev/null 2> $O/${fam}_sq2.log
  python3 profiles/summarize_pmc.py $O/${fam}_pmc_per_launch.csv $O/${fam}_fetch $O/${fam}_write $O/${fam}_sq $O/${fam}_sq2
done
head -8 $O/stats/s_kernel_stats.csv | cut -c1-150
grep -E "k1_pairs|k2_tally" $O/t0_pmc_per_launch.csv $O/t1_pmc_per_launch.csv
