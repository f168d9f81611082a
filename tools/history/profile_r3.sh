# Round-3 profiles of the hot path (run on the GPU box through gpurun): kernel stats of the default bench, and PMC
# passes (FETCH_SIZE / WRITE_SIZE / SQ counters in separate passes) for the tie-free and the tie-rich family.
# Summaries land in gpurun_out/$1 (default r3a); copy what is kept to profiles/.
set -e
TAG=${1:-r3a}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.log
for fam in t0 t1; do
  B="python3 tools/k1_only.py $fam"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${fam}_fetch -o p -- $B > /dev/null 2> $O/${fam}_fetch.log
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${fam}_write -o p -- $B > /dev/null 2> $O/${fam}_write.log
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/${fam}_sq -o p -- $B > /dev/null 2> $O/${fam}_sq.log
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O/${fam}_sq2 -o p -- $B > /dev/null 2> $O/${fam}_sq2.log
  python3 profiles/summarize_pmc.py $O/${fam}_pmc_per_launch.csv $O/${fam}_fetch $O/${fam}_write $O/${fam}_sq $O/${fam}_sq2
done
cp $O/stats/*/s_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null || cp $(find $O/stats -name "s_kernel_stats.csv" | head -1) $O/kernel_stats.csv
head -8 $O/kernel_stats.csv | cut -c1-150
grep -E "k1w?_pairs|k2_tally" $O/t0_pmc_per_launch.csv $O/t1_pmc_per_launch.csv
