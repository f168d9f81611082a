# Round-4 profiles (run on the GPU box through gpurun): kernel stats of the default bench, PMC passes (FETCH_SIZE / WRITE_SIZE / SQ
# counters in separate passes) for K1 on both families and for the tally scan, kernel stats of the shapes that had none.
# Summaries land in gpurun_out/$1 (default r4p); copy what is kept to profiles/.
set -e
TAG=${1:-r4p}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.log
cp $(find $O/stats -name "s_kernel_stats.csv" | head -1) $O/kernel_stats.csv
for fam in t0 t1; do
  B="python3 tools/k1_only.py $fam"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${fam}_fetch -o p -- $B > /dev/null 2> $O/${fam}_fetch.log
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${fam}_write -o p -- $B > /dev/null 2> $O/${fam}_write.log
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/${fam}_sq -o p -- $B > /dev/null 2> $O/${fam}_sq.log
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O/${fam}_sq2 -o p -- $B > /dev/null 2> $O/${fam}_sq2.log
  python3 profiles/summarize_pmc.py $O/${fam}_pmc_per_launch.csv $O/${fam}_fetch $O/${fam}_write $O/${fam}_sq $O/${fam}_sq2
done
B="python3 tools/k2_only.py"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/k2_fetch -o p -- $B > /dev/null 2> $O/k2_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/k2_write -o p -- $B > /dev/null 2> $O/k2_write.log
python3 profiles/summarize_pmc.py $O/k2_pmc_per_launch.csv $O/k2_fetch $O/k2_write
for job in "g70000:tools/k1_shape.py 70000 24 t1" "g140000:tools/k1_shape.py 140000 16 t0" "s66100:tools/k1_shape.py 260 66100 t1" "float:tools/k1_shape.py 20000 1000 float"; do
  name=${job%%:*}; cmd=${job#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o s -- python3 $cmd > $O/$name.txt 2> $O/$name.log
  cp $(find $O/$name -name "s_kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv
  tail -1 $O/$name.txt
done
head -8 $O/kernel_stats.csv | cut -c1-150
grep -E "k1w?_pairs|k2_" $O/t0_pmc_per_launch.csv $O/t1_pmc_per_launch.csv $O/k2_pmc_per_launch.csv | grep -E "FETCH|WRITE|GRBM"
