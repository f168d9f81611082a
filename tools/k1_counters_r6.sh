set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6h
mkdir -p $O
python tools/k1_ab.py REO_K1_ORDER 0 1 t0 12 > $O/k1_order_ab.txt 2>&1
tail -4 $O/k1_order_ab.txt
for ord in 0 1; do
  export REO_K1_ORDER=$ord
  B="python3 tools/k1_only.py t0"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/o${ord}_fetch -o p -- $B > /dev/null 2> $O/o${ord}_fetch.log
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/o${ord}_write -o p -- $B > /dev/null 2> $O/o${ord}_write.log
  python3 profiles/summarize_pmc.py $O/order${ord}_pmc_per_launch.csv $O/o${ord}_fetch $O/o${ord}_write
  rm -rf $O/o${ord}_fetch $O/o${ord}_write
  grep k1w_pairs $O/order${ord}_pmc_per_launch.csv
done
unset REO_K1_ORDER
for job in "c3:tools/k1_only.py t0" "c4:tools/k1_shape.py 30000 4000 t0"; do
  name=${job%%:*}; B="python3 ${job#*:}"
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/${name}_sq -o p -- $B > /dev/null 2> $O/${name}_sq.log
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/${name}_sq2 -o p -- $B > /dev/null 2> $O/${name}_sq2.log
  python3 profiles/summarize_pmc.py $O/${name}_sq_per_launch.csv $O/${name}_sq $O/${name}_sq2
  rm -rf $O/${name}_sq $O/${name}_sq2
  grep k1w_pairs $O/${name}_sq_per_launch.csv
done
