# Round 6, verdict item 7: the order of the pair kernel's items and its write traffic (mirror words leave L2 as partial lines).
# REO_K1_ORDER 0 (i-tile-major, chunks fastest: rounds 3-5), 1 (chunk-major, i-tiles fastest), 2 (blocks of i-tiles; inside a block chunk by chunk).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6j
mkdir -p $O
tools/bin/mb_issue > $O/microbench_issue.txt 2>&1
cat $O/microbench_issue.txt
python tools/k1_ab.py REO_K1_ORDER 0 2 t0 12 2>&1 | tail -4 > $O/k1_order_0_2_ab.txt
cat $O/k1_order_0_2_ab.txt
for ord in 0 2; do
  export REO_K1_ORDER=$ord
  for job in "c3:tools/k1_only.py t0" "t1:tools/k1_only.py t1" "c4:tools/k1_shape.py 30000 4000 t0"; do
    name=${job%%:*}; B="python3 ${job#*:}"
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -o p -- $B > /dev/null 2> $O/f.log
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -o p -- $B > /dev/null 2> $O/w.log
    python3 profiles/summarize_pmc.py $O/order${ord}_${name}_pmc_per_launch.csv $O/f $O/w
    rm -rf $O/f $O/w
    grep "k1w_pairs," $O/order${ord}_${name}_pmc_per_launch.csv | sed "s/^/order $ord $name: /"
  done
done
unset REO_K1_ORDER
for o in 0 2 0 2; do REO_K1_ORDER=$o python3 tools/k1_shape.py 30000 4000 t0 2>&1 | tail -1 | sed "s/^/order $o: /"; done
python tools/k1_ab.py REO_K1_ORDER 0 2 t1 8 2>&1 | tail -4
for o in 0 2; do REO_K1_ORDER=$o python3 tools/k1_shape.py 70000 1000 t0 2>&1 | tail -1 | sed "s/^/order $o: /"; done
for o in 0 2; do REO_K1_ORDER=$o python3 tools/k1_shape.py 5000 200 t1 2>&1 | tail -1 | sed "s/^/order $o: /"; done
