# Kernel-stats evidence for the shapes beside the headline one (round 3): config 4, config 5, ten groups.
# usage (on the GPU box): bash tools/profile_shapes_r3.sh [TAG]; summaries land in gpurun_out/TAG, copy what is kept to profiles/.
set -e
TAG=${1:-r3s}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
for job in "config4:tools/config4.py" "config5:tools/config5.py" "multigroup10:tools/multigroup.py 10 t0"; do
  name=${job%%:*}; cmd=${job#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o s -- python3 $cmd > $O/$name.txt 2> $O/$name.log
  cp $(find $O/$name -name "s_kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv
  head -6 $O/${name}_kernel_stats.csv | cut -c1-160
  tail -4 $O/$name.txt
done
