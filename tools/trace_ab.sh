#!/bin/bash
# Kernel traces of two library builds side by side: tools/trace_ab.sh NAME_A LIB_A NAME_B LIB_B [-- bench args]
# (per build: rocprofv3 kernel statistics and tools/trace_busy.py's timeline summary under gpurun_out/trace_ab/)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/trace_ab; mkdir -p $O
NA=$1; LA=$(realpath $2); NB=$3; LB=$(realpath $4); shift 4; [ "$1" = "--" ] && shift
cd /tmp && export TMPDIR=/tmp
EXTRA="$*"
for pair in "$NA $LA" "$NB $LB"; do
  N=${pair%% *}; export MCPT_LIB=${pair#* }
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$N -- python3 $R/bench.py --warmup 0 --steps 2 --no-cpu-baseline --no-psnr $EXTRA > $O/$N.log 2>&1 || { tail -5 $O/$N.log; exit 1; }
  f=$(find $O/kt_$N -name "*kernel_stats.csv" | head -1); cp $f $O/${N}_kernel_stats.csv
  t=$(find $O/kt_$N -name "*kernel_trace.csv" | head -1); python3 $R/tools/trace_busy.py $t 0.3 > $O/${N}_busy.txt; python3 $R/tools/trace_iter.py $t > $O/${N}_iter.txt
  rm -rf $O/kt_$N
  grep "^{" $O/$N.log | tail -1 | cut -c1-200
done
