#!/bin/bash
# Collects the profiles of a round on the GPU box: tools/profile_round.sh TAG  (outputs under gpurun_out/prof_TAG/)
#   chess 1080p (the headline configuration): bench line, rocprofv3 kernel trace + statistics of the default (overlapped) run, and of the
#   SERIALISED 64-spp step with separate --pmc passes (HBM bytes, instruction counts, wait fractions);
#   cornell_rc 784^2 (config 2), cornell_demo 1080p, chess_high: the serialised step with the same --pmc passes;
#   bench lines of the other BASELINE configurations.
# tools/make_profile_summary.py turns the result into profiles/traffic.json (one entry per profiled configuration) and copies the summaries.
set -o pipefail
TAG=${1:-vX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err || exit 1
echo "bench done"; tail -c 300 $O/bench.json; echo
# (--warmup 0: every launch of the profiled process lies in bench.py's timed region, so the per-kernel averages of its JSON line and of
# the rocprofv3 statistics are averages over the same launches)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --warmup 0 --no-cpu-baseline --no-psnr > $O/kt.log 2>&1 || exit 1
grep "^{" $O/kt.log | tail -1 > $O/bench_profiled_overlap.json
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_overlap.csv
t=$(find $O/kt -name "*kernel_trace.csv" | head -1); python3 $R/tools/trace_busy.py $t 0.3 > $O/trace_busy_overlap.txt
rm -rf $O/kt; echo "kernel trace done"
SMALL="--serialized --steps 1 --warmup 0 --spp-per-step 64 --no-cpu-baseline --no-psnr"
# name | bench arguments of the configuration
profile_config() {
  local NAME=$1; shift
  local ARGS="$* $SMALL"
  python3 $R/bench.py $ARGS > $O/${NAME}_bench_serialized_64spp.json 2>> $O/bench.err || return 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt0 -- python3 $R/bench.py $ARGS > $O/kt0.log 2>&1 || return 1
  f=$(find $O/kt0 -name "*kernel_stats.csv" | head -1); cp $f $O/${NAME}_kernel_stats_no_overlap_64spp.csv; rm -rf $O/kt0
  for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
    n=$(echo $C | tr ' ' '_')
    rocprofv3 --pmc $C --output-format csv -d $O/pmc_$n -- python3 $R/bench.py $ARGS > $O/pmc_$n.log 2>&1 || { tail -3 $O/pmc_$n.log; return 1; }
    f=$(find $O/pmc_$n -name "*counter_collection.csv" | head -1); cp $f $O/pmc_$n.csv; rm -rf $O/pmc_$n
  done
  python3 $R/profiles/summarize_pmc.py $O/pmc_*.csv > $O/${NAME}_pmc_summary_no_overlap_64spp_step.json
  rm -f $O/pmc_*.csv
  echo "config $NAME profiled"
}
profile_config chess || exit 1
profile_config cornell_rc --scene cornell_rc --width 784 --height 784 || exit 1
profile_config cornell_demo --scene cornell_demo || exit 1
profile_config chess_high --scene chess_high || exit 1
# the other BASELINE configurations (bench lines only)
python3 $R/bench.py --scene cornell_rc --width 784 --height 784 --steps 1 --spp-per-step 256 --cpu-spp 16 > $O/bench_config2_cornell_rc_784_spp256.json 2>> $O/bench.err || exit 1
python3 $R/bench.py --steps 2 --no-cpu-baseline > $O/bench_config3_chess_spp512.json 2>> $O/bench.err || exit 1
python3 $R/bench.py --n-dir 32 --no-cpu-baseline > $O/bench_config4_chess_spp2048_ndir32.json 2>> $O/bench.err || exit 1
python3 $R/bench.py --scene cornell_demo --no-cpu-baseline > $O/bench_cornell_demo_1080p.json 2>> $O/bench.err || exit 1
python3 $R/bench.py --scene chess_high --no-cpu-baseline > $O/bench_chess_high_sah.json 2>> $O/bench.err || exit 1
MCPT_BVH=ploc python3 $R/bench.py --scene chess_high --no-cpu-baseline > $O/bench_chess_high_ploc.json 2>> $O/bench.err || exit 1
MCPT_BVH=ploc python3 $R/bench.py --no-cpu-baseline > $O/bench_chess_ploc.json 2>> $O/bench.err || exit 1
MCPT_BVH=lbvh python3 $R/bench.py --no-cpu-baseline > $O/bench_chess_lbvh.json 2>> $O/bench.err || exit 1
echo "config benches done"
ls $O
