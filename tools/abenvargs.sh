#!/bin/bash
# tools/abenvargs.sh "BENCH ARGS" "ENV1=a" "ENV1=b" ... : same bench arguments under several environments, round-robin, twice.
ARGS=$1; shift
for i in 1 2; do
  for E in "$@"; do
    env $E timeout -k 10 300 python bench.py --no-cpu-baseline --no-psnr $ARGS > gpurun_out/ab_tmp.log 2>&1 || { tail -5 gpurun_out/ab_tmp.log; exit 1; }
    python - "$E" <<'PY'
import json,sys
l=[x for x in open('gpurun_out/ab_tmp.log') if x.startswith('{')][-1]
d=json.loads(l)
print(sys.argv[1], d['value'], {k:round(v) for k,v in d['roofline']['kernel_ms'].items()}, flush=True)
PY
  done
done
