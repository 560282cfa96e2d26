#!/bin/bash
# tools/abargs.sh "ARGS1" "ARGS2" ... : runs bench.py once per argument string, round-robin, twice.
for i in 1 2; do
  for A in "$@"; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-psnr $A > gpurun_out/ab_tmp.log 2>&1 || { tail -5 gpurun_out/ab_tmp.log; exit 1; }
    python - "$A" <<'PY'
import json,sys
l=[x for x in open('gpurun_out/ab_tmp.log') if x.startswith('{')][-1]
d=json.loads(l)
print(sys.argv[1], d['value'], d['job']['wavefront_iterations'], {k:round(v) for k,v in d['roofline']['kernel_ms'].items()}, flush=True)
PY
  done
done
