#!/bin/bash
# A/B bench of library builds on one box: tools/ab.sh LIB_A LIB_B [LIB_C ...] [-- extra bench args]; round-robin, twice.
LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done; [ "$1" = "--" ] && shift
for i in 1 2; do
  for L in "${LIBS[@]}"; do
    MCPT_LIB=$L timeout -k 10 200 python bench.py --no-cpu-baseline --no-psnr "$@" > gpurun_out/ab_tmp.log 2>&1 || { tail -5 gpurun_out/ab_tmp.log; exit 1; }
    python - "$L" <<'PY'
import json,sys
l=[x for x in open('gpurun_out/ab_tmp.log') if x.startswith('{')][-1]
d=json.loads(l)
print(sys.argv[1], d['value'], {k:round(v) for k,v in d['roofline']['kernel_ms'].items()}, flush=True)
PY
  done
done
