#!/bin/bash
# A/B bench of environment knobs on one box: tools/ab_env.sh "ENV SETTINGS A" "ENV SETTINGS B" ... [-- extra bench args]; round-robin, twice.
# ("" = the defaults; MCPT_LIB=... in a setting selects a library build)
SETS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do SETS+=("$1"); shift; done; [ "$1" = "--" ] && shift
for i in 1 2; do
  for E in "${SETS[@]}"; do
    env $E timeout -k 10 200 python bench.py --no-cpu-baseline --no-psnr "$@" > gpurun_out/ab_tmp.log 2>&1 || { tail -5 gpurun_out/ab_tmp.log; exit 1; }
    python - "$E" <<'PY'
import json,sys
l=[x for x in open('gpurun_out/ab_tmp.log') if x.startswith('{')][-1]
d=json.loads(l)
print("%-60s" % (sys.argv[1] or "(defaults)"), d['value'], {k:round(v) for k,v in d['roofline']['kernel_ms'].items()}, flush=True)
PY
  done
done
