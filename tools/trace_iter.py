#!/usr/bin/env python3
"""Per-iteration timeline of the wavefront loop from a rocprofv3 kernel-trace CSV: for every k_shade launch, the kernels that follow it up to
the next k_shade, as offsets from k_shade's start (medians over the steady iterations, in microseconds)."""
import csv, sys, statistics as st
from trace_busy import short

def main(path, skip_frac=0.3):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    rows = [r for r in rows if r[0] >= t0 + (t1 - t0) * skip_frac]
    shades = [i for i, r in enumerate(rows) if r[2] == "k_shade"]
    per = {}
    period = []
    for a, b in zip(shades[:-1], shades[1:]):
        s0 = rows[a][0]
        period.append(rows[b][0] - s0)
        seen = set()
        for s, e, n in rows[a:b]:
            if n in seen:
                continue
            seen.add(n)
            per.setdefault(n, []).append((s - s0, e - s0))
        # k_primary of this iteration usually STARTS before k_shade: look back
        for s, e, n in rows[max(0, a - 4):a]:
            if n == "k_primary" and e > s0 and n not in seen:
                per.setdefault(n, []).append((s - s0, e - s0))
    print("iterations %d, period median %.1f us (mean %.1f)" % (len(period), st.median(period) / 1e3, st.mean(period) / 1e3))
    for n, v in sorted(per.items(), key=lambda kv: st.median(x[0] for x in kv[1])):
        print("  %-28s n=%4d start %8.1f  end %8.1f  dur %8.1f" % (n, len(v), st.median(x[0] for x in v) / 1e3, st.median(x[1] for x in v) / 1e3,
                                                                  st.median(x[1] - x[0] for x in v) / 1e3))

if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.3)
