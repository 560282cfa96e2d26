#!/usr/bin/env python3
"""Reads a rocprofv3 kernel-trace CSV and reports how busy the GPU was: wall span of the traced kernels, the union
of their intervals (time with at least one kernel running), the idle gaps, and per-kernel summed / exclusive time."""
import csv, re, sys, collections

def short(name):
    m = re.search(r"(k_\w+?)(?:IL[ij](\d+)E)?(?:<|\(|E|$)", name)
    if m:
        return m.group(1)
    return name[:40]


def main(path, skip_frac=0.0):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    lo = t0 + (t1 - t0) * skip_frac
    rows = [r for r in rows if r[0] >= lo]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    ev = []
    for s, e, n in rows:
        ev.append((s, 1, n)); ev.append((e, -1, n))
    ev.sort()
    active = collections.Counter(); nact = 0; last = ev[0][0]
    busy = 0; conc = collections.Counter(); excl = collections.Counter(); gaps = []
    for t, d, n in ev:
        dt = t - last
        if dt > 0:
            if nact > 0:
                busy += dt; conc[nact] += dt
                if nact == 1:
                    excl[[k for k, v in active.items() if v > 0][0]] += dt
            else:
                gaps.append(dt)
        active[n] += d; nact += d; last = t
    tot = collections.Counter(); cnt = collections.Counter()
    for s, e, n in rows:
        tot[n] += e - s; cnt[n] += 1
    span = t1 - t0
    print("span %.1f ms, busy %.1f ms (%.1f %%), idle %.1f ms in %d gaps (median %.1f us, max %.1f us)" % (
        span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6, len(gaps),
        sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0, max(gaps) / 1e3 if gaps else 0))
    print("concurrency: " + ", ".join("%d kernels %.1f ms" % (k, v / 1e6) for k, v in sorted(conc.items())))
    for n, v in tot.most_common(12):
        print("  %-60s n=%5d sum %8.1f ms avg %8.1f us  alone %8.1f ms" % (n, cnt[n], v / 1e6, v / 1e3 / cnt[n], excl[n] / 1e6))

if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.0)
