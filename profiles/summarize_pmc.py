"""Summarises rocprofv3 --pmc counter_collection CSVs per kernel (sums over dispatches, per-launch averages).
Usage: python profiles/summarize_pmc.py <counter_collection.csv> [...]"""
import collections
import csv
import json
import sys


def short(name):
    if "k_trace<true" in name or "k_trace_shadow" in name:
        return "k_trace<shadow>"
    if "k_trace<false" in name or "k_trace_closest" in name:
        return "k_trace<closest>"
    for k in ("k_shade", "k_direct", "k_primary", "k_accumulate", "k_init_free", "k_generate_explicit"):
        if k in name:
            return k
    return None


def main(paths):
    out = {}
    for path in paths:
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        launches = collections.defaultdict(set)
        dur = collections.defaultdict(float)
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            if not k:
                continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in launches[k]:
                launches[k].add(r["Dispatch_Id"])
                dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        for k, v in agg.items():
            e = out.setdefault(k, {})
            e["launches"] = len(launches[k])
            e.setdefault("ms_profiled", {})[",".join(sorted(v))[:40]] = round(dur[k], 3)
            for c, x in v.items():
                e[c] = x
    return out


if __name__ == "__main__":
    print(json.dumps(main(sys.argv[1:]), indent=1))
