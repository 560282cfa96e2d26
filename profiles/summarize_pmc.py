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
    for k, e in out.items():  # derived figures
        w = e.get("SQ_WAVES")
        if w:
            for c, name in (("SQ_INSTS_VALU", "valu_per_wave"), ("SQ_INSTS_SALU", "salu_per_wave")):
                if c in e:
                    e[name] = round(e[c] / w, 1)
        if e.get("SQ_WAVE_CYCLES"):
            for c, name in (("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_ACTIVE_INST_ANY", "active_frac")):
                if c in e:
                    e[name] = round(e[c] / e["SQ_WAVE_CYCLES"], 3)
        # FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled on gfx950 (MI355X_MICROARCH.md, HBM section)
        if "FETCH_SIZE" in e:
            e["hbm_read_GB_corrected"] = round(2 * e["FETCH_SIZE"] * 1024 / 1e9, 2)
        if "WRITE_SIZE" in e:
            e["hbm_write_GB"] = round(e["WRITE_SIZE"] * 1024 / 1e9, 2)
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e and e.get("launches"):
            e["hbm_bytes_per_launch"] = int((2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024 / e["launches"])
    return out


if __name__ == "__main__":
    print(json.dumps(main(sys.argv[1:]), indent=1))
