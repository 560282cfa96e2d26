#!/usr/bin/env python3
"""Builds profiles/traffic.json from the outputs of tools/profile_round.sh.  Per profiled configuration (the SERIALISED run `bench.py
--serialized --steps 1 --warmup 0 --spp-per-step 64 [scene arguments]`, kernels on one stream so that durations and counters attribute
cleanly):

  kernels.<kernel>  HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 / launches   (separate --pmc passes; FETCH_SIZE
                    doubled per MI355X_MICROARCH.md: gfx950 tallies 128-B requests at 64 B)
  serialized        per kernel: launches, avg_launch_ms (rocprofv3 --kernel-trace --stats), ms_per_step, VALU instructions per
                    wave, valu_busy_frac = SQ_INSTS_VALU * 4 cycles / (1024 SIMDs * 2.4 GHz) / kernel time, wait_any_frac
  job               HBM bytes per sample over all kernels of the step

and, at the top level, the source tag of the build the profile was taken on (bench.py applies an entry only to that build rendering that
configuration).

Usage: tools/make_profile_summary.py gpurun_out/prof_TAG TAG   (writes profiles/traffic.json and copies the summaries to profiles/)"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "profiles"))
sys.path.insert(0, ROOT)
from summarize_pmc import short  # noqa: E402

SIMDS, CLOCK_HZ, CYCLES_PER_VALU = 1024, 2.4e9, 4
CONFIGS = {"chess": ("chess", 1920, 1080), "cornell_rc": ("cornell_rc", 784, 784), "cornell_demo": ("cornell_demo", 1920, 1080),
           "chess_high": ("chess_high", 1920, 1080)}


def one_config(d, name, tag):
    scene, W, H = CONFIGS[name]
    pmc = json.load(open(os.path.join(d, "%s_pmc_summary_no_overlap_64spp_step.json" % name)))
    bench = json.loads(open(os.path.join(d, "%s_bench_serialized_64spp.json" % name)).read().strip().splitlines()[-1])
    samples = W * H * 64
    stats = {}
    for r in csv.DictReader(open(os.path.join(d, "%s_kernel_stats_no_overlap_64spp.csv" % name))):
        k = short(r["Name"])
        if not k:
            continue
        e = stats.setdefault(k, {"launches": 0, "ms": 0.0})
        e["launches"] += int(r["Calls"])
        e["ms"] += float(r["TotalDurationNs"]) / 1e6
    kernels, ser, total_bytes = {}, {}, 0.0
    for k, e in pmc.items():
        if "hbm_bytes_per_launch" in e:
            kernels[k] = e["hbm_bytes_per_launch"]
            total_bytes += e["hbm_bytes_per_launch"] * e["launches"]
        st = stats.get(k)
        if not st or k in ("k_init_free",):
            continue
        row = {"launches": st["launches"], "avg_launch_ms": round(st["ms"] / st["launches"], 4), "ms_per_step": round(st["ms"], 3),
               "hbm_bytes_per_launch": e.get("hbm_bytes_per_launch"), "valu_per_wave": e.get("valu_per_wave"),
               "wait_any_frac": e.get("wait_any_frac")}
        if "SQ_INSTS_VALU" in e:
            t = [v for c, v in e["ms_profiled"].items() if "SQ_INSTS_VALU" in c][0]
            row["valu_busy_frac"] = round(e["SQ_INSTS_VALU"] * CYCLES_PER_VALU / (SIMDS * CLOCK_HZ) / (t * 1e-3), 3)
        ku, kl = bench["roofline"].get("kernel_units", {}), bench["roofline"].get("kernel_launches", {})
        if ku.get(k) and kl.get(k):
            row["units_per_launch"] = round(ku[k] / kl[k], 1)
            if row["hbm_bytes_per_launch"]:
                row["hbm_bytes_per_unit"] = round(row["hbm_bytes_per_launch"] / row["units_per_launch"], 1)
        if row["hbm_bytes_per_launch"]:
            row["hbm_GBps"] = round(row["hbm_bytes_per_launch"] / (row["avg_launch_ms"] * 1e-3) / 1e9, 1)
        ser[k] = row
    return {"config": {"scene": scene, "width": W, "height": H, "n_dir": 4, "workload": bench.get("config", {}).get("workload", "")},
            "kernels": kernels, "serialized": ser,
            "job": {"hbm_bytes_per_sample": round(total_bytes / samples, 1), "samples_in_step": samples,
                    "source": "profiles/%s_%s_pmc_summary_no_overlap_64spp_step.json" % (tag, name)},
            "bench_serialized": {"value": bench["value"], "kernel_ms": bench["roofline"]["kernel_ms"], "kernel_launches": bench["roofline"].get("kernel_launches")}}


def main(d, tag):
    import mcpt_loader
    out = {"_build_tag": mcpt_loader.load().build.source_tag(), "_profile": tag, "configs": {}}
    for name in CONFIGS:
        if os.path.exists(os.path.join(d, "%s_pmc_summary_no_overlap_64spp_step.json" % name)):
            out["configs"][name] = one_config(d, name, tag)
    out["_note"] = ("Serialised runs (`bench.py --serialized --steps 1 --warmup 0 --spp-per-step 64 [scene]`, build %s = source tag %s, "
                    "tools/profile_round.sh): HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 / launches from separate rocprofv3 --pmc passes; "
                    "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B for 16 B/lane streams; narrower accesses are "
                    "uncalibrated)." % (tag, out["_build_tag"]))
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    copies = [("kernel_stats_overlap.csv", "%s_kernel_stats_overlap.csv"), ("trace_busy_overlap.txt", "%s_timeline_overlap.txt"),
              ("bench.json", "%s_bench.json"), ("bench_profiled_overlap.json", "%s_bench_profiled_overlap.json")]
    for name in CONFIGS:
        copies += [("%s_kernel_stats_no_overlap_64spp.csv" % name, "%%s_%s_kernel_stats_no_overlap_64spp_step.csv" % name),
                   ("%s_pmc_summary_no_overlap_64spp_step.json" % name, "%%s_%s_pmc_summary_no_overlap_64spp_step.json" % name),
                   ("%s_bench_serialized_64spp.json" % name, "%%s_%s_bench_serialized_64spp.json" % name)]
    for f in sorted(os.listdir(d)):
        if f.startswith("bench_") and f.endswith(".json") and f != "bench_profiled_overlap.json":
            copies.append((f, "%s_" + f))
    for src, dst in copies:
        if os.path.exists(os.path.join(d, src)):
            shutil.copy(os.path.join(d, src), os.path.join(ROOT, "profiles", dst % tag))
    for name, c in out["configs"].items():
        print(name, json.dumps({k: {f: v.get(f) for f in ("ms_per_step", "valu_per_wave", "valu_busy_frac", "wait_any_frac", "hbm_bytes_per_unit", "hbm_GBps")}
                                for k, v in c["serialized"].items()}, indent=1))
        print(name, json.dumps(c["job"]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
