#!/usr/bin/env python3
"""Builds profiles/traffic.json from the outputs of tools/profile_round.sh (the SERIALISED run: `bench.py --serialized --steps 1
--warmup 0 --spp-per-step 64`, kernels on one stream so that durations and counters attribute cleanly):

  <kernel>          HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 / launches   (separate --pmc passes; FETCH_SIZE
                    doubled per MI355X_MICROARCH.md: gfx950 tallies 128-B requests at 64 B)
  _serialized       per kernel: launches, avg_launch_ms (rocprofv3 --kernel-trace --stats), ms_per_step, VALU instructions per
                    wave, valu_busy_frac = SQ_INSTS_VALU * 4 cycles / (1024 SIMDs * 2.4 GHz) / kernel time, wait_any_frac
  _job              HBM bytes per sample over all kernels of the step

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


def main(d, tag):
    pmc = json.load(open(os.path.join(d, "pmc_summary_no_overlap_64spp_step.json")))
    bench = json.loads(open(os.path.join(d, "bench_serialized_64spp.json")).read().strip().splitlines()[-1])
    samples = 1920 * 1080 * 64
    stats = {}
    for r in csv.DictReader(open(os.path.join(d, "kernel_stats_no_overlap_64spp.csv"))):
        k = short(r["Name"])
        if not k:
            continue
        e = stats.setdefault(k, {"launches": 0, "ms": 0.0})
        e["launches"] += int(r["Calls"])
        e["ms"] += float(r["TotalDurationNs"]) / 1e6
    out, ser, total_bytes = {}, {}, 0.0
    for k, e in pmc.items():
        if "hbm_bytes_per_launch" in e:
            out[k] = e["hbm_bytes_per_launch"]
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
    out["_serialized"] = ser
    out["_job"] = {"hbm_bytes_per_sample": round(total_bytes / samples, 1), "samples_in_step": samples,
                   "source": "profiles/%s_pmc_summary_no_overlap_64spp_step.json" % tag}
    cfg = bench.get("config", {}).get("workload", "")
    import mcpt_loader
    out["_build_tag"] = mcpt_loader.load().build.source_tag()  # bench.py applies this profile only to the build it was measured on ...
    out["_config"] = {"scene": "chess", "width": 1920, "height": 1080, "n_dir": 4, "workload": cfg}  # ... rendering this configuration (tools/profile_round.sh)
    out["_bench_serialized"] = {"value": bench["value"], "kernel_ms": bench["roofline"]["kernel_ms"], "kernel_launches": bench["roofline"].get("kernel_launches")}
    out["_note"] = ("Serialised run (`bench.py --serialized --steps 1 --warmup 0 --spp-per-step 64`, build %s, tools/profile_round.sh): "
                    "HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 / launches from separate rocprofv3 --pmc passes; FETCH_SIZE doubled "
                    "per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B for 16 B/lane streams; narrower accesses are uncalibrated). "
                    "Launch sizes in the default run (pool-limited) are the same as in this run." % tag)
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    for src, dst in (("kernel_stats_no_overlap_64spp.csv", "%s_kernel_stats_no_overlap_64spp_step.csv"), ("kernel_stats_overlap.csv", "%s_kernel_stats_overlap.csv"),
                     ("pmc_summary_no_overlap_64spp_step.json", "%s_pmc_summary_no_overlap_64spp_step.json"), ("trace_busy_overlap.txt", "%s_timeline_overlap.txt"),
                     ("bench.json", "%s_bench.json"), ("bench_profiled_overlap.json", "%s_bench_profiled_overlap.json"), ("bench_serialized_64spp.json", "%s_bench_serialized_64spp.json"),
                     ("bench_config2_cornell_rc_784_spp256.json", "%s_bench_config2_cornell_rc_784_spp256.json"),
                     ("bench_config3_chess_spp512.json", "%s_bench_config3_chess_spp512.json"),
                     ("bench_config4_chess_spp2048_ndir32.json", "%s_bench_config4_chess_spp2048_ndir32.json"),
                     ("bench_cornell_demo_1080p.json", "%s_bench_cornell_demo_1080p.json"),
                     ("bench_chess_high_sah.json", "%s_bench_chess_high_sah.json"), ("bench_chess_high_lbvh.json", "%s_bench_chess_high_lbvh.json")):
        if os.path.exists(os.path.join(d, src)):
            shutil.copy(os.path.join(d, src), os.path.join(ROOT, "profiles", dst % tag))
    print(json.dumps(out["_serialized"], indent=1))
    print(json.dumps(out["_job"]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
