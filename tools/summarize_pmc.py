"""Summarise rocprofv3 output of `bench.py` for the committed profiles/ files.

    python tools/summarize_pmc.py --stats gpurun_out/prof --pmc gpurun_out/pmc/p1 gpurun_out/pmc/p2 ... \
        --kernel k_p1_rings --out profiles/r01_bench_pmc_summary.json --n 2236 --order 3

Per kernel instantiation whose name contains --kernel: mean of every collected counter per
dispatch, mean duration (kernel trace of the PMC passes), and the HBM traffic per launch by
the rule of MI355X_MICROARCH.md (HBM section): FETCH_SIZE (KB) counts half of the fetched
bytes on gfx950 -> doubled; WRITE_SIZE (KB) exact; cross-check TCC_MISS_sum x 128 B.
"""
import argparse
import csv
import glob
import json
import os
import sys
from collections import defaultdict

p = argparse.ArgumentParser()
p.add_argument("--pmc", nargs="+", required=True, help="output directories of the --pmc passes")
p.add_argument("--kernel", default="k_p1_rings")
p.add_argument("--out", required=True)
p.add_argument("--n", type=int, default=2236)
p.add_argument("--order", type=int, default=3)
p.add_argument("--command", default="")
p.add_argument("--trace", default=None, help="output directory of a --kernel-trace --stats run (no counters): "
               "steady-state durations = the steadiest stretch of 40 % of every kernel's dispatches")
args = p.parse_args()


def source_sha():
    """bench.source_sha: digest of the kernel sources and build flags the profile was taken from."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench

    return bench.source_sha()


counters = defaultdict(lambda: defaultdict(list))  # kernel -> counter -> values
durations = defaultdict(list)
waves = {}  # kernel -> waves of one dispatch (grid size / 64)
for d in args.pmc:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            seen = set()
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"]
                if args.kernel not in name:
                    continue
                counters[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
                waves[name] = int(row["Grid_Size"]) // 64
                key = (name, row["Dispatch_Id"])
                if key not in seen:
                    seen.add(key)
                    durations[name].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)

steady = {}
if args.trace:
    per_kernel = defaultdict(list)
    for path in glob.glob(os.path.join(args.trace, "**", "*kernel_trace.csv"), recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                if args.kernel in row["Kernel_Name"]:
                    per_kernel[row["Kernel_Name"]].append((int(row["Start_Timestamp"]), (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3))
    for name, rows in per_kernel.items():
        rows.sort()
        # steady state = the stretch of 40 % of the kernel's dispatches, consecutive in time, with the
        # lowest mean: the back-to-back launches of the timed region once the device has settled (the
        # first dispatches come out of an idle device; the sections of bench.py behind the timed region
        # -- single probes, API calls with host gaps, launches behind cache-flushing fills -- start
        # from a cooler device again: 190 -> 166 us over their first hundred launches)
        ds_time = [d for _, d in rows]
        width = max(1, int(0.4 * len(ds_time)))
        sums = [sum(ds_time[:width])]
        for i in range(1, len(ds_time) - width + 1):
            sums.append(sums[-1] - ds_time[i - 1] + ds_time[i + width - 1])
        first = min(range(len(sums)), key=sums.__getitem__)
        tail = sorted(ds_time[first:first + width])
        steady[name] = {"dispatches": len(rows), "tail_dispatches": len(tail), "window_first_dispatch": first,
                        "mean": sum(tail) / len(tail),
                        "median": tail[len(tail) // 2], "min": tail[0], "max": tail[-1],
                        "all_mean": sum(ds_time) / len(ds_time)}

out = {"command": args.command, "workload": {"n": args.n, "order": args.order}, "source_sha": source_sha(), "kernels": {}}
for name, cs in counters.items():
    entry = {"counters_per_dispatch": {c: {"dispatches": len(v), "mean": sum(v) / len(v)} for c, v in sorted(cs.items())}}
    ds = durations[name]
    entry["kernel_us_under_pmc"] = {"n": len(ds), "mean": sum(ds) / len(ds)}
    mean = lambda c: (sum(cs[c]) / len(cs[c])) if c in cs else None  # noqa: E731
    if mean("FETCH_SIZE") is not None and mean("WRITE_SIZE") is not None:
        fetch = 2.0 * mean("FETCH_SIZE") * 1024.0
        write = mean("WRITE_SIZE") * 1024.0
        entry["hbm_traffic_bytes_per_launch"] = {
            "rule": "MI355X_MICROARCH.md HBM section: FETCH_SIZE (KB) counts half of the fetched bytes "
                    "on gfx950 -> doubled; WRITE_SIZE (KB) exact",
            "fetch_bytes_corrected": fetch,
            "write_bytes": write,
            "total": fetch + write,
        }
        if mean("TCC_MISS_sum") is not None:
            entry["hbm_traffic_bytes_per_launch"]["cross_check_TCC_MISS_x128B"] = mean("TCC_MISS_sum") * 128.0
    if name in steady:
        entry["kernel_us_steady"] = steady[name]  # un-profiled-counter run, steadiest 40 % stretch of the dispatches
    if mean("SQ_ACTIVE_INST_VALU") is not None:
        # SQ_ACTIVE_INST_VALU counts quad-cycles with a vector instruction in a SIMD's pipe, summed over
        # the waves: x 4 / 1024 SIMDs = cycles a SIMD's vector pipe is busy per dispatch.  Against the
        # kernel's length at the clock the chip holds (2.26-2.40 GHz by in-kernel s_memtime /
        # s_memrealtime stamps, profiles/r03_wave_loop_spread.log; GRBM_GUI_ACTIVE / 8 reads high on
        # dispatches this short, SQ_WAVE_CYCLES does not cover a wave's whole life).
        busy = mean("SQ_ACTIVE_INST_VALU") * 4.0 / 1024.0
        entry["valu_busy_cycles_per_simd"] = busy
        if name in steady:
            entry["valu_utilisation"] = busy / (steady[name]["mean"] * 1e3 * 2.35)
            entry["valu_utilisation_note"] = "busy cycles / (steady kernel time x 2.35 GHz, the in-kernel clock)"
    out["kernels"][name] = entry
with open(args.out, "w") as fh:
    json.dump(out, fh, indent=1)
print(json.dumps({k: v.get("hbm_traffic_bytes_per_launch", {}).get("total") for k, v in out["kernels"].items()}, indent=1))
