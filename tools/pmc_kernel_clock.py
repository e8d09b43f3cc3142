"""Join a rocprofv3 --pmc counter_collection.csv with the kernel_trace.csv of the SAME run: for every kernel whose name
contains the given substring, mean duration and mean counters over its full launches (no-op launches of the device-resident
loop, a few microseconds each, are dropped), and the effective shader clock GRBM_GUI_ACTIVE / 8 XCDs / duration.

    python tools/pmc_kernel_clock.py counter_collection.csv kernel_trace.csv k_fused [min_us] [--json out.json]

With --json the clock (and the matrix-pipe share, when SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE are there) is
also written as a small record: bench.py quotes the latest profiles/r*_pmc_fused_clock.json as `roofline.frac_at_measured_clock`.
"""
import collections
import csv
import sys


def main():
    argv = list(sys.argv)
    json_out = None
    if "--json" in argv:
        i = argv.index("--json")
        json_out = argv[i + 1]
        del argv[i:i + 2]
    ctr_csv, trace_csv, needle = argv[1:4]
    min_us = float(argv[4]) if len(argv) > 4 else 100.0
    dur = {}
    for r in csv.DictReader(open(trace_csv)):
        if needle in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    vals = collections.defaultdict(dict)
    for r in csv.DictReader(open(ctr_csv)):
        if needle in r["Kernel_Name"]:
            d = vals[r["Dispatch_Id"]]
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    full = [i for i, us in dur.items() if us >= min_us and i in vals]
    if not full:
        print("no full launches of", needle)
        return
    mean_us = sum(dur[i] for i in full) / len(full)
    print(f"{needle}: {len(full)} full launches (of {len(dur)}), mean {mean_us:.1f} us, min {min(dur[i] for i in full):.1f}, max {max(dur[i] for i in full):.1f}")
    names = sorted({c for i in full for c in vals[i]})
    means = {c: sum(vals[i].get(c, 0.0) for i in full) / len(full) for c in names}
    for c in names:
        print(f"   {c:28s} {means[c]:.5g}")
    if "GRBM_GUI_ACTIVE" in means:
        ghz = means["GRBM_GUI_ACTIVE"] / 8 / mean_us / 1e3
        print(f"   effective clock = GRBM_GUI_ACTIVE / 8 / duration = {ghz:.3f} GHz")
        if json_out:
            import json

            rec = {"kernel": needle, "full_launches": len(full), "mean_us": mean_us, "effective_clock_GHz": ghz,
                   "how": "rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE ...: GRBM_GUI_ACTIVE / 8 XCDs / kernel duration (tools/pmc_kernel_clock.py)",
                   "counters_mean_per_launch": means}
            json.dump(rec, open(json_out, "w"), indent=1)


if __name__ == "__main__":
    main()
