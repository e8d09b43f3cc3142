"""Build profiles/rN_pmc_fetch_write.json from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected in
separate runs as MI355X_MICROARCH.md prescribes) of the same bench.py command.
Usage: python tools/pmc_fetch_write.py FETCH_counter_collection.csv WRITE_counter_collection.csv K N_per_gpu out.json"""
import collections
import csv
import json
import sys

fetch_csv, write_csv, K, n_loc, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"].split("(")[0].strip()].append(float(r["Counter_Value"]))
    return agg


kernels = collections.defaultdict(dict)
for path, counter in ((fetch_csv, "FETCH_SIZE"), (write_csv, "WRITE_SIZE")):
    for name, vals in per_kernel(path, counter).items():
        # iterations enqueued past convergence by the device-resident loop are no-op launches (every kernel exits on the
        # stop flag): they are not part of a sweep's traffic, so launches far below the largest one are left out
        full = [v for v in vals if v >= 0.5 * max(vals)] if max(vals) > 0 else vals
        kernels[name][counter + "_KB_mean_per_launch"] = sum(full) / len(full)
        kernels[name][counter + "_launches"] = len(full)
        kernels[name][counter + "_noop_launches_left_out"] = len(vals) - len(full)
doc = {
    "_about": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python bench.py "
              "--steps 3 --warmup 1 --cpu-sample 0 --api-e2e 0` on MI355X; per-launch means in KB as reported.  Correction per "
              "MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE counts 1/2 of the bytes of a wide coalesced "
              "read -> hbm_read_bytes = 2*FETCH_SIZE*1024.  WRITE_SIZE is 1:1 (calibrated on k_generate_harmonic, "
              "which writes K*N*8 bytes).",
    "workload": {"K": K, "N_per_gpu": n_loc},
    "kernels": dict(sorted(kernels.items())),
}
json.dump(doc, open(out, "w"), indent=1)
print("wrote", out, "with", len(kernels), "kernels")
