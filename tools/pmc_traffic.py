#!/usr/bin/env python3
"""HBM-side traffic of one CSR SpMV launch from the two rocprofv3 --pmc passes over
tools/pmc_spmv.py (FETCH_SIZE and WRITE_SIZE, each its own run):

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> \
        [placement text] [sliced_algorithmic_bytes]

FETCH_SIZE is reported in KiB and tallies a 128-byte read request as 64 bytes on gfx950; the dot
product in the same run (2 * rows * 8 bytes read, nothing else) calibrates that factor."""
import csv
import json
import sys


def means(path, counter):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        key = ("spmv_sliced" if ("sliced_spmv" in name or "sliced_wave" in name) else
               "spmv_c32" if "csr_spmv" in name and "unsigned int" in name else
               "spmv" if "csr_spmv" in name else "dot" if "dot_kernel" in name else None)
        if key:
            acc.setdefault(key, []).append(float(r["Counter_Value"]))
    # the last three launches of each kind are the measured ones (earlier SpMV launches belong
    # to the set-up-time placement tuning)
    return {k: sum(v[-3:]) / len(v[-3:]) for k, v in acc.items()}


def main():
    fetch, write, out = sys.argv[1:4]
    placement = sys.argv[4] if len(sys.argv) > 4 else None
    rows, nnz = 512 ** 3, (3 * 512 - 2) ** 3
    f, w = means(fetch, "FETCH_SIZE"), means(write, "WRITE_SIZE")
    dot_true = 2 * rows * 8
    ratio = dot_true / (f["dot"] * 1024)
    # 2.0 for wide coalesced reads on gfx950 (a 128-byte request tallied as 64 bytes), 1.0 where the
    # counter reads true; anything else means the calibration stream is not understood: stop
    if 1.9 < ratio < 2.1:
        factor = 2
    elif 0.95 < ratio < 1.05:
        factor = 1
    else:
        raise SystemExit(f"FETCH_SIZE calibration ratio {ratio:.3f} is neither 1 nor 2")
    traffic = factor * f["spmv"] * 1024 + w["spmv"] * 1024
    algo = nnz * 16 + rows * 32
    c32 = {}
    if "spmv_c32" in f:  # the form on the set-up-time 32-bit structure
        t32, a32 = factor * f["spmv_c32"] * 1024 + w["spmv_c32"] * 1024, nnz * 12 + rows * 20
        c32 = {"bytes_per_launch_c32": int(round(t32)), "algorithmic_bytes_per_launch_c32": a32,
               "ratio_traffic_over_algorithmic_c32": t32 / a32,
               "FETCH_SIZE_KB_mean_c32": f["spmv_c32"], "WRITE_SIZE_KB_mean_c32": w["spmv_c32"]}
    if "spmv_sliced" in f:  # the sliced-pattern copy: algorithmic bytes as tools/pmc_spmv.py prints them
        ts = factor * f["spmv_sliced"] * 1024 + w["spmv_sliced"] * 1024
        a_s = int(sys.argv[5]) if len(sys.argv) > 5 else None
        c32.update({"bytes_per_launch_sliced": int(round(ts)), "algorithmic_bytes_per_launch_sliced": a_s,
                    "ratio_traffic_over_algorithmic_sliced": ts / a_s if a_s else None,
                    "FETCH_SIZE_KB_mean_sliced": f["spmv_sliced"], "WRITE_SIZE_KB_mean_sliced": w["spmv_sliced"]})
    json.dump({
        "workload": "csr_spmv f64, 27-pt 512^3, 1 GPU",
        "tile_placement": placement,
        "FETCH_SIZE_KB_mean": f["spmv"], "WRITE_SIZE_KB_mean": w["spmv"],
        "calibration": {"kernel": f"dot_kernel f64 (reads 2*rows*8 = {dot_true} B)",
                        "FETCH_SIZE_KB_mean": f["dot"], "ratio_true_over_reported": ratio},
        "correction": f"bytes = {factor}*FETCH_SIZE*1024 + WRITE_SIZE*1024 (factor from the calibration)",
        "bytes_per_launch": int(round(traffic)), "algorithmic_bytes_per_launch": algo,
        "ratio_traffic_over_algorithmic": traffic / algo, **c32,
    }, open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()
