#!/usr/bin/env python3
"""HBM-side traffic of the COO product kernels from two rocprofv3 --pmc passes over
tools/coo_binned_exp.py (FETCH_SIZE and WRITE_SIZE, each its own run):

    python tools/pmc_coo_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <exp.log> <out.json>

FETCH_SIZE is reported in KiB and tallies a 128-byte read request as 64 bytes on gfx950; the dot
product in the same run (a known number of bytes read, nothing else) calibrates that factor."""
import csv
import json
import re
import sys

KINDS = (("binned_multiply", "pass1"), ("binned_reduce", "pass2"), ("coo_spmv_scan", "scan"), ("dot_kernel", "dot"))


def means(path, counter):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for needle, key in KINDS:
            if needle in r["Kernel_Name"]:
                acc.setdefault(key, []).append(float(r["Counter_Value"]))
    return {k: sum(v[-3:]) / len(v[-3:]) for k, v in acc.items()}  # the last three launches of each kind


def main():
    fetch, write, log, out = sys.argv[1:5]
    text = open(log).read()
    known = {k: int(v) for k, v in re.findall(r"(\w+)=(\d+)", text)}
    f, w = means(fetch, "FETCH_SIZE"), means(write, "WRITE_SIZE")
    ratio = known["dot_bytes"] / (f["dot"] * 1024)
    if 1.9 < ratio < 2.1:
        factor = 2
    elif 0.95 < ratio < 1.05:
        factor = 1
    else:
        raise SystemExit(f"FETCH_SIZE calibration ratio {ratio:.3f} is neither 1 nor 2")
    res = {"workload": "COO product, random 32-per-row matrix of 15 625 000 rows, fp64, 1 GPU",
           "calibration": {"kernel": f"dot_kernel f64 ({known['dot_bytes']} B read)", "FETCH_SIZE_KB_mean": f["dot"],
                           "ratio_true_over_reported": ratio},
           "correction": f"bytes = {factor}*FETCH_SIZE*1024 + WRITE_SIZE*1024 (factor from the calibration)"}
    algo = {"pass1": known.get("pass1"), "pass2": known.get("pass2"), "scan": known.get("c32_layout")}
    for key in ("pass1", "pass2", "scan"):
        if key in f:
            t = factor * f[key] * 1024 + w.get(key, 0.0) * 1024
            res[key] = {"FETCH_SIZE_KB_mean": f[key], "WRITE_SIZE_KB_mean": w.get(key, 0.0),
                        "bytes_per_launch": int(round(t)), "algorithmic_bytes_per_launch": algo[key],
                        "ratio_traffic_over_algorithmic": t / algo[key] if algo[key] else None}
    if "pass1" in res and "pass2" in res:
        both = res["pass1"]["bytes_per_launch"] + res["pass2"]["bytes_per_launch"]
        res["binned_product"] = {"bytes_per_product": both, "own_algorithmic_bytes": algo["pass1"] + algo["pass2"],
                                 "bytes_of_the_32bit_layout": known.get("c32_layout")}
    json.dump(res, open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()
