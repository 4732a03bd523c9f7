#!/usr/bin/env python3
"""HBM-side traffic per launch of the products tools/pmc_products.py ran, from its two rocprofv3 --pmc passes:

    python3 tools/pmc_products_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <workload log> <out.json>

FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE tallies a 128-byte request as 64 bytes on gfx950, which the dot
product of known size in the same run confirms (factor 2) or refutes (then this stops)."""
import csv
import json
import sys


def last_three(path, counter, prefix):
    values = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
              if r["Counter_Name"] == counter and prefix in r["Kernel_Name"]]
    if not values:
        raise SystemExit(f"no launch of {prefix!r} in {path}")
    return sum(values[-3:]) / len(values[-3:]), len(values)


def main():
    fetch, write, log, out = sys.argv[1:5]
    products, calib = [], None
    for line in open(log):
        parts = [p.strip() for p in line.split("|")]
        if line.startswith("product "):
            products.append((parts[0][len("product "):], parts[1], int(parts[2]), parts[3]))
        elif line.startswith("calibration "):
            calib = (parts[0][len("calibration "):], int(parts[1]))
    f_dot, _ = last_three(fetch, "FETCH_SIZE", calib[0])
    ratio = calib[1] / (f_dot * 1024)
    if not 1.9 < ratio < 2.1:
        raise SystemExit(f"FETCH_SIZE calibration ratio {ratio:.3f} is not 2")
    result = {"calibration": {"kernel": calib[0], "bytes": calib[1], "ratio_true_over_reported": ratio},
              "correction": "bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024"}
    for label, prefix, algo, placement in products:
        f, launches = last_three(fetch, "FETCH_SIZE", prefix)
        w, _ = last_three(write, "WRITE_SIZE", prefix)
        traffic = 2 * f * 1024 + w * 1024
        result[label] = {"kernel": prefix + " ...>", "launches_seen": launches, "placement": placement,
                         "FETCH_SIZE_KB_mean": f, "WRITE_SIZE_KB_mean": w, "bytes_per_launch": int(round(traffic)),
                         "algorithmic_bytes_per_launch": algo, "ratio_traffic_over_algorithmic": traffic / algo}
    json.dump(result, open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()
