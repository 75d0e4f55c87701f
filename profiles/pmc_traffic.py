#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, then
WRITE_SIZE; --output-format csv): average per dispatch of every snapmi kernel.

usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <workload text>

Counters are KB per dispatch.  FETCH_SIZE on gfx950 under-reports reads of
>= 16 B/lane by 2x (MI355X_MICROARCH.md, HBM section; checked here on
k_compact, a pure 16-byte copy: FETCH_SIZE = 0.50 x WRITE_SIZE), so the
table carries both the raw sum and the sum with FETCH_SIZE doubled; the
doubled figure is an upper bound for kernels with narrower reads."""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    seen = set()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "snapmi::" not in k or r["Counter_Name"] != counter:
            continue
        k = k.split("snapmi::")[1].split("(")[0]
        tot[k] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            cnt[k] += 1
    return {k: tot[k] / cnt[k] for k in tot}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"workload": sys.argv[3],
           "method": __doc__.split("\n\n", 2)[2].replace("\n", " "),
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        out["kernels"][k] = {
            "FETCH_SIZE_KB": round(f), "WRITE_SIZE_KB": round(w),
            "traffic_bytes_raw": round((f + w) * 1024),
            "traffic_bytes_fetch_x2": round((2 * f + w) * 1024)}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
