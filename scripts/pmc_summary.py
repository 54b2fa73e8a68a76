#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc CSV output (one directory per pass) into per-kernel totals.
Usage: python scripts/pmc_summary.py <dir-with-pass_*> [--json out.json] > summary.txt
The JSON form ({"kernels": {name: {counter: per-dispatch value}}}) is what bench.py reads for roofline.traffic."""
import json
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:40]


def main(root, json_path=None):
    table = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(int))
    for path in sorted(glob.glob(os.path.join(root, "pass_*", "**", "*counter_collection.csv"), recursive=True)):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                c = row["Counter_Name"]
                table[k][c] += float(row["Counter_Value"])
                calls[k][c] += 1
    for k in sorted(table, key=lambda k: -table[k].get("SQ_WAVE_CYCLES", 0)):
        print(k)
        for c in sorted(table[k]):
            n = calls[k][c]
            print("    %-36s total %18.0f  per-dispatch %16.1f  (%d dispatches)" % (c, table[k][c], table[k][c] / n, n))
    if json_path:
        with open(json_path, "w") as f:
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            from bench import source_sha                  # the kernels these counters were taken on
            json.dump({"source": "rocprofv3 --pmc, one run per counter group (scripts/pmc_passes.sh); values are per dispatch",
                       "source_sha": source_sha(),
                       "kernels": {k: {c: table[k][c] / calls[k][c] for c in sorted(table[k])} for k in sorted(table)},
                       "dispatches": {k: max(calls[k].values()) for k in sorted(table)}}, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[3] if len(sys.argv) > 3 and sys.argv[2] == "--json" else None)
