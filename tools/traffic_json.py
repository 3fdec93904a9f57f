#!/usr/bin/env python3
"""HBM bytes per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) -> a json under profiles/.

    python tools/traffic_json.py <fetch pass dir> <write pass dir> <out json> <agents per launch> <kernel> "<command>"

<kernel>: a part of the kernel's name (k_group_batches, k_group<).  <agents per launch>: all the agents one launch
solves (twenty batches of 65536: 1310720).  gfx950: FETCH_SIZE counts a 128-byte request as 64 bytes, so
read bytes = FETCH_SIZE * 1024 * 2; WRITE_SIZE is in KiB as documented.  The json carries the hash of the sources
the library was built from: bench.py shows the number only next to a library built from the same sources.
"""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sccav_cbf_amd import _build  # noqa: E402


def mean_counter(d, name, part):
    vals, kern = [], None
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if part in row["Kernel_Name"] and row["Counter_Name"] == name:
                vals.append(float(row["Counter_Value"]))
                kern = row["Kernel_Name"]
    # the launches of the ramp and the warm-up may be shorter ones: keep the full-size launches (the largest values)
    top = max(vals)
    vals = [v for v in vals if v > 0.9 * top]
    return sum(vals) / len(vals), len(vals), kern


def main():
    fdir, wdir, out, agents, part, cmd = sys.argv[1:7]
    agents = int(agents)
    fs, nf, kern = mean_counter(fdir, "FETCH_SIZE", part)
    ws, nw, _ = mean_counter(wdir, "WRITE_SIZE", part)
    rd, wr = fs * 1024 * 2, ws * 1024
    alg = agents * (4 * (4 + 2 + 8 * 16 + 2) + 4)
    json.dump({
        "round": 3, "kernel": kern[:160], "source_hash": _build.source_hash(),
        "workload": "%d agents x 16 rows per launch, fp32, prepared obstacle slots in 64-agent tiles" % agents,
        "FETCH_SIZE_KB_mean": fs, "WRITE_SIZE_KB_mean": ws, "launches_averaged": [nf, nw],
        "correction": "read bytes = FETCH_SIZE * 1024 * 2 (FETCH_SIZE counts 128-B requests at 64 B on gfx950); "
                      "write bytes = WRITE_SIZE * 1024",
        "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
        "algorithmic_bytes_per_launch": alg, "ratio": (rd + wr) / alg, "command": cmd}, open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()
