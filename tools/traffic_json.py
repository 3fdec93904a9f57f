#!/usr/bin/env python3
"""HBM bytes per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) -> a json under profiles/.

    python tools/traffic_json.py <fetch pass dir> <write pass dir> <out json> <agents> <round> "<command>"

gfx950: FETCH_SIZE counts a 128-byte request as 64 bytes, so read bytes = FETCH_SIZE * 1024 * 2;
WRITE_SIZE is in KiB as documented.
"""
import csv
import glob
import json
import sys


def mean_counter(d, name):
    vals, kern = [], None
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_group" in row["Kernel_Name"] and row["Counter_Name"] == name:
                vals.append(float(row["Counter_Value"]))
                kern = row["Kernel_Name"].split("(")[1 if row["Kernel_Name"].startswith("void (") else 0]
                kern = row["Kernel_Name"]
    return sum(vals) / len(vals), len(vals), kern


def main():
    fdir, wdir, out, agents, rnd, cmd = sys.argv[1:7]
    agents = int(agents)
    fs, nf, kern = mean_counter(fdir, "FETCH_SIZE")
    ws, nw, _ = mean_counter(wdir, "WRITE_SIZE")
    rd, wr = fs * 1024 * 2, ws * 1024
    name = kern[kern.index("k_group"):kern.index(">") + 1]
    alg = agents * (4 * (4 + 2 + 8 * 16 + 2) + 4)
    json.dump({
        "round": int(rnd), "kernel": name,
        "workload": "%d agents x 16 rows, fp32, prepared obstacle slots in 64-agent tiles" % agents,
        "FETCH_SIZE_KB_mean": fs, "WRITE_SIZE_KB_mean": ws, "launches_averaged": [nf, nw],
        "correction": "read bytes = FETCH_SIZE * 1024 * 2 (FETCH_SIZE counts 128-B requests at 64 B on gfx950); "
                      "write bytes = WRITE_SIZE * 1024",
        "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
        "algorithmic_bytes_per_launch": alg, "ratio": (rd + wr) / alg, "command": cmd}, open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()
