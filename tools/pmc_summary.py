"""Average the counters of a rocprofv3 --pmc csv per kernel name."""
import csv
import glob
import sys
from collections import defaultdict

files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = defaultdict(lambda: defaultdict(list))
for f in files:
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        if "k_group" not in name and "k_wave" not in name and "k_" not in name:
            continue
        acc[name[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, d in acc.items():
    print(name)
    for c, v in sorted(d.items()):
        print("   %-28s mean %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
