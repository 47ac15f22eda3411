"""Summarise rocprofv3 --pmc counter CSVs: per kernel, mean counter value per launch.
usage: python tools/pmc_summary.py <dir> [<dir> ...]"""
import csv
import glob
import re
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                m = re.search(r"k_[a-z0-9_]+", row["Kernel_Name"])
                if m:
                    acc[m.group(0)][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print(f"  {c:32s} {sum(v) / len(v):16.1f}   (n={len(v)})")
