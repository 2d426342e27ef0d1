#!/usr/bin/env python
"""Average rocprofv3 --pmc counters per kernel name: python tools/pmc_kernels.py <dir> [name-filter]"""
import csv
import re
import sys
from collections import defaultdict
from pathlib import Path

acc = defaultdict(lambda: defaultdict(list))
for path in Path(sys.argv[1]).rglob("*counter_collection.csv"):
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            m = re.search(r"(ag_[A-Za-z0-9_]+(<\d+>)?|rocprim[A-Za-z0-9_:]*)", row["Kernel_Name"])
            if not m:
                continue
            acc[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for name in sorted(acc):
    if flt and flt not in name:
        continue
    parts = [f"{c}={sum(v) / len(v):.4g}" for c, v in sorted(acc[name].items())]
    n = len(next(iter(acc[name].values())))
    print(f"{name} (n={n}): " + " ".join(parts))
