#!/usr/bin/env python
"""Summarise rocprofv3 counter CSVs into per-kernel HBM traffic per launch.

    python tools/pmc_summary.py --fetch <dir with FETCH_SIZE pass> --write <dir with WRITE_SIZE pass> \
        --workload o1280_r7 --out profiles/r01_hbm_traffic.json

FETCH_SIZE and WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE tallies 128-byte read
requests at 64 bytes, so the read side is doubled; WRITE_SIZE is taken as is.  Only dispatches
of the library's own kernels (names starting with ``ag_``) are kept; template arguments are
stripped so that the names match the ones the library's event profiler reports.
"""
from __future__ import annotations

import argparse
import csv
import json
import re
from collections import defaultdict
from pathlib import Path


def kernel_key(name: str) -> str:
    m = re.search(r"(ag_[A-Za-z0-9_]+)", name)
    return m.group(1) if m else ""


def collect(directory: str, counter: str) -> dict[str, tuple[int, float]]:
    acc: dict[str, list[float]] = defaultdict(list)
    for path in Path(directory).rglob("*counter_collection.csv"):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                key = kernel_key(row.get("Kernel_Name", ""))
                if key:
                    acc[key].append(float(row["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in acc.items()}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--workload", required=True)
    ap.add_argument("--out", required=True)
    args = ap.parse_args()

    fetch = collect(args.fetch, "FETCH_SIZE")
    write = collect(args.write, "WRITE_SIZE")
    kernels = {}
    for name in sorted(set(fetch) | set(write)):
        nf, f_kib = fetch.get(name, (0, 0.0))
        nw, w_kib = write.get(name, (0, 0.0))
        read_bytes = 2.0 * f_kib * 1024.0
        write_bytes = w_kib * 1024.0
        kernels[name] = {
            "dispatches": max(nf, nw),
            "fetch_size_kib_avg": round(f_kib, 2),
            "write_size_kib_avg": round(w_kib, 2),
            "read_bytes_per_launch": round(read_bytes),
            "write_bytes_per_launch": round(write_bytes),
            "hbm_bytes_per_launch": round(read_bytes + write_bytes),
        }
    out_path = Path(args.out)
    doc = json.loads(out_path.read_text()) if out_path.exists() else {}
    doc[args.workload] = {
        "correction": "read bytes = 2 x FETCH_SIZE x 1024 (gfx950 counts 128-B read requests as 64 B); "
                      "write bytes = WRITE_SIZE x 1024; separate --pmc passes; averages over all dispatches",
        "kernels": kernels,
    }
    out_path.write_text(json.dumps(doc, indent=1, sort_keys=True) + "\n")
    for name, k in kernels.items():
        print(f"{name:36s} n={k['dispatches']:4d} read={k['read_bytes_per_launch']/1e6:9.2f} MB "
              f"write={k['write_bytes_per_launch']/1e6:9.2f} MB")


if __name__ == "__main__":
    main()
