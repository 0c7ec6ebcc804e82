#!/usr/bin/env python3
"""Known bytes of tools/calibration/fetch_calibration against rocprofv3's counters:
   python tools/calibration/summarize.py <dir with run.log and pass*/ counter CSVs>  >  profiles/r3_fetch_size_calibration.txt"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]
requested, best_ms = {}, {}
for line in open(os.path.join(root, "run.log")):
    m = re.match(r"(k_cal_\w+)\s+requested\s+(\d+) bytes\s+([\d.]+) ms", line)
    if m:
        requested[m.group(1)] = float(m.group(2))
        best_ms[m.group(1)] = min(best_ms.get(m.group(1), 1e30), float(m.group(3)))
    elif line.startswith("table"):
        print("# " + line.strip())
tot, calls = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
for path in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"] or 0)
        calls[k][row["Counter_Name"]] += 1
print("# per dispatch; FETCH_SIZE is reported in KiB by rocprofv3 (x 1024 here); best of 3 timed dispatches without the profiler")
print("%-16s %15s %15s %9s %14s %14s %10s %9s" % ("kernel", "requested B", "FETCH_SIZE B", "req/FETCH", "EA0_RDREQ", "EA0_RDREQ_32B", "B/RDREQ", "GB/s req"))
for k in ("k_cal_stream", "k_cal_gather64", "k_cal_gather48", "k_cal_gather32", "k_cal_gather16"):
    c = {n: tot[k][n] / max(calls[k][n], 1) for n in tot[k]}
    fetch = c.get("FETCH_SIZE", 0.0) * 1024.0
    rd, rd32 = c.get("TCC_EA0_RDREQ_sum", 0.0), c.get("TCC_EA0_RDREQ_32B_sum", 0.0)
    req = requested.get(k, 0.0)
    print("%-16s %15.0f %15.0f %9.3f %14.0f %14.0f %10.1f %9.1f" % (k, req, fetch, req / fetch if fetch else 0.0, rd, rd32, req / rd if rd else 0.0,
                                                                    req / (best_ms.get(k, 1e30) * 1e-3) / 1e9))
