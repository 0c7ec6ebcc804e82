#!/usr/bin/env python3
"""Turn one tools/measure_solo.sh directory into a per-kernel JSON record:
durations from the --stats pass, HBM-side bytes from the FETCH_SIZE / WRITE_SIZE passes (both reported in KB).  FETCH_SIZE books
64 B per fabric request: right for scattered 16-64 B fetches, half the bytes for the 128 B requests of wide coalesced reads
(profiles/r3_fetch_size_calibration.txt), so  hbm bytes = FETCH_SIZE + WRITE_SIZE + (streamed read bytes) / 2  with the streamed
bytes a kernel is known to read in slot order (STREAM_BYTES_PER_SLOT x the slots of a launch; everything for the reductions).  VALU lane
utilisation = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64), VALU issue = SQ_INSTS_VALU x 4 / (GRBM_GUI_ACTIVE x 128),
share of the waves' cycles spent waiting = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root, out_path = sys.argv[1], sys.argv[2]
args = sys.argv[3] if len(sys.argv) > 3 else ""


def short(name):
    n = name.replace("void ", "").replace("ptrk::", "")
    return n.split("(")[0].replace(" ", "")


dur = {}
for path in glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = short(row["Name"])
            dur[k] = {"calls": int(row["Calls"]), "avg_ms": float(row["AverageNs"]) / 1e6, "total_ms": float(row["TotalDurationNs"]) / 1e6}
tot = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for path in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = short(row.get("Kernel_Name", ""))
            c = row.get("Counter_Name")
            tot[k][c] += float(row.get("Counter_Value", 0) or 0)
            calls[k][c] += 1
HBM_PEAK = 8.0e12
# coalesced reads per path slot and launch (16 B per lane, lanes on consecutive slots): k_extend walks ray0 + ray1, k_shade ray1, ray0, hit,
# thr and accum; k_connect's records and every scene fetch are gathers.  The reductions read nothing but streams.
STREAM_BYTES_PER_SLOT = {"k_extend": 32.0, "k_shade<": 72.0}
ALL_STREAM = ("k_resolve", "k_flush", "k_generate", "k_tail_collect")


def pool_slots(bench_args):
    """Slots of a --solo launch (the whole pool as one group), as csrc/host/hip_backend.cpp sizes it."""
    import re
    def opt(name, default):
        m = re.search(r"--%s[ =](\d+)" % name, bench_args)
        return int(m.group(1)) if m else default
    items = opt("width", 1920) * opt("height", 1080) * opt("spp", 256)
    return min(32 << 20, max(1 << 20, items // 2), items) & ~255


SLOTS = pool_slots(args)
rec = {"command": "rocprofv3 --kernel-trace [--stats | --pmc <group>] -- python3 bench.py --solo --steps 1 --warmup 1 --no-cpu-baseline " + args,
       "pool_slots": SLOTS,
       "hbm_bytes": "FETCH_SIZE + WRITE_SIZE + streamed_read_bytes / 2 (FETCH_SIZE books the 128 B requests of coalesced reads at 64 B and "
                    "every scattered fetch at the 64 B it moves: profiles/r3_fetch_size_calibration.txt)",
       "note": "pool as ONE group: kernels never overlap; per-dispatch means over every dispatch of the process (warm-up, timed step, solo "
               "re-render); <true,..> instantiations are the counting build of bench.py's extra render",
       "kernels": {}}
for k in sorted(set(dur) | set(tot)):
    if not k.startswith("k_"):
        continue
    r = {}
    if k in dur:
        r.update({"dispatches": dur[k]["calls"], "avg_ms": round(dur[k]["avg_ms"], 5), "total_ms": round(dur[k]["total_ms"], 3)})
    c = tot.get(k, {})
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c and k in dur and dur[k]["avg_ms"] > 0:
        fetch = c["FETCH_SIZE"] / max(calls[k]["FETCH_SIZE"], 1) * 1024.0
        write = c["WRITE_SIZE"] / max(calls[k]["WRITE_SIZE"], 1) * 1024.0
        if k.startswith(ALL_STREAM):
            streamed = 2.0 * fetch
        else:
            per_slot = next((v for key, v in STREAM_BYTES_PER_SLOT.items() if k.startswith(key)), 0.0)
            streamed = min(per_slot * SLOTS, 2.0 * fetch)   # (an end-of-frame launch over a drained pool reads less)
        hbm = fetch + 0.5 * streamed + write
        r.update({"fetch_size_bytes_per_dispatch": round(fetch), "write_size_bytes_per_dispatch": round(write),
                  "streamed_read_bytes_per_dispatch": round(streamed),
                  "hbm_bytes_per_dispatch": round(hbm), "hbm_gbs": round(hbm / (dur[k]["avg_ms"] * 1e-3) / 1e9, 1),
                  "hbm_frac_of_8TBs": round(hbm / (dur[k]["avg_ms"] * 1e-3) / HBM_PEAK, 4),
                  "hbm_bytes_if_all_reads_were_streams": round(2.0 * fetch + write)})
    if c.get("SQ_ACTIVE_INST_VALU", 0) > 0:
        r["valu_lane_utilisation"] = round(c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0), 4)
        r["valu_insts_per_dispatch"] = round(c["SQ_INSTS_VALU"] / max(calls[k]["SQ_INSTS_VALU"], 1))
        r["salu_insts_per_dispatch"] = round(c.get("SQ_INSTS_SALU", 0) / max(calls[k].get("SQ_INSTS_SALU", 1), 1))
        if c.get("GRBM_GUI_ACTIVE", 0) > 0:
            r["valu_issue"] = round(c["SQ_INSTS_VALU"] / max(calls[k]["SQ_INSTS_VALU"], 1) * 4.0 /
                                    (c["GRBM_GUI_ACTIVE"] / max(calls[k]["GRBM_GUI_ACTIVE"], 1) * 128.0), 4)
    if c.get("SQ_WAVE_CYCLES", 0) > 0:
        # how the resident waves spend their cycles: waiting on any outstanding instruction (memory above all) against the
        # share in which a vector-memory or LDS instruction is in flight for them
        r["wave_cycles_waiting"] = round(c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"], 4)
        r["wave_cycles_vmem_active"] = round(c.get("SQ_ACTIVE_INST_VMEM", 0) / c["SQ_WAVE_CYCLES"], 4)
        r["wave_cycles_lds_active"] = round(c.get("SQ_ACTIVE_INST_LDS", 0) / c["SQ_WAVE_CYCLES"], 4)
    rec["kernels"][k] = r
with open(out_path, "w") as f:
    json.dump(rec, f, indent=1)
