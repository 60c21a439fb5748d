#!/usr/bin/env python
"""Summarises rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel.
Usage: pmc_summary.py dir1 [dir2 ...]"""
import collections
import csv
import sys


def main(dirs):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for d in dirs:
        seen = set()
        for r in csv.DictReader(open(d + "/run_counter_collection.csv")):
            k = r["Kernel_Name"].split("(")[0][:48]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            key = (d, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k in sorted(acc, key=lambda k: -sum(dur[k])):
        if "rocclr" in k or "at::" in k:
            continue
        print("%-48s calls/pass %4d  avg %9.1f us" % (k, len(dur[k]) // len(dirs), sum(dur[k]) / len(dur[k])))
        print("    " + "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items())))
        m = {c: sum(v) / len(v) for c, v in acc[k].items()}
        if m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0 and m.get("GRBM_GUI_ACTIVE", 0) > 0:
            # MFMA busy cycles are summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs
            print("    -> matrix pipe busy %.1f %% of SIMD cycles" % (100.0 * (m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (m["GRBM_GUI_ACTIVE"] / 8.0)))


if __name__ == "__main__":
    main(sys.argv[1:])
