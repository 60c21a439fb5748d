"""For every kernel name matching the regex: the counters of its LONGEST dispatch in each rocprofv3 --pmc pass (the octave-0 launch
of the scale-space kernels).  Usage: pmc_biggest.py dir1 [dir2 ...] 'regex'"""
import collections
import csv
import re
import sys

dirs, pat = sys.argv[1:-1], re.compile(sys.argv[-1])
for d in dirs:
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(d + "/run_counter_collection.csv")):
        name = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("void ", ""))[:40]
        if not pat.search(name):
            continue
        key = (name, r["Dispatch_Id"])
        per[key]["us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        per[key]["wgs"] = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
        per[key][r["Counter_Name"]] = per[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    best = {}
    for (name, did), v in per.items():
        if name not in best or v["us"] > best[name]["us"]:
            best[name] = v
    print("==", d)
    for name, v in sorted(best.items(), key=lambda kv: -kv[1]["us"]):
        print("%-40s %9.1f us %7d wgs  " % (name, v["us"], v["wgs"]) + "  ".join("%s=%.4g" % (k, x) for k, x in sorted(v.items()) if k not in ("us", "wgs")))
