"""Reduces a rocprofv3 kernel-trace CSV to the columns the tracked profiles need (dispatch id, short kernel name, start / end ns
relative to the first dispatch, workgroups, workgroup size, LDS bytes, VGPRs).  Usage: trim_trace.py in.csv out.csv"""
import csv
import re
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kind"] == "KERNEL_DISPATCH"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["dispatch", "kernel", "start_ns", "end_ns", "workgroups", "workgroup_size", "lds_bytes", "vgpr", "sgpr"])
for r in rows:
    name = re.sub(r"^void ", "", r["Kernel_Name"])
    name = re.sub(r"\(.*$", "", name)[:80]
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    n = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, wg)
    w.writerow([r["Dispatch_Id"], name, int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, n, wg, r["LDS_Block_Size"], r["VGPR_Count"], r["SGPR_Count"]])
