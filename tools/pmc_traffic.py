#!/usr/bin/env python
"""HBM traffic per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate passes, --kernel-trace only).
Usage: pmc_traffic.py <fetch_dir> <write_dir> <images_per_launch> [calibration.json] > profiles/rNN_traffic.json

FETCH_SIZE / WRITE_SIZE are reported in units of 1024 B (rocprofv3 derived counters).  The bytes behind one unit depend on the
access width (MI355X_MICROARCH.md, HBM section: a 16 B/lane streaming read is tallied at half its size, everything else is
uncalibrated), so each kernel's counters are scaled with the factor MEASURED for its own load / store width on a known byte count
(tools/fetch_calib.py -> profiles/rNN_fetch_calibration.json).  Without a calibration file the raw counters are reported and
`hbm_bytes` falls back to the guide's 2 x FETCH_SIZE + WRITE_SIZE."""
import collections
import csv
import json
import sys

# dominant global load / store width (bytes per lane) of each kernel's HBM-facing accesses
WIDTHS = {
    "blur2d_kernel": ("tile", "4B_per_lane_halo7", 16),        # 4-byte tile loader (apron 4..7 px), float4 row stores
    "hessian_nms_kernel": ("tile", "4B_per_lane_halo2", 4),    # 4-byte tile loader (2-px apron), 24-byte RawMax records
    "cnn32_trunk_kernel": ("read", "4B_per_lane", 16),         # sampler: 4-byte gathers (weights: 16-byte loads, L2 hits); conv5 tile: float4 stores
    "hardnet_head_kernel": ("read", "16B_per_lane", 4),        # conv5 slabs + weights as 16-byte buffer loads; partials as 4-byte stores
    "hardnet_finish_kernel": ("read", "4B_per_lane", 4),
    "grid_sample_kernel": ("read", "4B_per_lane", 4),
    "affnet_finish_kernel": ("read", "16B_per_lane", 4),
    "orinet_finish_kernel": ("read", "4B_per_lane", 4),
    "blur2d_pair_kernel": ("tile", "4B_per_lane_halo7", 16),
}


def mean_per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(d + "/run_counter_collection.csv")):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main(fetch_dir, write_dir, imgs, calib_path=None):
    f, w = mean_per_kernel(fetch_dir, "FETCH_SIZE"), mean_per_kernel(write_dir, "WRITE_SIZE")
    cal = json.load(open(calib_path)) if calib_path else None
    out = {"images_per_launch": int(imgs), "unit": "bytes per launch", "kernels": {},
           "correction": ("FETCH_SIZE / WRITE_SIZE scaled per kernel by the factor measured for its access width (%s)" % calib_path) if cal
                         else "2 x FETCH_SIZE + WRITE_SIZE (guide's factor for 16 B/lane reads applied to every kernel: uncalibrated)"}
    for k in sorted(f, key=lambda k: -f[k]):
        if "rocclr" in k or "at::" in k:
            continue
        fb, wb = f[k] * 1024.0, w.get(k, 0.0) * 1024.0
        rec = {"fetch_bytes_raw": fb, "write_bytes_raw": wb}
        rf, wf, how = 2.0, 1.0, "uncalibrated (2 x FETCH)"
        if cal:
            for pat, (table, key, wwidth) in WIDTHS.items():
                if pat in k:
                    rf = (cal.get(table, {}).get(key) or {}).get("factor") or rf
                    wf = (cal["write"].get("%dB_per_lane" % wwidth) or {}).get("factor") or wf
                    how = "read x%.3f (%s/%s), write x%.3f (%d B/lane)" % (rf, table, key, wf, wwidth)
                    break
        rec.update(read_factor=rf, write_factor=wf, calibration=how, fetch_bytes=rf * fb, write_bytes=wf * wb, hbm_bytes=rf * fb + wf * wb)
        out["kernels"][k] = rec
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:5])
