#!/usr/bin/env python
"""Calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on KNOWN byte counts in this library's own access patterns
(MI355X_MICROARCH.md, HBM section: only 16 B/lane streaming reads are calibrated there; "calibrate on a known byte count in
your own access pattern before trusting an absolute").

  run    : launches the known-byte-count kernels (affnet_debug_stream) - wrap in rocprofv3:
             rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/calib_f -o run -- python tools/fetch_calib.py run
             rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/calib_w -o run -- python tools/fetch_calib.py run
  reduce : python tools/fetch_calib.py reduce gpurun_out/calib_f gpurun_out/calib_w > profiles/rNN_fetch_calibration.json
           -> bytes per counter unit for every (pattern, width); tools/pmc_traffic.py applies them per kernel.

Buffers are 1 GiB (4x the 256 MiB Infinity Cache) so that reads really come from HBM; the 'tile' pattern is the 64 x 64 (+apron)
4-byte tile loader of blur2d_kernel / hessian_nms_kernel (apron re-reads are L2 hits: the known byte count is the image size)."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N_BYTES = 1 << 30
CASES = [("read", 4, 0, 0), ("read", 8, 0, 0), ("read", 16, 0, 0), ("write", 4, 1, 0), ("write", 8, 1, 0), ("write", 16, 1, 0),
         ("tile", 4, 2, 2), ("tile", 4, 2, 7)]
KERNEL_OF = {("read", 4): "stream_read_kernel<float>", ("read", 8): "stream_read_kernel<f32x2s>", ("read", 16): "stream_read_kernel<HIP_vector_type<float, 4",
             ("write", 4): "stream_write_kernel<float>", ("write", 8): "stream_write_kernel<f32x2s>", ("write", 16): "stream_write_kernel<HIP_vector_type<float, 4",
             ("tile", 4): "tile_read_kernel"}


def run():
    import torch
    import os as _os
    _os.environ.setdefault("AFFNET_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "affnet_amd", "libaffnet_hip_probes.so"))   # probe kernels live there (include/affnet_hip_probes.h)
    from affnet_amd._lib import lib, ptr
    dev = torch.device("cuda:0")
    src = torch.rand(N_BYTES // 4, device=dev)
    dst = torch.empty(N_BYTES // 4, device=dev)
    torch.cuda.synchronize()
    for name, width, mode, halo in CASES:
        for _ in range(3):
            rc = lib.affnet_debug_stream(ptr(src), ptr(dst), N_BYTES, width, mode, halo, None)
            assert rc == 0, (name, width, rc)
        torch.cuda.synchronize()
    print("fetch_calib: launched", len(CASES), "cases x 3")


def mean_per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(d, "run_counter_collection.csv"))):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def reduce(fetch_dir, write_dir):
    f, w = mean_per_kernel(fetch_dir, "FETCH_SIZE"), mean_per_kernel(write_dir, "WRITE_SIZE")
    out = {"known_bytes_per_launch": N_BYTES, "counter_unit_bytes": 1024,
           "note": "factor = known bytes / (counter value x 1024 B); a factor of 2.0 = the counter tallies a 128-B request as 64 B",
           "read": {}, "write": {}, "tile": {}}
    tile_seen = 0
    for name, width, mode, halo in CASES:
        pat = KERNEL_OF[(name, width)]
        tab = w if name == "write" else f
        vals = [v for k, vs in tab.items() if pat in k for v in vs]
        if not vals:
            continue
        if name == "tile":           # two tile cases share one kernel name: dispatch order = case order, 3 launches each
            vals = vals[3 * tile_seen:3 * tile_seen + 3]
            tile_seen += 1
        raw = sum(vals) / len(vals) * 1024.0
        key = "%dB_per_lane" % width if name != "tile" else "4B_per_lane_halo%d" % halo
        out[name][key] = {"counter_bytes_raw": raw, "factor": N_BYTES / raw if raw else None}
        if name != "write":          # what the OTHER counter saw for this kernel (reads of a write kernel etc.): should be ~0
            other = [v for k, vs in w.items() if pat in k for v in vs]
            out[name][key]["write_counter_bytes_raw"] = (sum(other) / len(other) * 1024.0) if other else None
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "run":
        run()
    elif len(sys.argv) >= 4 and sys.argv[1] == "reduce":
        reduce(sys.argv[2], sys.argv[3])
    else:
        sys.exit(__doc__)
