#!/usr/bin/env python
"""Regenerates the bandwidth / FLOP-rate table of DESIGN.md section 4 from tracked files:
    python tools/roofline_table.py profiles/r02_s2          (prefix of *_kernel_stats.csv, *_traffic.json, *_bench_default.json or, since round 6, *_bench_driver_detail.json)
Per kernel: launches per 32-image call, mean duration (rocprofv3 --kernel-trace --stats), HBM bytes per launch from the
FETCH_SIZE / WRITE_SIZE passes scaled by the calibration measured for the kernel's access width (tools/fetch_calib.py), the
resulting GB/s and its fraction of 8 TB/s; for the CNN kernels the algorithmic FLOP rate against the 157.3 TFLOP/s fp32 MFMA peak.
Algorithmic bytes / FLOPs are SURVEY.md section 8d's figures."""
import csv
import json
import sys

IMGS, H, W, NKP, C = 32, 768, 1024, 2000, 3000
# FLOPs per image.  AffNet: the candidates actually evaluated (device counter, bench line roofline.affnet_patches_evaluated_per_image:
# ~2400 of 3000 with the lazy shape evaluation) over the time of ALL its launches of a call - the second, predicated launch is a
# few us when no image needs it, so a per-launch mean charged with 3000 patches printed 204 % of the peak in round 2.
FLOP = {"cnn32_trunk_kernel<0": C * 19193856.0, "cnn32_trunk_kernel<1": NKP * 19316736.0, "cnn32_trunk_kernel<2": NKP * (78184448.0 - 2.0 * 8192 * 128),
        "hardnet_head_kernel": NKP * 2.0 * 8192 * 128}


def octave_pixels(h, w, border=5):
    tot, min_size = 0, 2 * border + 3
    while True:
        tot += h * w
        nh, nw = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        if nh <= min_size or nw <= min_size:
            return tot
        h, w = nh, nw


def split_terms(name):
    """Arithmetic of a cnn32_trunk_kernel instantiation from its last template argument: 0 = exact fp32 MFMA, 3 = three bf16 terms (round 3 / early
    round 4 builds print it as `true`), 2 = two fp16 terms."""
    n = name.rstrip()
    if "cnn32_trunk_kernel" not in n:
        return 0
    if n.endswith("true>") or n.endswith(", 3>"):
        return 3
    if n.endswith(", 2>"):
        return 2
    return 0


def main(prefix):
    stats = {}
    for r in csv.DictReader(open(prefix + "_kernel_stats.csv")):
        stats[r["Name"].split("(")[0]] = (int(r["Calls"]), float(r["AverageNs"]))
    traffic = json.load(open(prefix + "_traffic.json"))["kernels"]
    import os
    # rounds 1 - 5: the full record was the bench line itself (*_bench_default.json); since round 6 the line is compact and the record a side file
    bfile = prefix + "_bench_default.json" if os.path.exists(prefix + "_bench_default.json") else prefix + "_bench_driver_detail.json"
    bench = json.loads(open(bfile).readline())
    aff_eval = bench.get("roofline", {}).get("affnet_patches_evaluated_per_image")
    if aff_eval:
        FLOP["cnn32_trunk_kernel<0"] = aff_eval * 19193856.0
    # one HardNet trunk launch per 32-image call; a run that also took the arith fp32_split3 steps (bench.py without --no-split3)
    # has them under their own template instantiations <2, 8, false, 3> / <2, 8, false, 2> - every other kernel is shared by all kinds of step
    calls_per_batch = sum(v[0] for k, v in stats.items() if k.startswith("void cnn32_trunk_kernel<2, 8, false"))
    P0, P = H * W, octave_pixels(H, W)
    alg = {"blur2d_kernel": (P0 + 9 * P) * 4.0 * IMGS, "hessian_nms_kernel": 5 * P * 4.0 * IMGS}
    print("| kernel | launches / 32-image call | mean us | HBM bytes / launch (PMC, calibrated) | PMC GB/s | % of 8 TB/s | algorithmic rate |")
    print("|---|---|---|---|---|---|---|")
    groups = {}
    for name, (calls, avg_ns) in sorted(stats.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
        if "rocclr" in name or "at::" in name or "elementwise" in name:
            continue
        t = traffic.get(name)
        per_call = calls / float(calls_per_batch)
        hbm = t["hbm_bytes"] if t else None
        gbs = hbm / avg_ns if hbm else None                                  # bytes / ns = GB/s
        algs = ""
        split = split_terms(name)
        if "cnn32_trunk_kernel" in name:                                     # exact and split instantiations: launches of their own steps only
            own = sum(v[0] for k, v in stats.items() if k.startswith("void cnn32_trunk_kernel<2, 8, false") and split_terms(k) == split)
            per_call = calls / float(max(1, own))
        for key, fl in FLOP.items():
            if key in name and split:
                tf = fl * IMGS / (per_call * avg_ns * 1e-9) / 1e12
                prod = 6 if split == 3 else 3
                algs = ("arith %s: %.1f TFLOP/s fp32-equivalent; executed %s products (%d per fp32 product) %.0f TFLOP/s = %.1f %% of "
                        "the 2517 bf16 / fp16 MFMA peak" % ("fp32_split3" if split == 3 else "fp32_split2h", tf, "bf16" if split == 3 else "fp16", prod, prod * tf,
                                                          100 * prod * tf / 2516.8))
            elif key in name:
                tf = fl * IMGS / (per_call * avg_ns * 1e-9) / 1e12            # all launches of the kernel in one 32-image call
                algs = "%.1f TFLOP/s = %.1f %% of 157.3" % (tf, 100 * tf / 157.3)
                if per_call > 1.01:
                    algs += " (over its %.0f launches per call; %.0f patches / image evaluated)" % (per_call, FLOP[key] / 19193856.0)
        for key in alg:
            if key in name:
                g = groups.setdefault(key, [0.0, 0.0])
                g[0] += per_call * avg_ns
                g[1] += per_call * (hbm or 0.0)
        print("| `%s` | %.2f | %.1f | %s | %s | %s | %s |" % (name.replace("void ", "")[:60], per_call, avg_ns / 1e3, "%.3g" % hbm if hbm else "-",
                                                           "%.0f" % gbs if gbs else "-", "%.1f" % (100 * gbs / 8000.0) if gbs else "-", algs))
    print()
    for key, (ns, hbm) in groups.items():
        print("%s (all launches of one 32-image call): %.1f us, algorithmic %.1f MB/image -> %.0f GB/s = %.1f %% of 8 TB/s; PMC (calibrated) %.1f MB/image -> %.0f GB/s"
              % (key, ns / 1e3, alg[key] / IMGS / 1e6, alg[key] / ns, 100 * alg[key] / ns / 8000.0, hbm / IMGS / 1e6, hbm / ns))
    r = bench["roofline"]
    print("bench line: %.0f kp/s, HardNet trunk %.1f TFLOP/s = %.1f %% (HIP events), all CNN kernels %.1f TFLOP/s"
          % (bench["value"], r["achieved"], 100 * r["frac"], r["all_cnn_tflops"]))
    for s in bench.get("secondary_rooflines", []):
        print("secondary: %s: %.1f %s = %.1f %% of the %s peak %.4g (%.4f ms/image)" % (s["kernel"][:70], s["achieved"], s["unit"], 100 * s["frac"], s["bound"], s["peak"],
                                                                                   s["ms_per_image"]))


if __name__ == "__main__":
    main(sys.argv[1])
