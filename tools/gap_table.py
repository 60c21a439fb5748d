"""Per-call timeline of the B = 1 path from a rocprofv3 --kernel-trace CSV: for the median call (delimited by hardnet_finish_kernel)
every kernel with its start offset, duration and the idle gap in front of it, then per-stage sums (kernel time, gaps, launches).
Usage: python tools/gap_table.py gpurun_out/prof_c2/run_kernel_trace.csv [--graph]   (--graph: the calls replayed as one HIP graph =
the second half of bench.py --config2's calls)"""
import csv
import re
import sys

STAGE = [("pyramid", r"blur2d|aff_copy|decimate"), ("shape", r"shape_"), ("levelsel", r"scale_lafs|level_select"),
         ("detector", r"hessian_nms|level_resolve|resolve_|select_|aff_zero|onepass"),
         ("affnet", r"cnn32_trunk_kernel<0|cnn16_finish_kernel<0|affnet_finish"),
         ("orinet", r"cnn32_trunk_kernel<1|cnn16_finish_kernel<1|orinet_finish|apply_rotation"),
         ("hardnet", r"cnn32_trunk_kernel<2|hardnet_")]


def stage_of(name):
    for s, pat in STAGE:
        if re.search(pat, name):
            return s
    return "other"


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)[:44]


def main():
    path = sys.argv[1]
    rows = list(csv.DictReader(open(path)))
    if rows and "kernel" in rows[0]:            # tools/trim_trace.py format (what profiles/ tracks)
        rows = [(r["kernel"], int(r["start_ns"]), int(r["end_ns"]), int(r["workgroups"])) for r in rows]
    else:                                       # raw rocprofv3 --kernel-trace CSV
        rows = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                 int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])))
                for r in rows if r["Kind"] == "KERNEL_DISPATCH"]
    rows.sort(key=lambda r: r[1])
    calls, cur = [], []
    for r in rows:
        n = r[0]
        if n.startswith("__amd_rocclr") or "at::native" in n:
            continue
        cur.append(r)
        if "hardnet_finish_kernel" in n:
            calls.append(cur)
            cur = []
    if not calls:
        sys.exit("no complete call in the trace")
    half = len(calls) // 2
    sel = calls[half + 2:] if "--graph" in sys.argv else calls[2:half]
    span = lambda c: c[-1][2] - c[0][1]
    sel.sort(key=span)
    c = sel[len(sel) // 2]
    t0 = c[0][1]
    print("# %s calls in the trace; %s; median call: %d kernels, first start -> last end %.1f us"
          % (len(calls), "HIP-graph replays" if "--graph" in sys.argv else "eager calls", len(c), span(c) / 1e3))
    print("| # | kernel | workgroups | start us | duration us | gap before us | stage |\n|---|---|---|---|---|---|---|")
    per = {}
    prev_end = t0
    for i, (n, s, e, wg) in enumerate(c):
        st = stage_of(n)
        gap = max(0, s - prev_end)
        ov = max(0, prev_end - s)
        d = per.setdefault(st, [0.0, 0.0, 0])
        d[0] += (e - max(s, prev_end)) / 1e3 if e > prev_end else 0.0
        d[1] += gap / 1e3
        d[2] += 1
        print("| %d | %s | %d | %.1f | %.1f | %.1f%s | %s |" % (i, short(n), wg, (s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, (" (overlap %.1f)" % (ov / 1e3)) if ov else "", st))
        prev_end = max(prev_end, e)
    print("\n| stage | launches | kernel time us | idle gaps us | share of the call |\n|---|---|---|---|---|")
    tot = span(c) / 1e3
    for st, (k, g, n) in per.items():
        print("| %s | %d | %.1f | %.1f | %.1f %% |" % (st, n, k, g, 100.0 * (k + g) / tot))
    print("| total | %d | %.1f | %.1f | %.1f us |" % (sum(v[2] for v in per.values()), sum(v[0] for v in per.values()), sum(v[1] for v in per.values()), tot))


if __name__ == "__main__":
    main()
