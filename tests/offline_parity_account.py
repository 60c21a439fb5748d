#!/usr/bin/env python
"""Re-runs the parity accounting (oracle/fp64_referee.py) on rows the HIP path produced earlier - on ANY host, no GPU:

    AFFNET_DUMP_ROWS=gpurun_out/rows python -m pytest tests -m gpu ...        (on the GPU box: tests/test_gpu_parity.py::_row_stats writes the rows)
    python tests/offline_parity_account.py gpurun_out/rows [out.json]         (here)

The live oracle of a GPU run is the reference on the GPU box's host CPU; this script compares the same GPU rows with the reference on
THIS host (the authoring container = the host of tests/golden): the reference's own host-to-host variation (MKL sgemm order of the
small-map centroid convolutions, affine_grid with / without fma: tools/probes/cpu_conv_order.py) shows as the difference between the two
accounts.  Test infrastructure (imports the oracle)."""
import glob
import json
import os
import re
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "oracle"), HERE):
    sys.path.insert(0, p)
import affnet_oracle as orc  # noqa: E402
import fp64_referee as rf  # noqa: E402
from conftest import load_gray  # noqa: E402

GOLD = os.path.join(HERE, "golden")
CASES = [  # (file-name pattern, image factory, N)
    (r"^configs_1___graf_img1", lambda m: load_gray(os.path.join(GOLD, "graf_img1.png")), 2000),
    (r"^graf_img1_800x640__500", lambda m: load_gray(os.path.join(GOLD, "graf_img1.png")), 500),
    (r"^synthetic_320x240_seed_1", lambda m: orc.synthetic_image(240, 320, 1), 300),
    (r"^configs_2__metric_configuration__image_(\d+)_", lambda m: orc.synthetic_image(768, 1024, int(m.group(1))), 2000),
    (r"^configs_4___3840x2160_seed_0", lambda m: orc.synthetic_image(2160, 3840, 0), 8000),
    (r"^odd_sized_input_(hesaffnet_cat|hesaffnet_fox1)", lambda m: load_gray(os.path.join(GOLD, m.group(1) + ".png")), 2000),
    (r"^odd_sized_input_synth_481x641_s5", lambda m: orc.synthetic_image(481, 641, 5), 2000),
    # tests/dump_rows_gpu.py: the wide sweep
    (r"^sweep_synth_(\d+)x(\d+)_s(\d+)_n(\d+)", lambda m: orc.synthetic_image(int(m.group(1)), int(m.group(2)), int(m.group(3))), None),
    (r"^sweep_graf_img6_n(\d+)", lambda m: load_gray(os.path.join(GOLD, "graf_img6.png")), None),
    (r"^sweep_hesaffnet_cat_n(\d+)", lambda m: load_gray(os.path.join(GOLD, "hesaffnet_cat.png")), None),
]


def main(argv):
    rows_dir = argv[0]
    sd = {k: torch.load(os.path.join(ROOT, "pretrained", k + ".pth"), map_location="cpu", weights_only=False)["state_dict"] for k in ("AffNet", "OriNet")}
    runs, out = {}, {}
    for f in sorted(glob.glob(os.path.join(rows_dir, "*.npz"))):
        name = os.path.basename(f)[:-4]
        for pat, img, n in CASES:
            m = re.match(pat, name)
            if not m:
                continue
            if n is None:                                # N is the last group of the file name
                n = int(m.groups()[-1])
            key = (pat, m.groups(), n)
            if key not in runs:
                x = img(m)
                ex = orc.OracleExtractor(mrSize=5.192, num_features=n, border=5, num_Baum_iters=1, affnet_sd=sd["AffNet"], orinet_sd=sd["OriNet"])
                ex(x, do_ori=True)
                runs[key] = rf.Referee(ex, x.size(3), x.size(2))
            g = np.load(f)
            acc = rf.parity_account(runs[key], g["ids"], g["LAFs"], n)
            out[name] = acc
            print("%-100s matched %d/%d  unmatched %d (unexplained %d)  rows >= 1e-3 px: %d (worse than cpu vs fp64 %d, unexplained %d)  max %.3g px"
                  % (name[:100], acc["matched"], acc["keypoints_cpu"], acc["unmatched_keys"], acc["unmatched_unexplained"], acc["rows_outside_1e-3"],
                     acc["rows_worse_than_cpu_vs_fp64"], acc["rows_outside_1e-3_beyond_referee"], acc["laf_max_px_gpu_vs_cpu"]))
            break
    if len(argv) > 1:
        json.dump({"what": "oracle/fp64_referee.parity_account of dumped GPU rows against the reference on this host (%s)" % os.uname().nodename,
                   "cases": out}, open(argv[1], "w"), indent=1, sort_keys=True)
    bad = [k for k, a in out.items() if a["unmatched_unexplained"] or a["rows_outside_1e-3_beyond_referee"] > a["beyond_budget"] or a["rows_outside_1e-2"] or a["rows_outside_5e-3_unexplained"]]
    tot = sum(a["matched"] for a in out.values())
    print("matched rows %d, rows >= 1e-3 px %d, worst %.3g px" % (tot, sum(a["rows_outside_1e-3"] for a in out.values()), max(a["laf_max_px_gpu_vs_cpu"] for a in out.values())))
    print("cases: %d, with unexplained keys / rows: %d %s" % (len(out), len(bad), bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
