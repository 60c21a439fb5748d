"""Parity of RECORDED MI355X outputs (tests/golden/gpu_rows/*.npz, written on the GPU by tests/dump_rows_gpu.py / bench.py) against the CPU
oracle run live on this host - no GPU needed.  The same bars as the GPU tests: keys matched exactly, responses of matched rows bit-equal,
>= 99.5 % of the LAF rows within 1e-3 px, descriptors within 1e-3, every unmatched key and every row outside 1e-3 px accounted for by the
float64 referee (oracle/fp64_referee.py).  On the authoring host these files reproduce profiles/archive/r05_s1_offline_parity_*_authoring_host.json."""
import glob
import os
import re

import numpy as np
import pytest
import torch

import affnet_oracle as orc
import fp64_referee as rf
from conftest import load_gray

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "gpu_rows", "*.npz")))


def _image_and_n(name, golden_dir):
    m = re.match(r"sweep_synth_(\d+)x(\d+)_s(\d+)_n(\d+)", name)
    if m:
        return orc.synthetic_image(int(m.group(1)), int(m.group(2)), int(m.group(3))), int(m.group(4))
    m = re.match(r"sweep_graf_img6_n(\d+)", name)
    if m:
        return load_gray(os.path.join(golden_dir, "graf_img6.png")), int(m.group(1))
    m = re.match(r"configs_2__metric_configuration__image_(\d+)_", name)
    if m:
        return orc.synthetic_image(768, 1024, int(m.group(1))), 2000
    raise AssertionError("unknown fixture " + name)


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_recorded_mi355x_rows_match_the_reference(path, weights, golden_dir):
    assert FILES, "tests/golden/gpu_rows is empty"
    g = np.load(path)
    x, n = _image_and_n(os.path.basename(path), golden_dir)
    ex = orc.OracleExtractor(mrSize=5.192, num_features=n, border=5, num_Baum_iters=1, affnet_sd=weights["AffNet"], orinet_sd=weights["OriNet"])
    if "desc" in g.files:
        Lw, rw, Pw, Dw = orc.describe(x, ex, weights["HardNet"], do_ori=True, ps=32)
    else:
        Lw, rw = ex(x, do_ori=True)
        Dw = None
    ref = rf.Referee(ex, x.size(3), x.size(2))
    acc = rf.parity_account(ref, g["ids"], g["LAFs"], n)
    kg, kc = rf.keys_of(*g["ids"].astype(np.int64).T), rf.keys_of(*ex.keys.numpy().T)
    pos = {int(k): i for i, k in enumerate(kc)}
    gi = np.array([i for i, k in enumerate(kg) if int(k) in pos], dtype=np.int64)
    wi = np.array([pos[int(kg[i])] for i in gi], dtype=np.int64)
    assert len(g["ids"]) == len(kc) and len(gi) >= 0.995 * len(kc)
    assert np.array_equal(g["resp"][gi], rw.numpy()[wi]), "responses of matched keypoints must be bit-identical"
    dl = np.abs(g["LAFs"][gi] - Lw.numpy()[wi]).reshape(len(gi), -1).max(axis=1)
    assert (dl < 1e-3).mean() >= 0.995
    assert acc["unmatched_unexplained"] == 0, [r for r in acc["unmatched_rows"] if r["why"] in ("UNEXPLAINED", "NOT A DETECTOR CANDIDATE")]
    assert acc["rows_outside_1e-3_beyond_referee"] <= acc["beyond_budget"] and acc["rows_outside_1e-2"] == 0 and acc["rows_outside_5e-3_unexplained"] == 0, acc["rows_outside_1e-3_vs_fp64"]
    print("rows outside 1e-3 px vs the float64 referee:", [(r["gpu_vs_cpu_px"], r["gpu_vs_fp64_px"], r["cpu_vs_fp64_px"], r["beyond_referee"]) for r in acc["rows_outside_1e-3_vs_fp64"]])
    if Dw is not None:
        dd = np.abs(g["desc"][gi] - Dw.numpy()[wi]).max(axis=1)
        assert (dd[dl < 1e-3] < 1e-3).all() and np.percentile(dd, 99.5) < 1e-3
    print(os.path.basename(path), "matched %d / %d, rows >= 1e-3 px: %d, worst %.3g px, unmatched keys %d (all accounted for)"
          % (len(gi), len(kc), acc["rows_outside_1e-3"], dl.max(), acc["unmatched_keys"]))
