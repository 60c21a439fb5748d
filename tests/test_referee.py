"""CPU tests of oracle/fp64_referee.py (the float64 referee of the post-detector stages and the accounting of keys only one side
returns).  No GPU: the "other side" is a stand-in made from the referee's own float64 results rounded to float32 - a second fp32-class
evaluation of the same path, like the HIP kernels - plus deliberately broken variants that the accounting must flag."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import affnet_oracle as orc
import fp64_referee as rf
from conftest import load_gray


def test_fp64_nets_follow_the_fp32_restatement(weights):
    g = torch.Generator().manual_seed(3)
    p = torch.rand(24, 1, 32, 32, generator=g) * 255
    x = torch.randn(3, 5, 12, 12, generator=g, dtype=torch.float64)
    w = torch.randn(7, 5, 3, 3, generator=g, dtype=torch.float64)
    for st in (1, 2):
        assert torch.equal(rf.conv64(x, w, None, st, 1), F.conv2d(x, w, None, st, 1))
    A32 = orc.affnet_forward(weights["AffNet"], p).double()
    A64 = rf.affnet64(rf._double_sd(weights["AffNet"]), p.double())
    assert float((A32 - A64).abs().max()) < 5e-6
    v32 = orc.orinet_vector(weights["OriNet"], p).double()
    v64 = rf.orinet_vec64(rf._double_sd(weights["OriNet"]), p.double())
    assert float((v32 - v64).abs().max()) < 5e-6
    d32 = orc.hardnet_forward(weights["HardNet"], p).double()
    d64 = rf.hardnet64(rf._double_sd(weights["HardNet"]), p.double())
    assert float((d32 - d64).abs().max()) < 5e-6
    lafs = torch.tensor([[[0.1, 0.02, 0.5], [-0.01, 0.12, 0.4]], [[0.3, 0.0, 0.9], [0.0, 0.3, 0.1]]])
    img = orc.synthetic_image(60, 90, 4)
    assert float((orc.extract_patches(img, lafs, 32).double() - rf.extract_patches64(img.double(), lafs.double(), 32)).abs().max()) < 1e-2


@pytest.fixture(scope="module")
def graf_run(weights, golden_dir):
    x = load_gray(os.path.join(golden_dir, "graf_img1.png"))
    ex = orc.OracleExtractor(mrSize=5.192, num_features=500, border=5, num_Baum_iters=1, affnet_sd=weights["AffNet"], orinet_sd=weights["OriNet"])
    ex(x, do_ori=True)
    ref = rf.Referee(ex, x.size(3), x.size(2))
    # stand-in for the other side: the fp64 decisions and the fp64 LAFs rounded to fp32
    dec = ref.decisions(np.arange(len(ref.cand_keys)))
    val = ref.resp * dec["good"]
    order = np.argsort(-val, kind="stable")[:500]
    ids = np.stack([ref.octs[order], ref.levs[order], ref.pixs[order]], 1)
    L = ref.lafs_px(order).numpy().astype(np.float32)
    return ex, ref, dec, ids, L


def test_shape_stage_is_exposed_and_consistent(graf_run):
    ex, ref, dec, ids, L = graf_run
    st = ex.shape_stage
    assert st["good"].shape[0] == ex.detected["resp"].shape[0] == 750 and st["n_out"] == 500
    # the rows the oracle returned are the first N good candidates in response order (SparseImgRepresenter.py:152-158)
    good = st["good"].numpy()
    first = np.nonzero(good)[0][:500]
    assert np.array_equal(rf.keys_of(*ex.keys.numpy().T), ref.cand_keys[first])
    # the fp64 decisions agree with the CPU's except on borderline candidates
    for c in np.nonzero(good != dec["good"])[0]:
        assert ref.borderline(dec, c) is not None, "candidate %d flips between fp32 and fp64 without being borderline" % c


def test_accounting_explains_a_second_fp32_class_evaluation(graf_run):
    ex, ref, dec, ids, L = graf_run
    rec = rf.parity_account(ref, ids, L, 500, full=True)
    assert rec["unmatched_unexplained"] == 0, rec["unmatched_rows"]
    assert rec["rows_worse_than_cpu_vs_fp64"] == 0 and rec["rows_outside_1e-3_beyond_referee"] == 0 and rec["rows_outside_1e-2"] == 0
    # graf img1 holds near-isotropic AffNet outputs whose fp32 discriminant tr^2 - 4 det has the sign of its rounding error: the two sides
    # legitimately differ there, and every such key is traced to that decision or to the shifted cut
    assert rec["unmatched_keys"] > 0 and rec["unmatched_borderline_flips"] > 0
    whys = {r["why"] for r in rec["unmatched_rows"]}
    assert whys <= {"borderline discriminant", "borderline eigen_ratio", "borderline corner", "displaced at the top-N cut by a borderline row", "tie at the top-N cut"}
    ra = rec["referee_all_rows"]
    assert ra["rows"] == rec["matched"] and ra["gpu_vs_fp64_px_p50_p99_max"][2] < 1e-4          # the stand-in IS the fp64 result rounded to fp32
    assert ra["cpu_vs_fp64_px_p50_p99_max"][0] < 1e-3


def test_accounting_flags_a_dropped_keypoint_and_a_wrong_row(graf_run):
    ex, ref, dec, ids, L = graf_run
    kc = set(int(k) for k in rf.keys_of(*ex.keys.numpy().T))
    # (1) drop a keypoint both sides agree on (a kernel that loses rows for a wrong reason) and return a lower-ranked one instead
    both = [i for i in range(len(ids)) if int(rf.keys_of(*ids[i])) in kc]
    victim = both[len(both) // 2]
    val = ref.resp * dec["good"]
    nxt = np.argsort(-val, kind="stable")[500]
    ids2, L2 = ids.copy(), L.copy()
    ids2[victim] = [ref.octs[nxt], ref.levs[nxt], ref.pixs[nxt]]
    L2[victim] = ref.lafs_px([nxt]).numpy()[0]
    rec = rf.parity_account(ref, ids2, L2, 500)
    assert rec["unmatched_unexplained"] >= 1
    assert any(r["why"] == "UNEXPLAINED" for r in rec["unmatched_rows"])
    # (2) a matched row that is off by 6e-3 px although the CPU row is close to fp64
    L3 = L.copy()
    L3[both[3], 0, 0] += 6e-3
    rec = rf.parity_account(ref, ids, L3, 500)
    assert rec["rows_outside_1e-3"] >= 1 and rec["rows_worse_than_cpu_vs_fp64"] >= 1 and rec["rows_outside_1e-3_beyond_referee"] >= 1 and rec["rows_outside_1e-2"] == 0
    assert rec["rows_outside_5e-3_unexplained"] >= 1            # 5e-3 px off while the reference's own row is within 5e-3 px of float64: beyond the hard ceiling
    # (2b) the unconditional ceiling: 1.2e-2 px off counts whatever the conditioning; a row 3e-3 px off whose reference row is fine fails both clauses
    L3 = L.copy()
    L3[both[3], 0, 0] += 1.2e-2
    L3[both[4], 1, 1] += 3e-3
    rec = rf.parity_account(ref, ids, L3, 500)
    assert rec["rows_outside_1e-2"] == 1 and rec["rows_outside_1e-3_beyond_referee"] == 2 and rec["beyond_budget"] == 1
    assert rf.beyond_budget(2000) == 1 and rf.beyond_budget(4000) == 1 and rf.beyond_budget(4001) == 2 and rf.beyond_budget(8000) == 2
    assert not hasattr(rf, "REF_QUANTILE")                  # round 5's fitted quantile clause is gone
    # (3) a row that is not a detector candidate at all
    ids4 = ids.copy()
    ids4[both[5]] = [0, 0, 12345678]
    rec = rf.parity_account(ref, ids4, L, 500)
    assert rec["unmatched_unexplained"] >= 1 and any(r["why"] == "NOT A DETECTOR CANDIDATE" for r in rec["unmatched_rows"])
