"""GPU parity tests (run with -m gpu on an MI355X): every stage entry point of the C ABI and the
fused pipeline against the CPU oracle (oracle/affnet_oracle.py, itself pinned bit-for-bit to the
unmodified reference) and against the committed golden vectors.

Bars (BASELINE.json: LAFs / descriptors within 1e-3 of the reference CPU path):
  * pyramid, Hessian response, detector (keys, responses, LAFs), patch sampler: the HIP kernels
    replay the fp32 operation sequence of the reference's CPU kernels, so they must be EXACTLY equal
    to the golden vectors (unmodified reference on the authoring host, Intel AVX-512 / MKL).  The
    same torch build on another CPU (the GPU box is an AMD EPYC) rounds a few operators differently
    (affine_grid's bmm, the 3-channel centroid conv), so against the oracle run live on this host a
    small tolerance applies where those operators are involved; blur and Hessian are exact on both;
  * CNN outputs: different summation order on MFMA -> 2e-5 abs on O(1) outputs;
  * end to end: keypoints matched by integer key (octave, level, pixel); >= 99.5% must match, descriptors within 1e-3, >= 99.5 % of the
    matched LAF rows within 1e-3 px and none outside 5e-3 px unless the reference's own row is that far from float64.  The statement covers 100 % of the KEYS and ROWS (round 5, _referee()):
    every key only one side returns is traced to a borderline decision of the reference's shape filter or to the top-N cut it shifted
    (`unmatched_unexplained == 0`), and every row outside 1e-3 px is judged by a float64 evaluation of the post-detector stages:
    (a) |GPU - fp64| <= |CPU reference - fp64| + 1e-3 px, or (b) the CPU reference's own row is >= 1e-3 px from fp64 (ill-conditioned row)
    and |GPU - fp64| <= 4 |CPU - fp64|.  Rows meeting neither are listed and counted (`rows_outside_1e-3_beyond_referee`); the gate is the count
    budget oracle/fp64_referee.py states up front (<= 1 per 4000 matched rows against the oracle run live on this - foreign - host; the
    golden vectors of the authoring host are compared at the plain tolerance), and NO matched row may differ by 1e-2 px, unconditionally
    (`rows_outside_1e-2 == 0`).  Round 5's quantile clause and round 4's fitted bar (_laf_bar: max(1e-3 px, S (1e-5 + 4e-5 / |o|)), still
    recorded) are not gates.
  * both arithmetic modes of the CNN contractions (include/affnet_hip.h AFFNET_ARITH_*: "fp32" = exact fp32 MFMA, "fp32_split3" = fp32 as
    three bf16 terms on the bf16 MFMA) run the full-path cases with the SAME bars.
"""
import os

import numpy as np
import pytest
import torch

import affnet_oracle as orc
import fp64_referee as rf
from _rowmatch import match_rows, tie_groups
from conftest import load_gray, record_parity

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
ARITH = ["fp32", "fp32_split3", "fp32_split2h"]        # include/affnet_hip.h AFFNET_ARITH_FP32_MFMA (default) / AFFNET_ARITH_FP32_SPLIT3 / AFFNET_ARITH_FP32_SPLIT2H


def _laf_bar(Lw, ori_norm=None):
    """Per-row LAF tolerance in px that EVERY matched row must meet (BASELINE: 1e-3).  Two fp32 evaluations of the CNNs in different
    summation orders differ by ~1e-7 .. 1e-5 at the outputs: AffNet's shape entries relatively, OriNet's (y, x) vector absolutely - the
    angle atan2(y, x) then moves by that / |o|.  A frame of scale S = sqrt|det A| px moves by S x (relative + angular) perturbation, which
    exceeds 1e-3 px only for LARGE frames (hundreds of px: 1e-3 px is then a few ulp of an entry) or SHORT OriNet vectors.
    Bar = max(1e-3 px, S (1e-5 + 4e-5 / |o|)); without an OriNet vector (hand-crafted orientation) max(1e-3 px, 1e-5 S).
    Where 4e-5 comes from (measured, round 4, profiles/r04_*_parity_report.json): over all rows outside 1e-3 px of all BASELINE
    configurations the implied error of the OriNet vector, err / S * |o|, is <= 1.4e-5 - and it is the SAME to three digits in both
    arithmetic modes of the GPU path, whose own results agree with each other to 2.7e-5 px: the difference to the oracle is dominated
    by the rounding of the CPU reference's own fp32 convolutions (oneDNN blocking order), not by the kernels under test."""
    S = np.sqrt(np.abs(Lw[:, 0, 0] * Lw[:, 1, 1] - Lw[:, 0, 1] * Lw[:, 1, 0]))
    rel = 1e-5 if ori_norm is None else 1e-5 + 4e-5 / np.maximum(ori_norm, 1e-12)
    return np.maximum(1e-3, S * rel), S


_ORACLE_RUNS = {}


def _oracle_describe(tag, x, n, weights):
    """orc.describe() on image `x` (cached per module run under `tag`: the arithmetic-mode parametrisation reuses the CPU oracle's output)."""
    key = (tag, n)
    if key not in _ORACLE_RUNS:
        ex = _oracle(x, n, weights)
        _ORACLE_RUNS[key] = (ex,) + tuple(orc.describe(x, ex, weights["HardNet"], do_ori=True, ps=32))
    return _ORACLE_RUNS[key]


_REFEREES = {}


def _referee(ex, ids_g, L, hw, n_out, full=False):
    """oracle/fp64_referee.parity_account for one image (the Referee and its float64 results are cached per oracle run: the arithmetic-mode
    parametrisation asks about mostly the same rows)."""
    if id(ex) not in _REFEREES:
        _REFEREES[id(ex)] = (ex, rf.Referee(ex, hw[1], hw[0]))
    return rf.parity_account(_REFEREES[id(ex)][1], ids_g, L, n_out, full=full)


def _assert_accounted(rec):
    """The statements of the module docstring: every unmatched key explained; rows outside 1e-3 px that meet neither referee clause within the
    count budget fixed in oracle/fp64_referee.py (1 per 4000 matched rows, live oracle on a foreign host); the unconditional 1e-2 px ceiling."""
    acc = rec["accounting"]
    assert acc["unmatched_unexplained"] == 0, "keys only one side returns and no borderline decision explains: %s" % [r for r in acc["unmatched_rows"] if r["why"] in ("UNEXPLAINED", "NOT A DETECTOR CANDIDATE")]
    assert acc["rows_outside_1e-3_beyond_referee"] <= acc["beyond_budget"], ("rows outside 1e-3 px that are farther from the float64 referee than the CPU reference's own row + 1e-3 px "
                                                                             "and not bounded by an ill-conditioned reference row: %s" % [r for r in acc["rows_outside_1e-3_vs_fp64"] if r["beyond_referee"]])
    assert acc["rows_outside_1e-2"] == 0, "a matched LAF row differs by %.3g px: beyond the unconditional ceiling" % rec["laf_max_px"]
    assert acc["rows_outside_5e-3_unexplained"] == 0, "a matched LAF row differs by %.3g px although the reference's own row is within 5e-3 px of float64" % rec["laf_max_px"]


@pytest.fixture(scope="module")
def amd():
    import affnet_amd
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return affnet_amd


@pytest.fixture(scope="module")
def nets(amd, weights):
    A = amd.AffNetFast(PS=32); A.load_state_dict(weights["AffNet"]); A = A.to(DEV)
    O = amd.OriNetFast(PS=32); O.load_state_dict(weights["OriNet"]); O = O.to(DEV)
    H = amd.HardNet(); H.load_state_dict(weights["HardNet"]); H = H.to(DEV)
    return A, O, H


def _report(name, got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    d = np.abs(got - want)
    print("%s: max abs diff %.3g, mismatching elements %d / %d" % (name, d.max() if d.size else 0, int((d > 0).sum()), d.size))
    record_parity(name, max_abs_diff=float(d.max() if d.size else 0), mismatching_elements=int((d > 0).sum()), elements=int(d.size))
    return d


def _row_stats(name, ids_g, L, D, r, keys_w, Lw, Dw, rw, extra=None, ori_vec=None, ex=None, hw=None, n_out=0, full_referee=False):
    """Key-matched comparison of one image's rows with the oracle's; records the numbers in the parity report.  ori_vec: the oracle's
    OriNet output vectors before atan2 (row order of Lw): for every LAF row outside 1e-3 px the report then carries the vector's
    length - a short vector is what makes the angle (and with it the frame) sensitive to the CNN's summation order.  ex (+ hw = image
    (H, W), n_out = N): the OracleExtractor that produced the rows - the record then carries `accounting` = the float64 referee's verdict on
    every row outside 1e-3 px and the trace of every key only one side returned (_referee); callers assert it with _assert_accounted."""
    gi, wi = _match(ids_g, keys_w)
    n = len(keys_w)
    dl = np.abs(L[gi] - Lw[wi]).reshape(len(gi), -1).max(axis=1)
    worst = int(np.argmax(dl)) if len(gi) else 0
    nv = None if ori_vec is None else np.linalg.norm(np.asarray(ori_vec, dtype=np.float64), axis=1)
    bar, S = _laf_bar(np.asarray(Lw, dtype=np.float64)[wi], None if nv is None else nv[wi])
    extra = dict(extra or {})
    extra["rows_outside_combined_bar"] = int((dl > bar).sum())          # must be 0: asserted by the callers
    extra["frame_scale_px_p50_p99_max"] = [float(np.percentile(S, q)) for q in (50, 99, 100)] if len(gi) else []
    if len(gi):
        out = np.nonzero(dl >= 1e-3)[0]
        # every row outside 1e-3 px with what explains it: the frame's scale (1e-3 px of a 300 px frame is 3e-6 relative), the relative
        # error, and the OriNet vector length (a pure rotation by d_angle moves the entries by S d_angle)
        extra["rows_outside_1e-3"] = [{"key_octave_level_pixel": [int(v) for v in np.asarray(ids_g)[gi[k]]], "laf_err_px": float(dl[k]),
                                       "frame_scale_px": float(S[k]), "rel_err": float(dl[k] / max(S[k], 1e-30)), "bar_px": float(bar[k]),
                                       "orinet_norm": None if nv is None else float(nv[wi[k]]),
                                       "centre_err_px": float(np.abs(L[gi[k]][:, 2] - Lw[wi[k]][:, 2]).max()),
                                       "det_rel_err": float(abs(np.linalg.det(L[gi[k]][:, :2]) / np.linalg.det(Lw[wi[k]][:, :2]) - 1.0))} for k in out]
        if nv is not None:
            extra["orinet_norm_percentiles_all_rows_p1_p10_p50"] = [float(np.percentile(nv, q)) for q in (1, 10, 50)]
    rec = {"keypoints": int(n), "matched": int(len(gi)), "match_rate": len(gi) / float(max(n, 1)), "same_row_order": bool(len(gi) == n and np.array_equal(gi, wi)),
           "laf_p50_px": float(np.percentile(dl, 50)), "laf_p99_px": float(np.percentile(dl, 99)), "laf_max_px": float(dl.max()),
           "laf_rows_within_1e-3": float((dl < 1e-3).mean()), "worst_row_key_octave_level_pixel": [int(v) for v in np.asarray(ids_g)[gi[worst]]],
           "responses_equal": bool(np.array_equal(r[gi], rw[wi]))}
    dd = None
    if D is not None:
        dd = np.abs(D[gi] - Dw[wi]).max(axis=1)
        rec.update({"desc_max": float(dd.max()), "desc_p99": float(np.percentile(dd, 99)), "desc_rows_within_1e-3": float((dd < 1e-3).mean())})
    rec.update(extra or {})
    if os.environ.get("AFFNET_DUMP_ROWS"):          # the HIP path's rows, for analysis against the oracle on a host without a GPU
        os.makedirs(os.environ["AFFNET_DUMP_ROWS"], exist_ok=True)
        np.savez_compressed(os.path.join(os.environ["AFFNET_DUMP_ROWS"], "".join(c if c.isalnum() else "_" for c in name)[:120] + ".npz"),
                            ids=np.asarray(ids_g), LAFs=L, resp=r, **({} if D is None else {"desc": D}))
    if ex is not None:
        rec["accounting"] = _referee(ex, np.asarray(ids_g), L, hw, n_out, full=full_referee)
    record_parity(name, **rec)
    print(name, {k: v for k, v in rec.items() if k != "accounting"})
    if ex is not None:
        a = rec["accounting"]
        print("  accounting: unmatched keys %d (borderline flips %d, unexplained %d); rows outside 1e-3 px %d, worse than the CPU row vs fp64 %d %s"
              % (a["unmatched_keys"], a["unmatched_borderline_flips"], a["unmatched_unexplained"], a["rows_outside_1e-3"], a["rows_worse_than_cpu_vs_fp64"],
                 a.get("referee_all_rows", "")))
    return gi, wi, dl, dd, rec


# ------------------------------------------------------------------------------------------------
def test_mfma_fragment_layout(amd):
    from affnet_amd._lib import lib, ptr
    g = torch.Generator().manual_seed(0)
    A = torch.randn(16, 4, generator=g)
    B = torch.randn(4, 16, generator=g)
    out = torch.zeros(16, 16, device=DEV)
    Ad, Bd = A.to(DEV), B.to(DEV)
    assert lib.affnet_selftest_mfma(ptr(Ad), ptr(Bd), ptr(out), None) == 0
    torch.cuda.synchronize()
    assert torch.allclose(out.cpu(), A @ B, atol=1e-5), "16x16x4 f32 MFMA fragment layout assumption is wrong"


def test_gauss_blur_bit_exact(amd):
    from affnet_amd import engine
    from affnet_amd._lib import lib, ptr, check
    from affnet_amd.host_plan import gaussian_taps
    import ctypes as C
    x = orc.synthetic_image(200, 333, 2)           # ragged sizes: partial tiles on both axes
    ctx = engine.utility_ctx(torch.device(DEV))
    for sigma in [1.5198684153570665, 1.2262734984654078, 1.9465878414647133, 2.4525469969308156, 0.7, 3.1]:
        taps = gaussian_taps(sigma)
        k = taps.shape[0]
        xin = x[0, 0].to(DEV).contiguous()
        out = torch.empty_like(xin)
        buf = (C.c_float * (k * k))(*taps.reshape(-1).tolist())
        check(lib.affnet_gauss_blur(ctx, ptr(xin), ptr(out), 200, 333, buf, k, None), ctx, "blur")
        torch.cuda.synchronize()
        d = _report("blur sigma=%.3f k=%d" % (sigma, k), out.cpu().numpy(), orc.gaussian_blur(x, sigma)[0, 0].numpy())
        assert d.max() == 0.0


def test_hessian_response_bit_exact(amd):
    from affnet_amd import engine
    from affnet_amd._lib import lib, ptr, check
    x = orc.gaussian_blur(orc.synthetic_image(130, 70, 3), 1.6)
    ctx = engine.utility_ctx(torch.device(DEV))
    xin = x[0, 0].to(DEV).contiguous()
    out = torch.empty_like(xin)
    sigma = 2.0158736798317967
    check(lib.affnet_hessian_response(ctx, ptr(xin), ptr(out), 130, 70, np.float32(sigma ** 4), None), ctx, "hessian")
    torch.cuda.synchronize()
    d = _report("hessian", out.cpu().numpy(), orc.hessian_response(x, sigma)[0, 0].numpy())
    assert d.max() == 0.0


def test_sampler_bit_exact_and_golden(amd, golden_dir):
    g = np.load(os.path.join(golden_dir, "sampler_synth.npz"))
    x = orc.synthetic_image(240, 320, 1)
    lafs = torch.from_numpy(g["lafs"])
    for ps, key in [(32, "p32"), (41, "p41")]:
        got = amd.LAF.extract_patches(x.to(DEV), lafs.to(DEV), PS=ps).cpu()
        dg = _report("sampler PS=%d vs golden (reference, authoring host)" % ps, got.numpy(), g[key])
        dl = _report("sampler PS=%d vs oracle on this host" % ps, got.numpy(), orc.extract_patches(x, lafs, ps).numpy())
        assert dg.max() == 0.0
        assert dl.max() < 1e-2
    # ragged / empty inputs
    assert amd.LAF.extract_patches(x.to(DEV), lafs[:0].to(DEV), PS=32).shape == (0, 1, 32, 32)
    one = amd.LAF.extract_patches(x.to(DEV), lafs[:1].to(DEV), PS=19).cpu()
    assert np.abs(one.numpy() - orc.extract_patches(x, lafs[:1], 19).numpy()).max() < 1e-2


def _trunk_layers(sd, p):
    import torch.nn.functional as F
    x = orc.input_norm(p)
    outs = []
    for ci, bi, st in orc._TRUNK:
        x = F.conv2d(x, sd["features.%d.weight" % ci], None, stride=st, padding=1)
        x = F.batch_norm(x, sd["features.%d.running_mean" % bi], sd["features.%d.running_var" % bi], None, None, False, 0.1, 1e-5)
        x = F.relu(x)
        outs.append(x)
    return outs


@pytest.mark.parametrize("kind,name", [(0, "AffNet"), (2, "HardNet")])
def test_cnn_trunk_layer_by_layer(amd, weights, nets, kind, name):
    """Localises any kernel bug to one layer (the fused kernel dumps its LDS activation buffer)."""
    from affnet_amd import engine
    from affnet_amd._lib import lib, ptr, check
    g = torch.Generator().manual_seed(5)
    p = torch.rand(1, 1, 32, 32, generator=g) * 255
    want = _trunk_layers(weights[name], p)
    net = nets[0] if kind == 0 else nets[2]
    packed = net.packed_weights(torch.device(DEV))
    ctx = engine.utility_ctx(torch.device(DEV))
    pd = p[0, 0].to(DEV).contiguous()
    for layer in range(6):
        ref = want[layer][0]
        out = torch.zeros(ref.numel(), device=DEV)
        check(lib.affnet_cnn32_debug_layer(ctx, kind, ptr(packed), ptr(pd), layer, ptr(out), None), ctx, "debug_layer")
        torch.cuda.synchronize()
        d = _report("%s trunk layer %d %s" % (name, layer, tuple(ref.shape)), out.cpu().numpy().reshape(ref.shape), ref.numpy())
        assert d.max() < 5e-5 * max(1.0, float(ref.abs().max())), "layer %d" % layer


def test_cnn_outputs_vs_oracle_and_golden(amd, weights, nets, golden_dir):
    g = np.load(os.path.join(golden_dir, "cnn_random_patches.npz"))
    A, O, H = nets
    p = torch.from_numpy(g["patches"])
    for net, key, fn, name in [(A, "affnet", orc.affnet_forward, "AffNet"), (O, "orinet", orc.orinet_forward, "OriNet"),
                               (H, "hardnet", orc.hardnet_forward, "HardNet")]:
        got = net(p.to(DEV)).cpu().numpy()
        with torch.no_grad():
            want = fn(weights[name], p).numpy()
        d = _report(name, got, want)
        assert d.max() < 2e-5
        assert np.abs(got - g[key]).max() < 2e-5
    # ragged batch sizes incl. 0, 1 and a non-multiple of the 16-patch head tile
    for n in (0, 1, 17):
        assert H(p[:n].to(DEV)).shape == (n, 128)
        if n:
            assert np.abs(H(p[:n].to(DEV)).cpu().numpy() - g["hardnet"][:n]).max() < 2e-5
    # constant patch: std = 0 -> (x-mean)/(0+1e-7) = 0 -> finite output (reference behaviour)
    flat = torch.full((2, 1, 32, 32), 7.0)
    with torch.no_grad():
        want = orc.affnet_forward(weights["AffNet"], flat).numpy()
    assert np.abs(A(flat.to(DEV)).cpu().numpy() - want).max() < 2e-5


def _oracle(x, n, weights, th=None, do_ori=True, iters=1):
    ex = orc.OracleExtractor(mrSize=5.192, num_features=n, border=5, num_Baum_iters=iters, th=th,
                             affnet_sd=weights["AffNet"], orinet_sd=weights["OriNet"])
    return ex


def _keys(ids):
    ids = np.asarray(ids).astype(np.int64)
    return ids[:, 0] * (1 << 40) + ids[:, 1] * (1 << 32) + ids[:, 2]


def test_pyramid_and_detector_exact(amd, weights, nets, golden_dir):
    """Pyramid levels, detected keypoint identities, responses and LAFs (before AffNet): exactly the
    reference's (golden), and equal / within float noise of the oracle on this host."""
    g = np.load(os.path.join(golden_dir, "synth_240x320_s1_n300.npz"))
    x = orc.synthetic_image(240, 320, 1)
    ex = _oracle(x, 300, weights)
    ex(x, do_ori=False)
    det = amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=450, border=5, num_Baum_iters=0).to(DEV)
    L, r = det(x.to(DEV))
    sums, k = [], 0
    for o in range(len(ex.scale_pyr)):
        for l in range(5):
            lvl = det.scale_pyr[o][l].cpu().numpy()
            d = np.abs(lvl - ex.scale_pyr[o][l].numpy()).max()
            assert d == 0.0, "pyramid level (%d,%d) differs by %g" % (o, l, d)
            flat = lvl.reshape(-1)
            assert np.array_equal(flat[np.linspace(0, flat.size - 1, 16).astype(np.int64)], g["pyr_samples"][k])
            sums.append(lvl.astype(np.float64).sum())
            k += 1
    assert np.array_equal(np.array(sums), g["pyr_sums"]), "pyramid differs from the reference's (golden digests)"
    assert det.sigmas == ex.sigmas and det.pix_dists == ex.pix_dists
    # golden: the unmodified reference's multiScaleDetector output (C = 450), x mrSize, denormalised
    ids = det.last_ids.cpu().numpy()
    assert L.shape[0] == 450
    assert np.array_equal(r.cpu().numpy(), g["det_resp"]), "responses / row order differ from the reference"
    assert np.array_equal(ids[:, 0], g["det_oct"].astype(np.int32)) and np.array_equal(ids[:, 1], g["det_lev"].astype(np.int32))
    dg = _report("detector LAFs px vs golden", L.cpu().numpy(), g["det_LAFs_px"])
    # The 27-tap centroid sums are fmaf chains in the order of the reference's CPU conv2d: (level, ky, kx) for maps up to 6826 px (ATen's
    # native im2col + sgemm path), (ky, kx, level) above (oneDNN) - round 5; before, the second order was missing and a few sub-pixel
    # offsets were one ulp off (1.5e-5 px), which the patch sampling amplified to the 1e-3 px LAF outliers of rounds 2 - 4.
    assert dg.max() == 0.0, "detector LAFs differ from the reference's (golden, authoring host)"
    # live oracle on this host
    want = ex.detected
    got_keys = _keys(ids)
    want_keys = _keys(np.stack([want["oct"].numpy(), want["lev"].numpy(), want["pix"].numpy()], 1))
    assert np.array_equal(got_keys, want_keys), "keypoint identities / order differ from the oracle"
    assert np.array_equal(r.cpu().numpy(), want["resp"].numpy())
    # (live oracle on ANOTHER host: the small-map sgemm of MKL on an AMD CPU is not a sequential fmaf chain - tools/probes/cpu_conv_order.py
    # on the GPU box: no loop order matches below 6826 px, (ky, kx, level) above - so a few offsets of the small octaves may differ by an ulp)
    d = _report("detector LAFs px vs oracle on this host", L.cpu().numpy(), orc.denormalize_lafs(want["lafs"], 320, 240).numpy())
    assert d.max() < 1e-4


def _match(ids_got, keys_want):
    kg, kw = _keys(ids_got), _keys(keys_want)
    pos = {k: i for i, k in enumerate(kw)}
    gi = [i for i, k in enumerate(kg) if k in pos]
    wi = [pos[kg[i]] for i in gi]
    return np.array(gi, dtype=np.int64), np.array(wi, dtype=np.int64)


def _check_full(amd, nets, x, n, weights, want=None, min_match=0.995, name=None, arith="fp32", tag=None, full_referee=False):
    A, O, H = nets
    name = (name or "full path %dx%d n=%d" % (x.size(3), x.size(2), n)) + ("" if arith == "fp32" else " [arith %s]" % arith)
    ex, Lw, rw, Pw, Dw = _oracle_describe(tag or name, x, n, weights)
    det = amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=n, border=5, num_Baum_iters=1, AffNet=A, OriNet=O, arith=arith).to(DEV)
    res = det.run(x.to(DEV), do_ori=True, desc=H)
    assert det._ctx.arith == amd._lib.arith_code(arith)
    L, r, D = res["LAFs"].cpu().numpy(), res["responses"].cpu().numpy(), res["descriptors"].cpu().numpy()
    assert L.shape[0] == Lw.shape[0], "keypoint count %d vs %d" % (L.shape[0], Lw.shape[0])
    gi, wi = _match(res["ids"].cpu().numpy(), ex.keys.numpy())
    rate = len(gi) / float(Lw.shape[0])
    dl = np.abs(L[gi] - Lw.numpy()[wi])
    dd = np.abs(D[gi] - Dw.numpy()[wi])
    print("matched %.4f of %d keypoints; LAF max %.3g px (p99 %.3g); descriptor max %.3g; same order: %s"
          % (rate, Lw.shape[0], dl.max(), np.percentile(dl, 99), dd.max(), np.array_equal(gi, wi)))
    assert rate >= min_match
    # BASELINE tolerance 1e-3 px on matched rows; EVERY row must meet the combined bar of _laf_bar (absolute 1e-3 px, or the scale- and
    # |o|-aware bound for large frames / short OriNet vectors), >= 99.5% of the rows the plain 1e-3 px, none outside 5e-3 px.
    row_err = dl.reshape(len(gi), -1).max(axis=1)
    inside = row_err < 1e-3 + 1e-6 * np.abs(Lw.numpy()).max()
    print("rows within 1e-3 px: %.4f ; worst row %.3g px" % (inside.mean(), row_err.max()))
    rec = _row_stats(name, res["ids"].cpu().numpy(), L, D, r, ex.keys.numpy(), Lw.numpy(), Dw.numpy(),
                     rw.numpy(), ori_vec=None if ex.ori_vec is None else ex.ori_vec.numpy(), ex=ex, hw=(x.size(2), x.size(3)), n_out=n,
                     full_referee=full_referee)[4]
    _assert_accounted(rec)
    assert inside.mean() >= 0.995, "LAF error above tolerance"
    assert dd[inside].max() < 1e-3 and np.percentile(dd, 99.5) < 1e-3, "descriptor error above 1e-3"
    assert np.array_equal(r[gi], rw.numpy()[wi]), "responses of matched keypoints must be bit-identical"
    # patches through the public API (level choice on the device instead of host scipy)
    P = det.extract_patches_from_pyr(res["LAFs"], PS=32).cpu().numpy()
    dp = np.abs(P[gi] - Pw.numpy()[wi])
    print("descriptor patches max diff %.3g (0..255 scale)" % dp.max())
    assert np.percentile(dp, 99.9) < 5e-2
    if want is not None:   # committed golden vectors from the unmodified reference (authoring host): rows matched by response bits + centre
        g2, w2 = match_rows(r, L, want["resp"], want["LAFs"])
        eg = np.abs(L[g2] - want["LAFs"][w2]).reshape(len(g2), -1).max(axis=1)
        print("vs golden: matched %d of %d, rows within 1e-3 px %.4f, worst %.3g" % (len(g2), len(want["resp"]), (eg < 1e-3).mean(), eg.max()))
        record_parity(name + " vs the reference's golden output", golden_rows=int(len(want["resp"])), matched=int(len(g2)),
                      golden_rows_with_tied_responses=tie_groups(want["resp"]), laf_max_px=float(eg.max()), laf_rows_within_1e_3=float((eg < 1e-3).mean()),
                      desc_max=float(np.abs(D[g2] - want["desc"][w2]).max()))
        # host-independent leg: EVERY matched row within 1e-3 px of the unmodified reference's own output (all callers are <= 1024 x 768; measured worst 6.9e-4 px)
        assert len(g2) >= min_match * len(want["resp"]) and (eg < 1e-3).all(), "%d of %d golden rows outside 1e-3 px (worst %.3g)" % ((eg >= 1e-3).sum(), len(g2), eg.max())
        assert np.percentile(np.abs(D[g2] - want["desc"][w2]), 99.5) < 1e-3
    return det, res


@pytest.mark.parametrize("arith", ARITH)
def test_full_path_synthetic_golden(amd, nets, weights, golden_dir, arith):
    g = np.load(os.path.join(golden_dir, "synth_240x320_s1_n300.npz"))
    _check_full(amd, nets, orc.synthetic_image(240, 320, 1), 300, weights, want=g, name="synthetic 320x240 seed 1, 300 kp", arith=arith, tag="synth1")


@pytest.mark.parametrize("arith", ARITH)
def test_full_path_graf_img1_golden_n500(amd, nets, weights, golden_dir, arith):
    g = np.load(os.path.join(golden_dir, "graf_img1_n500.npz"))
    _check_full(amd, nets, load_gray(os.path.join(golden_dir, "graf_img1.png")), 500, weights, want=g, name="graf img1 800x640, 500 kp", arith=arith, tag="graf1")


@pytest.mark.parametrize("arith", ARITH)
def test_full_path_graf_img1_n2000_config2(amd, nets, weights, golden_dir, arith):
    """BASELINE.json configs[1]: test-graf/img1.png, 2000 kp, full path."""
    _check_full(amd, nets, load_gray(os.path.join(golden_dir, "graf_img1.png")), 2000, weights, name="configs[1]: graf img1 800x640, 2000 kp",
                arith=arith, tag="graf1", full_referee=True)


def test_threshold_mode_hesaffnet_as_shipped(amd, nets, weights, golden_dir):
    """hesaffnet.py:26,50: th=-1 => num=-1 => every 3-D maximum of resp+1 is kept, (o,l,pixel) order."""
    g = np.load(os.path.join(golden_dir, "synth_240x320_s1_thmode.npz"))
    x = orc.synthetic_image(240, 320, 1)
    A = nets[0]
    det = amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=300, border=5, num_Baum_iters=1, th=-1, AffNet=A).to(DEV)
    L, r = det(x.to(DEV))
    ex = _oracle(x, 300, weights, th=-1)
    Lw, rw = ex(x)
    counts = det._ctx.read_counts()
    print("th mode: detected %d (oracle %d), after shape filter %d (oracle %d, golden %d)"
          % (counts[0], ex.detected["resp"].numel(), L.shape[0], Lw.shape[0], g["LAFs"].shape[0]))
    assert counts[0] == ex.detected["resp"].numel(), "detector must find exactly the oracle's maxima"
    # the eigen-ratio / boundary filter compares float32 values against hard thresholds: a row whose test
    # value sits within float noise of the threshold may flip (no top-N afterwards to hide it in this mode)
    assert abs(L.shape[0] - Lw.shape[0]) <= 0.005 * Lw.shape[0]
    gi, wi = _match(det.last_ids.cpu().numpy(), ex.keys.numpy())
    assert len(gi) >= 0.995 * Lw.shape[0]
    assert np.abs(L.cpu().numpy()[gi] - Lw.numpy()[wi]).max() < 1e-3
    assert np.array_equal(r.cpu().numpy()[gi], rw.numpy()[wi])
    # ellipses of the rows both have, against the reference's own output file content (golden, authoring host): matched by centre
    ell = amd.LAF.LAFs2ell(L.cpu().numpy())
    ge = g["ells"]
    pos = {(round(float(e[0]), 2), round(float(e[1]), 2)): i for i, e in enumerate(ge)}
    pairs = [(i, pos[(round(float(e[0]), 2), round(float(e[1]), 2))]) for i, e in enumerate(ell) if (round(float(e[0]), 2), round(float(e[1]), 2)) in pos]
    record_parity("threshold mode (hesaffnet.py as shipped) 320x240", detected=int(counts[0]), rows=int(L.shape[0]), oracle_rows=int(Lw.shape[0]),
                  golden_rows=int(ge.shape[0]), ellipses_matched_by_centre=len(pairs))
    assert len(pairs) >= 0.995 * ge.shape[0], "only %d of %d ellipses of the reference's output found" % (len(pairs), ge.shape[0])
    a, b = np.array([p[0] for p in pairs]), np.array([p[1] for p in pairs])
    np.testing.assert_allclose(ell[a], ge[b], rtol=5e-3, atol=1e-6)


TH_CASES = [("graf_img1", "graf_img1.png"), ("cat", "hesaffnet_cat.png"), ("fox1", "hesaffnet_fox1.png")]
_TH_ORACLE = {}


@pytest.mark.parametrize("arith", ARITH)
@pytest.mark.parametrize("case", TH_CASES, ids=[c[0] for c in TH_CASES])
def test_threshold_mode_hesaffnet_as_shipped_full_size(amd, nets, weights, golden_dir, case, arith):
    """hesaffnet.py EXACTLY as the reference ships it (examples/hesaffnet/hesaffnet.py:24-60): th = -1 => num = -1
    (SparseImgRepresenter.py:33-37) - no feature budget, AffNetFast slot, no orientation - on the reference's own input images at their own
    sizes (test-graf/img1.png 800x640: 7075 rows; img/cat.png, img/fox1.png).  Against the oracle on this host WITH the accounting (there is
    no top-N cut in this mode, so every key only one side returns must be a borderline discriminant / eigen-ratio / corner decision of the
    shape filter: `unmatched_unexplained == 0`), and against the unmodified reference's golden output (tests/golden/make_golden_thmode.py):
    the same detector maxima, rows in (octave, level, pixel) order, matched rows within 1e-3 px, the rows of the Oxford file."""
    tag, fname = case
    g = np.load(os.path.join(golden_dir, "thmode_%s.npz" % tag))
    x = load_gray(os.path.join(golden_dir, fname))
    assert tuple(g["hw"]) == (x.size(2), x.size(3))
    det = amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=2000, border=5, num_Baum_iters=1, th=-1, AffNet=nets[0], arith=arith).to(DEV)
    L, r = det(x.to(DEV))
    ids = det.last_ids.cpu().numpy()
    L, r = L.cpu().numpy(), r.cpu().numpy()
    counts = det._ctx.read_counts()
    if tag not in _TH_ORACLE:
        ex = orc.OracleExtractor(mrSize=5.192, num_features=2000, border=5, num_Baum_iters=1, th=-1, affnet_sd=weights["AffNet"])
        Lw, rw = ex(x)
        _TH_ORACLE[tag] = (ex, Lw.numpy(), rw.numpy(), rf.Referee(ex, x.size(3), x.size(2)))
    ex, Lw, rw, ref = _TH_ORACLE[tag]
    assert np.array_equal(rw, g["resp"]) or abs(len(rw) - len(g["resp"])) <= 0.005 * len(rw)      # the live oracle may flip borderline rows against the golden host
    assert counts[0] == ex.detected["resp"].numel(), "the detector must find exactly the oracle's 3-D maxima (%d vs %d)" % (counts[0], ex.detected["resp"].numel())
    kg = _keys(ids)
    assert np.all(np.diff(kg) > 0), "threshold mode returns rows in (octave, level, pixel) order (HandCraftedModules.py:283-291, no top-k)"
    acc = rf.parity_account(ref, ids, L, -1)
    gi, wi = _match(ids, ex.keys.numpy())
    dl = np.abs(L[gi] - Lw[wi]).reshape(len(gi), -1).max(axis=1)
    # vs the reference's golden output: rows matched through the response bit pattern (+ centre for tied responses)
    g2, w2 = match_rows(r, L, g["resp"], g["LAFs"])
    eg = np.abs(L[g2] - g["LAFs"][w2]).reshape(len(g2), -1).max(axis=1)
    ell = amd.LAF.LAFs2ell(L)
    # ellipse coefficients (a, b, c) of (A A^T)^-1 relative to the row's largest coefficient (b ~ 0 for axis-aligned frames: no scale of its own)
    rel = np.abs(ell[g2] - g["ells"][w2]) / np.abs(g["ells"][w2][:, 2:]).max(axis=1, keepdims=True)
    sfx = "" if arith == "fp32" else " [arith %s]" % arith
    record_parity("threshold mode (hesaffnet.py as shipped, th = -1) %s %dx%d%s" % (tag, x.size(3), x.size(2), sfx), detected=int(counts[0]), rows=int(L.shape[0]),
                  oracle_rows=int(Lw.shape[0]), golden_rows=int(g["LAFs"].shape[0]), matched=int(len(gi)), laf_max_px=float(dl.max()),
                  laf_rows_outside_1e_3=int((dl >= 1e-3).sum()), responses_equal=bool(np.array_equal(r[gi], rw[wi])),
                  unmatched_keys=acc["unmatched_keys"], unmatched_borderline_flips=acc["unmatched_borderline_flips"], unmatched_unexplained=acc["unmatched_unexplained"],
                  unmatched_rows=acc["unmatched_rows"], rows_outside_1e_3_beyond_referee=acc["rows_outside_1e-3_beyond_referee"],
                  golden_matched=int(len(g2)), golden_laf_max_px=float(eg.max()), golden_rows_outside_1e_3=int((eg >= 1e-3).sum()),
                  golden_ellipse_centre_max_px=float(np.abs(ell[g2, :2] - g["ells"][w2, :2]).max()), golden_ellipse_max_rel=float(rel[:, 2:].max()))
    print("th mode %s%s: detected %d, rows %d (oracle %d, golden %d), unmatched keys %d (flips %d, unexplained %d), worst row %.3g px; vs golden matched %d, worst %.3g px, ellipse rel %.3g"
          % (tag, sfx, counts[0], L.shape[0], Lw.shape[0], g["LAFs"].shape[0], acc["unmatched_keys"], acc["unmatched_borderline_flips"], acc["unmatched_unexplained"], dl.max(),
             len(g2), eg.max(), rel[:, 2:].max()))
    assert acc["unmatched_unexplained"] == 0, [q for q in acc["unmatched_rows"] if not str(q["why"]).startswith("borderline")]
    assert all(str(q["why"]).startswith("borderline") for q in acc["unmatched_rows"]), "no top-N cut in this mode: every differing key must be a borderline decision"
    assert len(gi) >= 0.995 * Lw.shape[0] and np.array_equal(r[gi], rw[wi])
    assert (dl < 1e-3).all() or (acc["rows_outside_1e-3_beyond_referee"] <= acc["beyond_budget"] and acc["rows_outside_1e-2"] == 0), acc["rows_outside_1e-3_vs_fp64"]
    assert len(g2) >= 0.995 * g["LAFs"].shape[0]
    assert (eg < 1e-3).all(), "%d rows outside 1e-3 px vs the reference's golden output (worst %.3g)" % ((eg >= 1e-3).sum(), eg.max())
    assert np.abs(ell[g2, :2] - g["ells"][w2, :2]).max() < 1e-3 and rel[:, 2:].max() < 1e-3, "Oxford ellipse rows differ from the reference's file"


def test_threshold_mode_capacity_is_an_error_not_a_truncation(amd, nets, golden_dir):
    """th = -1 keeps every maximum: the row capacity (`max_keep`, default 16384 - graf / cat / fox1 need 7885 / 8049 / 9888 detector rows) must
    fail loudly when an image exceeds it, never return the first max_keep rows; raising it on the live object makes the same call succeed."""
    x = load_gray(os.path.join(golden_dir, "graf_img1.png")).to(DEV)
    det = amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=2000, border=5, num_Baum_iters=1, th=-1, AffNet=nets[0]).to(DEV)
    det.max_keep = 4096
    with pytest.raises(amd._lib.AffnetHipError) as e:
        det(x)
    assert "overflow" in str(e.value).lower() or "capacity" in str(e.value).lower(), str(e.value)
    det.max_keep = 16384
    L, r = det(x)
    assert L.shape[0] > 7000


def test_foreign_slots_staged_path_equals_fused(amd, nets, weights):
    """Any callable with the reference's slot signature works: the stage entry points give the same rows."""
    A, O, H = nets
    x = orc.synthetic_image(240, 320, 1).to(DEV)

    class Foreign(torch.nn.Module):      # not a _HipPatchNet -> forces the staged path
        def __init__(self, net):
            super().__init__()
            self.net, self.PS = net, 32

        def forward(self, patches, *a):
            return self.net(patches)

    fused = amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=300, border=5, num_Baum_iters=1, AffNet=A, OriNet=O).to(DEV)
    L1, r1 = fused(x, do_ori=True)
    staged = amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=300, border=5, num_Baum_iters=1, AffNet=Foreign(A),
                                                OriNet=Foreign(O)).to(DEV)
    L2, r2 = staged(x, do_ori=True)
    assert torch.equal(r1, r2) and torch.equal(fused.last_ids, staged.last_ids)
    assert float((L1 - L2).abs().max()) == 0.0


def test_foreign_affnet_with_several_baumberg_iterations(amd, nets):
    """SparseImgRepresenter.py:127-146 with a foreign AffNet slot and num_Baum_iters > 1: patches re-extracted along
    [base_A * LAF | centre] around the slot call (affnet_shape_iterate), bit-equal to the fused path on the native net."""
    A, O, H = nets
    x = orc.synthetic_image(240, 320, 1).to(DEV)

    class Foreign(torch.nn.Module):
        def __init__(self, net):
            super().__init__()
            self.net, self.PS = net, 32

        def forward(self, patches, *a):
            return self.net(patches)

    for iters in (2, 3):
        fused = amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=300, border=5, num_Baum_iters=iters, AffNet=A, OriNet=O).to(DEV)
        L1, r1 = fused(x, do_ori=True)
        staged = amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=300, border=5, num_Baum_iters=iters, AffNet=Foreign(A), OriNet=O).to(DEV)
        L2, r2 = staged(x, do_ori=True)
        assert torch.equal(r1, r2) and torch.equal(fused.last_ids, staged.last_ids), iters
        assert torch.equal(L1, L2), "iters = %d: max diff %g" % (iters, float((L1 - L2).abs().max()))
    one = amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=300, border=5, num_Baum_iters=1, AffNet=A, OriNet=O).to(DEV)
    assert not torch.equal(one(x, do_ori=True)[0], L1), "the iterations must change the frames"


def test_shape_filter_select_on_caller_supplied_unsorted_rows(amd, nets):
    """the public stage entry affnet_shape_filter_select promises the N largest responses among the rows that pass the shape
    filter for ANY caller-supplied rows.  Its selection kernel has a shortcut for response-sorted rows that used to be taken on the strength
    of a flag the context's LAST detector call left behind: shuffled rows handed to the entry after a sorted detection silently got the
    first N good rows instead of the N best.  The kernel now verifies the order itself.  Reference: SparseImgRepresenter.py:147-156
    (topk of response * good)."""
    import ctypes as C
    from affnet_amd._lib import lib, ptr, check
    from affnet_amd import engine
    A, O, H = nets
    x = orc.synthetic_image(240, 320, 1).to(DEV)
    det = amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=300, border=5, num_Baum_iters=1, AffNet=A, OriNet=O).to(DEV)
    det.run(x, do_ori=False)                                 # leaves the context in "sorted detections" mode (450 candidates > C)
    ctx = det._ctx
    P, F = ctx.cap_pre, ctx.cap_final
    st = engine.stream_of(torch.device(DEV))
    resp = torch.empty(P, device=DEV); lafs = torch.empty(P, 2, 3, device=DEV); ids = torch.empty(P, 3, dtype=torch.int32, device=DEV)
    cnt = torch.zeros(1, dtype=torch.int32, device=DEV)
    check(lib.affnet_detected_list(ctx.handle, ptr(resp), ptr(lafs), ptr(ids), ptr(cnt), st), ctx.handle, "detected_list")
    n = int(cnt.item())
    assert n == P == 450
    # a mildly anisotropic shape for every row (an isotropic frame has a zero discriminant in batch_eig2x2, Utils.py:168-175, and the
    # reference's fallback then rejects it): the eigen-ratio test passes, the boundary test decides
    Amat = torch.tensor([[1.2, 0.0], [0.1, 0.9]], device=DEV).repeat(P, 1, 1).contiguous()

    def select(r, l, i):
        r2 = torch.empty(F, device=DEV); l2 = torch.empty(F, 2, 3, device=DEV); i2 = torch.empty(F, 3, dtype=torch.int32, device=DEV)
        c2 = torch.zeros(1, dtype=torch.int32, device=DEV)
        check(lib.affnet_shape_filter_select(ctx.handle, ptr(r), ptr(l), ptr(i), ptr(Amat), ptr(cnt), ptr(r2), ptr(l2), ptr(i2), ptr(c2), st), ctx.handle,
              "shape_filter_select")
        k = int(c2.item())
        return r2[:k].clone(), l2[:k].clone(), i2[:k].clone()

    rs, ls, is_ = select(resp, lafs, ids)                    # sorted rows: the shortcut applies
    assert rs.numel() == F == 300 and bool((rs[:-1] >= rs[1:]).all())
    g = torch.Generator().manual_seed(11)
    perm = torch.randperm(n, generator=g).to(DEV)
    ru, lu, iu = select(resp[perm].contiguous(), lafs[perm].contiguous(), ids[perm].contiguous())
    # the same SET of rows in the same (descending response) order, whatever order the caller supplied them in
    assert torch.equal(ru, rs), "shuffled input rows: the selection is not the N largest responses"
    assert torch.equal(iu, is_) and torch.equal(lu, ls)


def test_two_stream_pipelining_gives_identical_results(amd, nets):
    """bench.py's throughput mode: detector on a second stream, two contexts alternating over an image stream."""
    A, O, H = nets
    imgs = [orc.synthetic_image(240, 320, s).to(DEV) for s in (1, 2, 3, 4, 5)]
    mk = lambda: amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=300, border=5, num_Baum_iters=1, AffNet=A, OriNet=O).to(DEV)
    ref = [mk().run(x, do_ori=True, desc=H) for x in imgs]
    dets, ds, cs = [mk(), mk()], torch.cuda.Stream(device=DEV), torch.cuda.Stream(device=DEV)
    torch.cuda.synchronize()
    out = []
    with torch.cuda.stream(cs):
        for i, x in enumerate(imgs):
            out.append(dets[i % 2].enqueue(x, do_ori=True, desc=H, det_stream=ds))
    torch.cuda.synchronize()
    for r, o in zip(ref, out):
        n = int(o["count"].item())
        assert n == r["LAFs"].shape[0]
        for k in ("LAFs", "responses", "descriptors", "ids"):
            assert torch.equal(o[k][:n], r[k]), k


@pytest.mark.parametrize("arith", ARITH)
def test_batched_launches_equal_single_image_calls(amd, nets, arith):
    """BASELINE configs[2] runs as (B,1,H,W) batches: every kernel launch covers the B images.  Each image of the
    batch must come out bit-identical to its own single-image call - incl. an image with no detections at all
    (ragged per-image row counts) and the 'fewer detections than the budget' branch."""
    A, O, H = nets
    imgs = [orc.synthetic_image(240, 320, s) for s in (1, 2, 3)] + [torch.zeros(1, 1, 240, 320), orc.synthetic_image(240, 320, 7)]
    xb = torch.cat(imgs, 0).to(DEV)
    for nfeat in (300, 4000):
        mk = lambda: amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=nfeat, border=5, num_Baum_iters=1, AffNet=A,
                                                        OriNet=O, arith=arith).to(DEV)
        single = []
        for x in imgs:
            try:
                single.append(mk().run(x.to(DEV), do_ori=True, desc=H))
            except RuntimeError:          # flat image: the single-image API raises like the reference
                single.append(None)
        det = mk()
        batched = det.run_batch(xb, do_ori=True, desc=H)
        assert len(batched) == len(imgs)
        for b, (one, got) in enumerate(zip(single, batched)):
            if one is None:
                assert got["LAFs"].shape[0] == 0
                continue
            assert got["LAFs"].shape == one["LAFs"].shape, (b, got["LAFs"].shape, one["LAFs"].shape)
            for k in ("LAFs", "responses", "descriptors", "ids"):
                assert torch.equal(got[k], one[k]), (nfeat, b, k)
        # rows past each image's count are zero in the capacity-sized tensors
        r = det.enqueue(xb, do_ori=True, desc=H)
        torch.cuda.synchronize()
        for b, n in enumerate(r["count"].cpu().tolist()):
            assert float(r["LAFs"][b, n:].abs().sum()) == 0.0 and float(r["descriptors"][b, n:].abs().sum()) == 0.0


@pytest.mark.parametrize("iters", [2, 3])
def test_iterated_affnet_shape(amd, nets, weights, iters):
    """num_Baum_iters > 1 (SparseImgRepresenter.py:127-146): base_A = A_i * base_A with patches re-extracted from
    [base_A * LAF | centre] between the AffNet passes."""
    A, O, H = nets
    x = orc.synthetic_image(240, 320, 1)
    det = amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=300, border=5, num_Baum_iters=iters, AffNet=A, OriNet=O).to(DEV)
    res = det.run(x.to(DEV), do_ori=True, desc=H)
    ex = orc.OracleExtractor(mrSize=5.192, num_features=300, border=5, num_Baum_iters=iters, affnet_sd=weights["AffNet"],
                             orinet_sd=weights["OriNet"])
    Lw, rw, Pw, Dw = orc.describe(x, ex, weights["HardNet"], do_ori=True, ps=32)
    gi, wi = _match(res["ids"].cpu().numpy(), ex.keys.numpy())
    row_err = np.abs(res["LAFs"].cpu().numpy()[gi] - Lw.numpy()[wi]).reshape(len(gi), -1).max(axis=1)
    dd = np.abs(res["descriptors"].cpu().numpy()[gi] - Dw.numpy()[wi]).max()
    print("iters %d: matched %d / %d, LAF worst %.3g px, rows within 1e-3 px %.4f, descriptor worst %.3g" %
          (iters, len(gi), len(ex.keys), row_err.max(), (row_err < 1e-3).mean(), dd))
    record_parity("num_Baum_iters = %d (AffNet) 320x240" % iters, keypoints=int(len(ex.keys)), matched=int(len(gi)), laf_max_px=float(row_err.max()),
                  rows_within_1e_3=float((row_err < 1e-3).mean()), desc_max=float(dd))
    assert len(gi) >= 0.99 * len(ex.keys) and (row_err < 1e-3).mean() >= 0.99 and dd < 1e-3
    # differs from the single-iteration result (the iteration really happened)
    one = amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=300, border=5, num_Baum_iters=1, AffNet=A, OriNet=O).to(DEV)
    L1, _ = one(x.to(DEV), do_ori=True)
    assert L1.shape != res["LAFs"].shape or float((L1 - res["LAFs"]).abs().max()) > 1e-3


@pytest.mark.parametrize("nlevels", [2, 4])
def test_other_nlevels(amd, nets, weights, nlevels):
    """nlevels ctor kwarg (SparseImgRepresenter.py:20; ScalePyramid nLevels): 4 / 6 levels per octave instead of 5."""
    A, O, H = nets
    x = orc.synthetic_image(240, 320, 1)
    det = amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=300, border=5, num_Baum_iters=1, nlevels=nlevels, AffNet=A,
                                             OriNet=O).to(DEV)
    res = det.run(x.to(DEV), do_ori=True, desc=H)
    assert len(det.scale_pyr[0]) == nlevels + 2
    ex = orc.OracleExtractor(mrSize=5.192, num_features=300, border=5, num_Baum_iters=1, nlevels=nlevels, affnet_sd=weights["AffNet"],
                             orinet_sd=weights["OriNet"])
    Lw, rw, Pw, Dw = orc.describe(x, ex, weights["HardNet"], do_ori=True, ps=32)
    gi, wi = _match(res["ids"].cpu().numpy(), ex.keys.numpy())
    row_err = np.abs(res["LAFs"].cpu().numpy()[gi] - Lw.numpy()[wi]).reshape(len(gi), -1).max(axis=1)
    print("nlevels %d: matched %d / %d, worst row %.3g px" % (nlevels, len(gi), len(ex.keys), row_err.max()))
    record_parity("nlevels = %d 320x240" % nlevels, keypoints=int(len(ex.keys)), matched=int(len(gi)), laf_max_px=float(row_err.max()))
    assert len(gi) >= 0.99 * len(ex.keys) and (row_err < 1e-3).mean() >= 0.99
    assert np.array_equal(res["responses"].cpu().numpy()[gi], rw.numpy()[wi])
    assert np.abs(res["descriptors"].cpu().numpy()[gi] - Dw.numpy()[wi]).max() < 1e-3


def test_custom_respnet_slot(amd, nets, weights):
    """RespNet slot (SparseImgRepresenter.py:24,38-41): any callable(level, sigma) -> response map.  (1) a torch restatement of
    HessianResp through the slot gives the built-in detector's keypoints; (2) a different response (sigma^2 |Laplacian|) gives the
    oracle's keypoints for the same function."""
    import torch.nn.functional as F
    A, O, H = nets
    x = orc.synthetic_image(240, 320, 1)

    def hess(level, sigma):
        return orc.hessian_response(level.cpu(), sigma).to(level.device)

    def lap(level, sigma):                         # runs on whatever device the level lives on
        k = torch.tensor([[0.0, 1.0, 0.0], [1.0, -4.0, 1.0], [0.0, 1.0, 0.0]], device=level.device).view(1, 1, 3, 3)
        return (F.conv2d(F.pad(level, (1, 1, 1, 1), "replicate"), k).abs() * float(sigma * sigma))

    mk = lambda rn: amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=300, border=5, num_Baum_iters=1, AffNet=A, OriNet=O,
                                                       RespNet=rn).to(DEV)
    builtin = mk(None).run(x.to(DEV), do_ori=True, desc=H)
    via_slot = mk(hess).run(x.to(DEV), do_ori=True, desc=H)
    for k in ("LAFs", "responses", "ids", "descriptors"):
        assert torch.equal(builtin[k], via_slot[k]), k
    got = mk(lap).run(x.to(DEV), do_ori=True, desc=H)
    ex = orc.OracleExtractor(mrSize=5.192, num_features=300, border=5, num_Baum_iters=1, affnet_sd=weights["AffNet"],
                             orinet_sd=weights["OriNet"], resp_fn=lambda lv, s: lap(lv, s))
    Lw, rw = ex(x, do_ori=True)
    gi, wi = _match(got["ids"].cpu().numpy(), ex.keys.numpy())
    row_err = np.abs(got["LAFs"].cpu().numpy()[gi] - Lw.numpy()[wi]).reshape(len(gi), -1).max(axis=1)
    print("custom RespNet: matched %d / %d, rows within 1e-3 px %.4f" % (len(gi), len(ex.keys), (row_err < 1e-3).mean()))
    record_parity("custom RespNet slot (sigma^2 |Laplacian|) 320x240", keypoints=int(len(ex.keys)), matched=int(len(gi)),
                  rows_within_1e_3=float((row_err < 1e-3).mean()))
    assert len(gi) >= 0.99 * len(ex.keys) and (row_err < 1e-3).mean() >= 0.99     # conv2d on GPU vs CPU: response ties can flip
    assert not torch.equal(got["ids"], builtin["ids"])


def test_handcrafted_default_slots(amd, nets, golden_dir):
    """OrientationDetector / AffineShapeEstimator kernels against the unmodified reference classes
    (golden) - unit level, the default-constructed extractor, and 4 Baumberg iterations."""
    from affnet_amd.HandCraftedModules import OrientationDetector, AffineShapeEstimator
    g = np.load(os.path.join(golden_dir, "handcrafted_slots.npz"))
    p = torch.from_numpy(g["patches"]).to(DEV)
    ang = OrientationDetector(patch_size=19)(p).cpu().numpy()
    same = np.abs(ang - g["ori_angles"]) < 1e-6
    print("orientation: %d / %d angles identical" % (same.sum(), len(same)))
    record_parity("hand-crafted OrientationDetector(19): golden patches", patches=int(len(same)), identical_angles=int(same.sum()),
                  flipped=[int(i) for i in np.nonzero(~same)[0]])
    assert same.all(), "orientation bins flipped for golden patches %s (a flipped 10-degree bin is a wrong frame)" % np.nonzero(~same)[0].tolist()
    R = OrientationDetector(patch_size=19)(p, return_rot_matrix=True).cpu()
    assert np.abs(R.numpy()[same] - orc.angles_to_rotation(torch.from_numpy(g["ori_angles"])).numpy()[same]).max() < 1e-6
    A = AffineShapeEstimator(patch_size=19)(p).cpu().numpy()
    _report("Baumberg shape", A, g["baum_A"])
    assert np.abs(A - g["baum_A"]).max() < 2e-5
    x = orc.synthetic_image(240, 320, 1).to(DEV)
    det = amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=300, border=5, num_Baum_iters=0).to(DEV)     # default slots
    L, r = det(x, do_ori=True)
    assert np.array_equal(np.sort(r.cpu().numpy()), np.sort(g["default_resp"]))
    gi, wi = match_rows(r.cpu().numpy(), L.cpu().numpy(), g["default_resp"], g["default_LAFs"])
    assert len(gi) == len(g["default_resp"])
    row_err = np.abs(L.cpu().numpy()[gi] - g["default_LAFs"][wi]).reshape(len(gi), -1).max(axis=1)
    print("default extractor: rows within 1e-3 px %.4f (a flipped orientation bin rotates the frame)" % (row_err < 1e-3).mean())
    record_parity("default-constructed extractor (hand-crafted slots) 320x240", rows=int(len(L)), rows_within_1e_3=float((row_err < 1e-3).mean()),
                  worst_row_px=float(row_err.max()))
    assert (row_err < 1e-3).mean() >= 0.99
    # frames that differ must differ by a pure rotation: same centre, same determinant
    assert np.abs(L.cpu().numpy()[gi][:, :, 2] - g["default_LAFs"][wi][:, :, 2]).max() < 1e-3
    # LAFs2ellT (section 8f row 3) on the reference's own LAFs
    ell = amd.LAF.LAFs2ellT(torch.from_numpy(g["default_LAFs"]).to(DEV)).cpu().numpy()
    rel = np.abs(ell - g["default_ellT"]) / (np.abs(g["default_ellT"]) + 1e-6 * np.abs(g["default_ellT"]).max())
    print("LAFs2ellT: worst relative error %.3g" % rel.max())
    record_parity("LAFs2ellT on the device", worst_relative_error=float(rel.max()))
    assert rel.max() < 2e-4 and np.array_equal(ell[:, :2], g["default_ellT"][:, :2])
    det = amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=300, border=5, num_Baum_iters=4).to(DEV)
    L, r = det(x, do_ori=False)
    # The golden holds two rows with the SAME response (rows 206 / 207, centres 50.77 px apart: an exact tie, whose order torch.topk
    # leaves unspecified); rows are therefore matched by response bits + centre, and then every row has to agree.  (Round 2 compared row
    # by row, saw the swapped pair as a 50.77 px "error" and explained it with near-singular shapes - wrongly.)
    assert L.shape == g["baum4_LAFs"].shape and np.array_equal(np.sort(r.cpu().numpy()), np.sort(g["baum4_resp"]))
    gi, wi = match_rows(r.cpu().numpy(), L.cpu().numpy(), g["baum4_resp"], g["baum4_LAFs"])
    assert len(gi) == len(g["baum4_resp"])
    row_err = np.abs(L.cpu().numpy()[gi] - g["baum4_LAFs"][wi]).reshape(len(gi), -1).max(axis=1)
    print("Baumberg x4: worst row %.3g px, rows within 1e-3 px %.4f" % (row_err.max(), (row_err < 1e-3).mean()))
    record_parity("4 Baumberg iterations 320x240 vs golden", rows=int(len(L)), matched=int(len(gi)), golden_rows_with_tied_responses=tie_groups(g["baum4_resp"]),
                  rows_within_1e_3=float((row_err < 1e-3).mean()), worst_row_px=float(row_err.max()),
                  note="rows matched by response bits + centre (the golden holds one exact response tie)")
    assert (row_err < 1e-3).all(), "rows %s" % np.nonzero(row_err >= 1e-3)[0].tolist()
    # hesaffBaum.py:40 as shipped: SIXTEEN iterations, and the ellipses the script writes (LAFs2ellT, :47) - against the unmodified reference's
    det = amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=300, border=5, num_Baum_iters=16).to(DEV)
    L, r = det(x, do_ori=False)
    assert L.shape == g["baum16_LAFs"].shape and np.array_equal(np.sort(r.cpu().numpy()), np.sort(g["baum16_resp"]))
    gi, wi = match_rows(r.cpu().numpy(), L.cpu().numpy(), g["baum16_resp"], g["baum16_LAFs"])
    assert len(gi) == len(g["baum16_resp"])
    row_err = np.abs(L.cpu().numpy()[gi] - g["baum16_LAFs"][wi]).reshape(len(gi), -1).max(axis=1)
    ell = amd.LAF.LAFs2ellT(L).cpu().numpy()
    rel = np.abs(ell[gi] - g["baum16_ellT"][wi]) / (np.abs(g["baum16_ellT"][wi]) + 1e-6 * np.abs(g["baum16_ellT"]).max())
    print("Baumberg x16 (hesaffBaum.py): worst row %.3g px, rows within 1e-3 px %.4f, ellipses worst relative error %.3g" % (row_err.max(), (row_err < 1e-3).mean(), rel.max()))
    record_parity("16 Baumberg iterations (hesaffBaum.py:40) 320x240 vs golden", rows=int(len(L)), matched=int(len(gi)),
                  golden_rows_with_tied_responses=tie_groups(g["baum16_resp"]), rows_within_1e_3=float((row_err < 1e-3).mean()),
                  worst_row_px=float(row_err.max()), ellipse_worst_relative_error=float(rel.max()))
    assert (row_err < 1e-3).mean() >= 0.99 and row_err.max() < 1e-2, "rows %s" % np.nonzero(row_err >= 1e-3)[0].tolist()
    assert rel.max() < 1e-3


def test_matching_snn_and_homography_check(amd, golden_dir):
    """SURVEY section 8f row 1 (test() of train_AffNet_test_on_graffity.py:290-305): MFMA distance / SNN ratio kernel and the
    homography consistency check against the reference's golden outputs and the oracle."""
    from affnet_amd import Losses, ReprojectionStuff as RS
    g = np.load(os.path.join(golden_dir, "match_graf16_n500.npz"))
    L1, D1, L2, D2, H = (torch.from_numpy(g[k]) for k in ("LAFs1", "desc1", "LAFs2", "desc2", "H"))
    d1, d2 = D1.to(DEV), D2.to(DEV)
    # full distance matrix vs the oracle (fp32 MFMA vs sgemm summation order)
    full = Losses.distance_matrix_vector(d1, d2).cpu()
    want = orc.distance_matrix_vector(D1, D2)
    _report("distance matrix", full.numpy(), want.numpy())
    assert float((full - want).abs().max()) < 2e-6 * 2 + 1e-5
    t1, t2, md, md2 = RS.match_snn(d1, d2, 0.8)
    assert np.abs(md.cpu().numpy() - g["min_dist"]).max() < 1e-5 and np.abs(md2.cpu().numpy() - g["min_2nd"]).max() < 1e-5
    got = set(zip(t1.tolist(), t2.tolist()))
    ref = set(zip(g["tent1"].tolist(), g["tent2"].tolist()))
    print("tentatives: %d (reference %d), common %d" % (len(got), len(ref), len(got & ref)))
    record_parity("SNN matching graf 1-6, 500 kp", tentatives=len(got), reference_tentatives=len(ref), common=len(got & ref),
                  min_dist_max_abs_diff=float(np.abs(md.cpu().numpy() - g["min_dist"]).max()))
    assert len(got & ref) >= len(ref) - 1 and abs(len(got) - len(ref)) <= 1
    assert t1.tolist() == sorted(t1.tolist())                          # row order, like boolean-mask indexing
    # homography check on the reference's own tentatives: identical rows
    rt1, rt2 = torch.from_numpy(g["tent1"]), torch.from_numpy(g["tent2"])
    rp = RS.reprojectLAFs(L2[rt2].to(DEV), torch.inverse(H)).cpu().numpy()
    err = np.abs(rp - g["reproj"]).reshape(len(rp), -1).max(axis=1)
    scale = np.abs(g["reproj"]).reshape(len(rp), -1).max(axis=1)
    print("reprojectLAFs: worst abs err %.3g, worst err / row scale %.3g" % (err.max(), (err / scale).max()))
    assert (err <= 1e-4 + 1e-5 * scale).all()      # fp32: linH subtracts nearly equal terms; coordinates reach ~7000 px
    gd, plain, in2 = RS.get_GT_correspondence_indexes(L1[rt1].to(DEV), L2[rt2].to(DEV), H, dist_threshold=6)
    assert np.array_equal(plain.cpu().numpy(), g["gt_plain"]) and np.array_equal(in2.cpu().numpy(), g["gt_idx"])
    assert np.abs(gd.cpu().numpy() - g["gt_dist"]).max() < 0.05
    # ragged / edge sizes: n1 not a multiple of 64, n2 not a multiple of 16, a single row, zero rows
    gen = torch.Generator().manual_seed(5)
    for n1, n2 in ((1, 17), (67, 33), (130, 1), (0, 5)):
        a = torch.nn.functional.normalize(torch.randn(n1, 128, generator=gen), dim=1)
        b = torch.nn.functional.normalize(torch.randn(n2, 128, generator=gen), dim=1)
        t1, t2, md, md2 = RS.match_snn(a.to(DEV), b.to(DEV), 0.9)
        if n1 == 0:
            assert t1.numel() == 0
            continue
        w = orc.match_snn(a, b, 0.9)
        assert np.abs(md.cpu().numpy() - w[0].numpy()).max() < 1e-5 and np.array_equal(torch.stack([t1, t2]).cpu().numpy(), torch.stack([w[3], w[4]]).numpy())
    # size-independent property at the benchmark size: matching a set against a shuffled copy of itself recovers the
    # permutation at distance ~sqrt(1e-6).  (For identical vectors |a|^2 + |b|^2 - 2 a.b can round below -1e-6: sqrt gives
    # NaN, which torch.min - and this kernel - propagate; allow a handful of such rows.)
    x = torch.nn.functional.normalize(torch.randn(3000, 128, generator=gen), dim=1)
    perm = torch.randperm(3000, generator=gen)
    t1, t2, md, _ = RS.match_snn(x.to(DEV), x[perm].to(DEV), 0.8)
    md = md.cpu()
    ok = ~torch.isnan(md)
    assert int((~ok).sum()) <= 6 and float(md[ok].max()) < 2e-3
    inv = torch.empty_like(perm); inv[perm] = torch.arange(3000)
    assert torch.equal(t2.cpu(), inv[t1.cpu()]) and t1.numel() >= 3000 - 6


def test_just_shape_config1(amd, nets, golden_dir):
    """BASELINE.json configs[0]: detect_affine_shape on a patch column (examples/just_shape)."""
    g = np.load(os.path.join(golden_dir, "just_shape_column.npz"))
    col = g["column"]
    patches = torch.from_numpy(col.reshape(-1, 1, 32, 32).astype(np.float32) / 255.0)
    out = nets[0](patches.to(DEV)).reshape(-1, 4).cpu().numpy()
    assert np.abs(out - g["affine"]).max() < 2e-5 and np.all(out[:, 1] == 0)


def test_edge_cases(amd, nets):
    A, O, H = nets
    det = amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=100, border=5, num_Baum_iters=1, AffNet=A, OriNet=O).to(DEV)
    with pytest.raises(RuntimeError, match="no keypoints"):
        det(torch.zeros(1, 1, 64, 80, device=DEV))            # flat image: the reference raises in torch.cat([])
    # fewer detections than requested: small image, large budget -> (o,l,pixel) order branch
    x = orc.synthetic_image(96, 72, 9)
    det = amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=5000, border=5, num_Baum_iters=0).to(DEV)
    L, r = det(x.to(DEV))
    ex = orc.OracleExtractor(mrSize=5.192, num_features=5000, border=5, num_Baum_iters=0)
    Lw, rw = ex(x)
    assert L.shape == Lw.shape and np.array_equal(r.cpu().numpy(), rw.numpy())
    assert np.abs(L.cpu().numpy() - Lw.numpy()).max() < 1e-4          # (o,l,pixel) row order, LAFs to float noise


def test_capacity_overflow_is_an_error_not_a_truncation(amd, nets):
    """Fixed-capacity device lists: when a list overflows the library reports AFFNET_ERR_CAPACITY through the one read-back
    (affnet_read_counts) instead of silently dropping keypoints."""
    from affnet_amd._lib import AffnetHipError
    A, O, H = nets
    x = orc.synthetic_image(240, 320, 1).to(DEV)
    det = amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=300, border=5, num_Baum_iters=1, AffNet=A, OriNet=O).to(DEV)
    det.raw_div = 1 << 20                      # raw-maxima capacity 256 per octave; octave 0 of this image has ~700 maxima
    with pytest.raises(AffnetHipError, match="overflow"):
        det.run(x, do_ori=True)
    # the same flag without a host synchronisation: enqueue() hands out a device view of the per-image overflow counters
    r = det.enqueue(x, do_ori=True)
    torch.cuda.synchronize()
    assert int(r["overflow"][0]) != 0
    det.raw_div = 4
    r = det.enqueue(x, do_ori=True)
    torch.cuda.synchronize()
    assert int(r["overflow"][0]) == 0 and int(r["count"][0]) == 300
    assert det.run(x, do_ori=True)["LAFs"].shape[0] == 300
    det = amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=300, border=5, num_Baum_iters=1, th=-1, AffNet=A).to(DEV)
    det.max_keep = 64                          # threshold mode keeps every maximum: capacity 64 rows cannot hold ~1700
    with pytest.raises(AffnetHipError, match="overflow"):
        det.run(x)


def test_full_size_properties_config3(amd, nets, weights):
    """1024x768, 2000 kp (the metric's configuration): size-independent properties + determinism."""
    A, O, H = nets
    x = orc.synthetic_image(768, 1024, 0).to(DEV)
    det = amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=2000, border=5, num_Baum_iters=1, AffNet=A, OriNet=O).to(DEV)
    r1 = det.run(x, do_ori=True, desc=H)
    r2 = det.run(x, do_ori=True, desc=H)
    assert r1["LAFs"].shape == (2000, 2, 3) and r1["descriptors"].shape == (2000, 128)
    for k in ("LAFs", "responses", "descriptors", "ids"):
        assert torch.equal(r1[k], r2[k]), "non-deterministic " + k
    resp = r1["responses"].cpu().numpy()
    assert np.all(np.diff(resp) <= 0), "responses must be sorted descending (torch.topk order)"
    assert np.abs(np.linalg.norm(r1["descriptors"].cpu().numpy(), axis=1) - 1.0).max() < 1e-4
    L = r1["LAFs"].cpu().numpy()
    assert np.isfinite(L).all() and (L[:, 0, 2] >= 0).all() and (L[:, 0, 2] <= 1024).all() and (L[:, 1, 2] <= 768).all()
    assert len(np.unique(_keys(r1["ids"].cpu().numpy()))) == 2000


@pytest.mark.parametrize("arith", ARITH)
def test_metric_configuration_batched_b32_vs_oracle(amd, nets, weights, golden_dir, arith):
    """SURVEY section 8 row g1 - parity AT THE METRIC'S CONFIGURATION: bench.py's fused call = 32 synthetic 1024x768 images per
    launch, 2000 kp each (BASELINE.json configs[2]).  First, middle and last image of the batch against the oracle (keys, responses,
    LAFs, DESCRIPTORS), and batched == single-image bit-equality at this size."""
    A, O, H = nets
    B, seeds = 32, list(range(32))
    xb = torch.cat([orc.synthetic_image(768, 1024, s) for s in seeds], 0).to(DEV)
    mk = lambda: amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=2000, border=5, num_Baum_iters=1, AffNet=A, OriNet=O, arith=arith).to(DEV)
    sfx = "" if arith == "fp32" else " [arith %s]" % arith
    det = mk()
    batched = det.run_batch(xb, do_ori=True, desc=H)
    assert len(batched) == B and all(b["LAFs"].shape == (2000, 2, 3) and b["descriptors"].shape == (2000, 128) for b in batched)
    # every image of the batch: unit descriptors, sorted responses, unique keys (cheap, covers all 32)
    for b in batched:
        assert np.all(np.diff(b["responses"].cpu().numpy()) <= 0)
        assert np.abs(np.linalg.norm(b["descriptors"].cpu().numpy(), axis=1) - 1.0).max() < 1e-4
        assert len(np.unique(_keys(b["ids"].cpu().numpy()))) == 2000
    single = mk()
    for i in (0, 15, 31):
        got = batched[i]
        one = single.run(xb[i:i + 1], do_ori=True, desc=H)
        for k in ("LAFs", "responses", "descriptors", "ids"):
            assert torch.equal(got[k], one[k]), "image %d of the batch differs from its single-image call in %s" % (i, k)
        ex, Lw, rw, Pw, Dw = _oracle_describe("synth768x1024 seed %d" % seeds[i], orc.synthetic_image(768, 1024, seeds[i]), 2000, weights)
        gi, wi, dl, dd, rec = _row_stats("configs[2] metric configuration: image %d of a 32-image batch, 1024x768, 2000 kp%s" % (i, sfx),
                                         got["ids"].cpu().numpy(), got["LAFs"].cpu().numpy(), got["descriptors"].cpu().numpy(),
                                         got["responses"].cpu().numpy(), ex.keys.numpy(), Lw.numpy(), Dw.numpy(), rw.numpy(), ori_vec=ex.ori_vec.numpy(),
                                         ex=ex, hw=(768, 1024), n_out=2000, full_referee=(i == 0))
        assert rec["match_rate"] >= 0.995, rec
        assert rec["responses_equal"], "responses of matched keypoints must be bit-identical"
        assert rec["laf_rows_within_1e-3"] >= 0.995, rec
        _assert_accounted(rec)
        assert rec["desc_rows_within_1e-3"] >= 0.995, rec
        assert dd[dl < 1e-3].max() < 1e-3, "descriptor of a geometrically matching row off by more than 1e-3"
    # images 0, 1, 2, 31 of the batch (+ seed 63 = the last image of bench.py's second launch, as a single-image call: batched == single
    # bit for bit) against the UNMODIFIED reference's own output on the authoring host (tests/golden/make_golden_config3.py) - host
    # independent, no referee: EVERY matched row within 1e-3 px, every descriptor within 1e-3.  Rows matched through the response bit
    # pattern (the reference emits no integer keys).  These are the images bench.py's golden leg reads back (bench.py GOLDEN_SEEDS).
    for seed in (0, 1, 2, 31, 63):
        g = np.load(os.path.join(golden_dir, "synth_768x1024_s%d_n2000.npz" % seed))
        assert int(g["seed"]) == seed
        got = batched[seed] if seed < B else single.run(orc.synthetic_image(768, 1024, seed).to(DEV), do_ori=True, desc=H)
        gi, wi = match_rows(got["responses"].cpu().numpy(), got["LAFs"].cpu().numpy(), g["resp"], g["LAFs"])
        dl = np.abs(got["LAFs"].cpu().numpy()[gi] - g["LAFs"][wi]).reshape(len(gi), -1).max(axis=1)
        dd = np.abs(got["descriptors"].cpu().numpy()[gi] - g["desc"][wi]).max(axis=1)
        record_parity("configs[2] metric configuration: seed %d vs the reference's golden output%s" % (seed, sfx), keypoints=2000, matched=int(len(gi)),
                      same_row_order=bool(np.array_equal(gi, wi)), laf_max_px=float(dl.max()), rows_outside_1e_3=int((dl >= 1e-3).sum()),
                      desc_max=float(dd.max()), desc_rows_outside_1e_3=int((dd >= 1e-3).sum()))
        assert len(gi) >= 0.995 * 2000, (seed, len(gi))
        assert (dl < 1e-3).all() and (dd < 1e-3).all(), "seed %d: %d LAF rows / %d descriptors outside 1e-3 vs the golden output (worst %.3g px)" % (seed, (dl >= 1e-3).sum(), (dd >= 1e-3).sum(), dl.max())


@pytest.mark.parametrize("arith", ARITH)
@pytest.mark.parametrize("case", ["hesaffnet_cat", "hesaffnet_fox1", "synth_481x641_s5"])
def test_odd_sized_reference_images_full_path(amd, nets, weights, golden_dir, case, arith):
    """Odd-sized level 0 through the whole path (SparseImgRepresenter.py:189-209 on inputs the reference itself ships):
    examples/hesaffnet/img/cat.png (598 x 1000), fox1.png (1000 x 563) and a synthetic 641 x 481 image, 2000 kp, do_ori + HardNet,
    against the oracle on this host AND the unmodified reference's golden output (tests/golden/make_golden_oddsize.py); then the same
    image inside a batch of 4 (different images around it): bit-identical to its single-image call."""
    g = np.load(os.path.join(golden_dir, case + "_n2000.npz"))
    x = load_gray(os.path.join(golden_dir, case + ".png")) if case.startswith("hesaffnet") else orc.synthetic_image(481, 641, 5)
    assert tuple(x.shape[2:]) == tuple(int(v) for v in g["hw"])
    det, res = _check_full(amd, nets, x, 2000, weights, want=g, name="odd-sized input %s %dx%d, 2000 kp" % (case, x.size(3), x.size(2)), arith=arith,
                           tag="odd " + case)
    A, O, H = nets
    h, w = x.shape[2:]
    others = [orc.synthetic_image(h, w, s) for s in (11, 12, 13)]
    xb = torch.cat([others[0], x, others[1], others[2]], 0).to(DEV)
    bdet = amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=2000, border=5, num_Baum_iters=1, AffNet=A, OriNet=O, arith=arith).to(DEV)
    batched = bdet.run_batch(xb, do_ori=True, desc=H)
    for k in ("LAFs", "responses", "descriptors", "ids"):
        assert torch.equal(batched[1][k], res[k]), "image 1 of the batch of 4 differs from its single-image call in %s" % k


def test_exact_response_ties_are_resolved_deterministically(amd, nets):
    """A periodic image has many keypoints with bit-identical responses, so the top-k cuts fall inside groups of ties.  torch.topk's
    choice among equal values is unspecified; here ties are taken in (octave, level, pixel) order: the selected RESPONSES equal the
    oracle's as a multiset, and repeated runs give identical rows (no atomics decide who is in)."""
    g = torch.Generator().manual_seed(3)
    blk = torch.rand(1, 1, 48, 64, generator=g) * 255.0
    x = blk.repeat(1, 1, 5, 5).contiguous()                              # 240 x 320, period (48, 64)
    A, O, H = nets
    for n in (100, 300):
        runs = []
        for _ in range(3):
            det = amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=n, border=5, num_Baum_iters=0).to(DEV)
            L, r = det(x.to(DEV))
            runs.append((L.clone(), r.clone(), det.last_ids.clone()))
        for L, r, ids in runs[1:]:
            assert torch.equal(L, runs[0][0]) and torch.equal(r, runs[0][1]) and torch.equal(ids, runs[0][2])
        ex = orc.OracleExtractor(mrSize=5.192, num_features=n, border=5, num_Baum_iters=0)
        Lw, rw = ex(x)
        got, want = np.sort(runs[0][1].cpu().numpy()), np.sort(rw.numpy())
        assert got.shape == want.shape and np.array_equal(got, want), "selected responses differ from the oracle's as a multiset"
        u, c = np.unique(want, return_counts=True)
        assert c.max() >= 4, "the image was meant to produce exact ties"
        # within the tie group at the cut, the taken keypoints are the first ones in key order
        keys = _keys(runs[0][2].cpu().numpy())
        last = runs[0][1].cpu().numpy() == runs[0][1].cpu().numpy().min()
        assert np.all(np.diff(keys[last]) > 0) or last.sum() == 1
    # full path with ties: deterministic too
    det = amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=200, border=5, num_Baum_iters=1, AffNet=A, OriNet=O).to(DEV)
    a = det.run(x.to(DEV), do_ori=True, desc=H)
    b = det.run(x.to(DEV), do_ori=True, desc=H)
    for k in ("LAFs", "responses", "descriptors", "ids"):
        assert torch.equal(a[k], b[k]), k


def test_lazy_shape_evaluation_gives_identical_rows(amd, nets, weights):
    """affnet_config.lazy_shape_rows: AffNet first runs on the 1.2 N best of the 1.5 N response-sorted candidates and on the rest only
    for images that do not reach N survivors of the shape filter (device-side decision).  Every setting must give bit-identical
    rows: all-at-once (0), the default, a window so small that the second pass is needed, and the unsorted case (fewer than C
    candidates: the detections come in (octave, level, pixel) order and the second pass must always run)."""
    A, O, H = nets
    x = orc.synthetic_image(240, 320, 1).to(DEV)

    def run(n, lazy):
        det = amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=n, border=5, num_Baum_iters=1, AffNet=A, OriNet=O).to(DEV)
        det.lazy_shape_rows = lazy
        r = det.run(x, do_ori=True, desc=H)
        return r, int(det._ctx.counter_view(3)[0]), int(det._ctx.read_counts()[0])

    ref, ev0, n_det = run(300, 0)
    assert n_det == 450 and ev0 == 450                                   # all C candidates through AffNet, like the reference
    for lazy, want_eval in ((-1, None), (360, None), (301, 450), (449, None), (450, 450), (10 ** 6, 450)):
        got, ev, _ = run(300, lazy)
        for k in ("LAFs", "responses", "descriptors", "ids"):
            assert torch.equal(got[k], ref[k]), (lazy, k)
        assert ev in ((360 if lazy in (-1, 360) else lazy), 450), (lazy, ev)
        if want_eval is not None:
            assert ev == want_eval, (lazy, ev)
    _, ev_default, _ = run(300, -1)
    record_parity("lazy shape evaluation 320x240, N = 300", candidates=450, evaluated_default_window=ev_default, evaluated_all_at_once=ev0)
    # unsorted detections: total candidates n with 1.2 N < n <= 1.5 N
    big = amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=100000, border=5, num_Baum_iters=0).to(DEV)
    big.max_keep = 1 << 17
    small = orc.synthetic_image(120, 160, 3)
    n_total = big(small.to(DEV))[0].shape[0]
    N = int(np.ceil(n_total / 1.4))
    assert 1.2 * N < n_total <= int(1.5 * N), (n_total, N)
    outs = []
    for lazy in (0, -1):
        det = amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=N, border=5, num_Baum_iters=1, AffNet=A, OriNet=O).to(DEV)
        det.lazy_shape_rows = lazy
        outs.append(det.run(small.to(DEV), do_ori=True, desc=H))
        assert int(det._ctx.counter_view(3)[0]) == n_total               # not response-sorted: everything is evaluated
    for k in ("LAFs", "responses", "descriptors", "ids"):
        assert torch.equal(outs[0][k], outs[1][k]), k
    ex = orc.OracleExtractor(mrSize=5.192, num_features=N, border=5, num_Baum_iters=1, affnet_sd=weights["AffNet"], orinet_sd=weights["OriNet"])
    Lw, rw = ex(small, do_ori=True)
    gi, wi = _match(outs[1]["ids"].cpu().numpy(), ex.keys.numpy())
    assert len(gi) >= 0.99 * Lw.shape[0] and abs(outs[1]["LAFs"].shape[0] - Lw.shape[0]) <= 2
    assert np.abs(outs[1]["LAFs"].cpu().numpy()[gi] - Lw.numpy()[wi]).max() < 1e-3


@pytest.mark.parametrize("arith", ARITH)
def test_hip_graph_replay_equals_eager(amd, nets, arith):
    """affnet_graph_capture_extract / affnet_graph_launch: the whole path as one HIP graph gives bit-identical results to the eager call,
    for new image content copied into the captured input, on single images and on batches."""
    A, O, H = nets
    imgs = [orc.synthetic_image(240, 320, s).to(DEV) for s in (1, 2, 3)]
    mk = lambda: amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=300, border=5, num_Baum_iters=1, AffNet=A, OriNet=O, arith=arith).to(DEV)
    ref = [mk().run(x, do_ori=True, desc=H) for x in imgs]
    det = mk()
    cap = det.capture(imgs[0], do_ori=True, desc=H)
    for rep in range(2):
        for x, want in zip(imgs, ref):
            got = cap.run(x)
            for k in ("LAFs", "responses", "descriptors", "ids"):
                assert torch.equal(got[k], want[k]), (rep, k)
    # eager calls on the same extractor still work next to the captured graph
    again = det.run(imgs[1], do_ori=True, desc=H)
    assert torch.equal(again["LAFs"], ref[1]["LAFs"])
    # batch of 3 in one graph
    xb = torch.cat(imgs, 0)
    capb = mk().capture(xb, do_ori=True, desc=H)
    r = capb.launch(xb)
    torch.cuda.synchronize()
    for b, want in enumerate(ref):
        n = int(r["count"][b])
        assert n == want["LAFs"].shape[0]
        for k in ("LAFs", "responses", "descriptors", "ids"):
            assert torch.equal(r[k][b, :n], want[k]), (b, k)


def test_two_contexts_two_host_threads(amd, nets):
    """include/affnet_hip.h: different contexts may be driven from different host threads / streams concurrently (the library
    keeps no process-global state).  Two threads, each with its own extractor (= context) and stream, alternate over images; every
    result must be bit-identical to the single-threaded one."""
    import threading
    A, O, H = nets
    imgs = [orc.synthetic_image(240, 320, s).to(DEV) for s in (1, 2, 3, 4)]
    mk = lambda: amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=300, border=5, num_Baum_iters=1, AffNet=A, OriNet=O).to(DEV)
    ref = [mk().run(x, do_ori=True, desc=H) for x in imgs]
    A.packed_weights(torch.device(DEV)); O.packed_weights(torch.device(DEV)); H.packed_weights(torch.device(DEV))   # shared, read-only from here on
    torch.cuda.synchronize()
    out, errs = {}, []
    dets = [mk(), mk()]                 # built by the main thread: nn.Module.to() on the shared nets is not a thread-safe operation

    def worker(tid):
        try:
            det, st = dets[tid], torch.cuda.Stream(device=DEV)
            with torch.cuda.stream(st):
                for rep in range(6):
                    for i, x in enumerate(imgs):
                        r = det.run(x, do_ori=True, desc=H)
                        out[(tid, rep, i)] = {k: r[k].clone() for k in ("LAFs", "responses", "descriptors", "ids")}
            st.synchronize()
        except Exception as e:      # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    assert len(out) == 2 * 6 * len(imgs)
    for (tid, rep, i), r in out.items():
        for k in ("LAFs", "responses", "descriptors", "ids"):
            assert torch.equal(r[k], ref[i][k]), (tid, rep, i, k)


@pytest.mark.parametrize("kw,exact", [(dict(nlevels=1), True), (dict(init_sigma=0.5, nlevels=2), True), (dict(init_sigma=0.4), False)])
def test_pyramid_variants(amd, kw, exact):
    """Constructor kwargs the mirror accepts must give the reference's pyramid: nlevels = 1 (a 35 x 35 Gaussian) and
    init_sigma <= 0.5 (octave 0 keeps the raw image and its own blur sequence, later octaves restart at init_sigma:
    HandCraftedModules.py:25-31,49).  Bit-exact where the reference's convolutions are reproducible: with init_sigma = 0.4 the
    octaves >= 1 are blurred with 3 x 3 kernels, for which ATen leaves oneDNN and runs im2col + MKL sgemm on inputs of <= 20480
    elements (Convolution.cpp use_mkldnn: "for some case, native is faster"); that sgemm's K order is MKL-internal (not the
    row-major fmaf chain oneDNN and this kernel use), so those levels agree to 2 ulp only and a few keypoints may flip."""
    x = orc.synthetic_image(240, 320, 1)
    det = amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=300, border=5, num_Baum_iters=0, **kw).to(DEV)
    L, r = det(x.to(DEV))
    ex = orc.OracleExtractor(mrSize=5.192, num_features=300, border=5, num_Baum_iters=0, **kw)
    Lw, rw = ex(x)
    assert len(det.scale_pyr) == len(ex.scale_pyr) and det.sigmas == ex.sigmas
    worst, where = 0.0, None
    for o in range(len(ex.scale_pyr)):
        assert len(det.scale_pyr[o]) == len(ex.scale_pyr[o])
        for l in range(len(ex.scale_pyr[o])):
            d = float(np.abs(det.scale_pyr[o][l].cpu().numpy() - ex.scale_pyr[o][l].numpy()).max())
            if d > worst:
                worst, where = d, (o, l)
    gi, wi = _match(det.last_ids.cpu().numpy(), ex.keys.numpy())
    record_parity("pyramid variant %s" % kw, pyramid_max_abs_diff=worst, worst_level=list(where) if where else None, rows=int(L.shape[0]),
                  oracle_rows=int(Lw.shape[0]), keys_matched=int(len(gi)))
    if exact:
        assert worst == 0.0, "pyramid level %s differs by %g" % (where, worst)
        assert L.shape == Lw.shape and np.array_equal(r.cpu().numpy(), rw.numpy())
        assert np.abs(L.cpu().numpy() - Lw.numpy()).max() < 1e-4
    else:
        assert worst < 2e-4 and (where is None or where[0] >= 1), "octave 0 (11-tap-free, oneDNN path) must stay exact; got %g at %s" % (worst, where)
        assert L.shape == Lw.shape and len(gi) >= 0.97 * Lw.shape[0]
        assert np.abs(L.cpu().numpy()[gi] - Lw.numpy()[wi]).max() < 1e-3
    # changing nlevels on a live object rebuilds the plan (the context cache key includes it)
    if "nlevels" not in kw:
        det.nlevels = 2
        det(x.to(DEV))
        assert len(det.scale_pyr[0]) == 4


def test_cli_entry_points(amd, golden_dir, tmp_path):
    """L4 scripts run unchanged in spirit: same argv, same output files."""
    import subprocess, sys
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "img1.txt"
    env = dict(os.environ, HESAFFNET_TH="none")
    subprocess.check_call([sys.executable, os.path.join(root, "examples/hesaffnet/hesaffnet.py"),
                           os.path.join(golden_dir, "graf_img1.png"), str(out), "500"], env=env)
    lines = open(out).read().split("\n")
    assert lines[0].strip() == "1.0" and int(lines[1]) == 500
    ell = np.loadtxt(out, skiprows=2)
    assert ell.shape == (500, 5) and (ell[:, 2] * ell[:, 4] - ell[:, 3] ** 2 > 0).all()   # positive definite ellipses
    # VALUES: the file's rows against the unmodified reference's own output for this image and budget (tests/golden/graf_img1_n500.npz;
    # those frames carry OriNet's rotation, the CLI's do not - the Oxford ellipse (A A^T)^-1 does not depend on it), rows matched by centre
    gold = np.load(os.path.join(golden_dir, "graf_img1_n500.npz"))
    want = orc.lafs_to_ellipses(gold["LAFs"])
    pos = {(round(float(e[0]), 2), round(float(e[1]), 2)): i for i, e in enumerate(want)}
    pairs = [(i, pos[(round(float(e[0]), 2), round(float(e[1]), 2))]) for i, e in enumerate(ell) if (round(float(e[0]), 2), round(float(e[1]), 2)) in pos]
    a, b = np.array([q[0] for q in pairs]), np.array([q[1] for q in pairs])
    rel = np.abs(ell[a] - want[b]) / np.maximum(np.abs(want[b]), 1e-6)
    record_parity("hesaffnet.py CLI output vs the reference's golden rows (graf img1, 500 kp)", rows=500, matched_by_centre=len(pairs),
                  centre_max_px=float(np.abs(ell[a, :2] - want[b, :2]).max()), ellipse_max_rel=float(rel[:, 2:].max()))
    assert len(pairs) >= 0.995 * 500, "only %d of 500 golden rows found in the CLI output" % len(pairs)
    assert np.abs(ell[a, :2] - want[b, :2]).max() < 1e-3 and rel[:, 2:].max() < 1e-3, "ellipse values of the CLI output differ from the reference's"
    # the SHIPPED mode (hesaffnet.py:26 th = -1; no HESAFFNET_TH in the environment): every maximum kept, nfeats ignored - the file against the
    # rows the unmodified reference writes for this image (tests/golden/thmode_graf_img1.npz "ells"), matched by centre
    out3 = tmp_path / "img1_th.txt"
    env3 = {k: v for k, v in os.environ.items() if k != "HESAFFNET_TH"}
    subprocess.check_call([sys.executable, os.path.join(root, "examples/hesaffnet/hesaffnet.py"), os.path.join(golden_dir, "graf_img1.png"), str(out3), "2000"], env=env3)
    lines = open(out3).read().split("\n")
    gt = np.load(os.path.join(golden_dir, "thmode_graf_img1.npz"))["ells"]
    ell3 = np.loadtxt(out3, skiprows=2)
    assert lines[0].strip() == "1.0" and int(lines[1]) == ell3.shape[0] and abs(ell3.shape[0] - gt.shape[0]) <= 0.005 * gt.shape[0], (lines[1], gt.shape)
    pos = {(round(float(e[0]), 2), round(float(e[1]), 2), round(float(e[2]), 6)): i for i, e in enumerate(gt)}
    pairs = [(i, pos[k]) for i, k in enumerate((round(float(e[0]), 2), round(float(e[1]), 2), round(float(e[2]), 6)) for e in ell3) if k in pos]
    a, b = np.array([q[0] for q in pairs]), np.array([q[1] for q in pairs])
    rel3 = np.abs(ell3[a] - gt[b]) / np.abs(gt[b][:, 2:]).max(axis=1, keepdims=True)     # relative to the row's largest ellipse coefficient
    record_parity("hesaffnet.py CLI output AS SHIPPED (th = -1) vs the reference's golden file rows (graf img1)", rows=int(ell3.shape[0]), golden_rows=int(gt.shape[0]),
                  matched=len(pairs), centre_max_px=float(np.abs(ell3[a, :2] - gt[b, :2]).max()), ellipse_max_rel=float(rel3[:, 2:].max()))
    assert len(pairs) >= 0.99 * gt.shape[0], "only %d of %d golden rows found in the CLI output" % (len(pairs), gt.shape[0])
    assert np.abs(ell3[a, :2] - gt[b, :2]).max() < 1e-3 and rel3[:, 2:].max() < 1e-3
    g = np.load(os.path.join(golden_dir, "just_shape_column.npz"))
    col = tmp_path / "column.png"
    Image.fromarray(g["column"]).save(col)
    out2 = tmp_path / "shape.txt"
    subprocess.check_call([sys.executable, os.path.join(root, "examples/just_shape/detect_affine_shape.py"), str(col), str(out2)])
    got = np.loadtxt(out2)
    assert got.shape == (64, 4) and np.abs(got - g["affine"]).max() < 1e-4


def test_cli_threshold_script_variants(amd, weights, golden_dir, tmp_path):
    """The `...Th.py` variants of the reference's scripts: extract_geom_and_desc_upisupTh.py (OnePassSIR, num_features = -1,
    th = argv[3]; its line 63) and extract_geomOriTh.py (th = 28.41, OriNet slot, LAFs saved with np.save as lafs1.npy; its lines 31, 62, 88).  Their
    files against the same calls through the Python API: identical rows."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    img = os.path.join(golden_dir, "graf_img1.png")
    FC = amd.AffNetFastFullConv(PS=32); FC.load_state_dict(weights["AffNet"]); FC = FC.to(DEV)
    O = amd.OriNetFast(PS=32); O.load_state_dict(weights["OriNet"]); O = O.to(DEV)
    x = load_gray(img).to(DEV)
    out = tmp_path / "graf_th"
    subprocess.check_call([sys.executable, os.path.join(root, "examples/hesaffnet/extract_geomOriTh.py"), img, str(out)])
    L = np.load(str(out) + ".lafs1.npy")
    D = np.load(str(out) + ".desc.npy")
    det = amd.OnePassSIR(mrSize=5.192, num_features=-1, th=28.41, border=15, num_Baum_iters=1, AffNet=FC, OriNet=O).to(DEV)
    Lw, rw = det(x, do_ori=True)
    assert L.ndim == 3 and L.shape[1:] == (2, 3) and L.shape[0] > 100 and D.shape == (L.shape[0], 128)
    assert np.array_equal(L, Lw.cpu().numpy()), "lafs1.npy differs from OnePassSIR(th = 28.41)(img, do_ori = True)"
    out2 = tmp_path / "graf_th.txt"
    subprocess.check_call([sys.executable, os.path.join(root, "examples/hesaffnet/extract_geom_and_desc_upisupTh.py"), img, str(out2), "28.41"])
    lines = open(out2).read().split("\n")
    ell = np.loadtxt(out2, skiprows=2)
    det2 = amd.OnePassSIR(mrSize=5.192, num_features=-1, th=28.41, border=15, num_Baum_iters=1, AffNet=FC).to(DEV)
    L2, _ = det2(x)
    want = amd.LAF.LAFs2ellT(L2).cpu().numpy()
    assert lines[0].strip() == "1.0" and int(lines[1]) == ell.shape[0] == want.shape[0] and np.load(str(out2) + ".desc.npy").shape == (ell.shape[0], 128)
    np.testing.assert_allclose(ell, want, rtol=1e-6, atol=1e-9)
    record_parity("threshold-mode script variants (extract_geomOriTh.py, extract_geom_and_desc_upisupTh.py) on graf img1", rows_ori_th=int(L.shape[0]), rows_upisup_th=int(ell.shape[0]))


@pytest.mark.parametrize("arith", ARITH)
def test_c_host_program_drives_the_boundary_without_python(amd, nets, weights, golden_dir, tmp_path, arith):
    """examples/c_host/extract.c: a plain C99 program (gcc; hipMalloc, affnet_config_fill, flat weight files -> affnet_cnn32_pack_weights,
    affnet_extract_features, affnet_lafs_to_ellipses, affnet_read_counts, Oxford text) - no Python, no torch in that process.  Its LAFs,
    responses, ids and descriptors on graf img1 must be BYTE-equal to the Python mirror's, its text file equal to the mirror's device ellipses."""
    import subprocess
    from affnet_amd import _lib, engine
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "c_host", "extract")
    if not os.path.isfile(exe) or os.path.getmtime(exe) < os.path.getmtime(exe + ".c"):      # normally built by __graft_entry__.build(); gcc is on the GPU box too
        subprocess.check_call(["bash", os.path.join(root, "examples", "c_host", "build.sh")])
    assert os.path.isfile(exe), "examples/c_host/extract did not build"
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libaffnet_hip.so" in ldd and "python" not in ldd.lower() and "torch" not in ldd.lower(), ldd
    x = load_gray(os.path.join(golden_dir, "graf_img1.png"))
    x[0, 0].numpy().tofile(str(tmp_path / "img.f32"))
    for kind, name in ((_lib.NET_AFFNET, "AffNet"), (_lib.NET_ORINET, "OriNet"), (_lib.NET_HARDNET, "HardNet")):
        engine.save_flat_weights(kind, weights[name], str(tmp_path / (name + ".afnw")))
    prefix = str(tmp_path / "out")
    code = _lib.arith_code(arith)
    run = subprocess.run([exe, str(tmp_path / "img.f32"), str(x.size(2)), str(x.size(3)), "2000", str(tmp_path / "AffNet.afnw"),
                          str(tmp_path / "OriNet.afnw"), str(tmp_path / "HardNet.afnw"), prefix, str(code)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr
    print(run.stdout.strip())
    A, O, H = nets
    det = amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=2000, border=5, num_Baum_iters=1, AffNet=A, OriNet=O, arith=arith).to(DEV)
    res = det.run(x.to(DEV), do_ori=True, desc=H)
    n = res["LAFs"].shape[0]
    got = {"LAFs": np.fromfile(prefix + ".lafs.f32", dtype=np.float32).reshape(-1, 2, 3), "responses": np.fromfile(prefix + ".resp.f32", dtype=np.float32),
           "ids": np.fromfile(prefix + ".ids.i32", dtype=np.int32).reshape(-1, 3), "descriptors": np.fromfile(prefix + ".desc.f32", dtype=np.float32).reshape(-1, 128)}
    for k, v in got.items():
        assert v.shape[0] == n and np.array_equal(v, res[k].cpu().numpy()), "the C host program's %s differ from the Python mirror's" % k
    lines = open(prefix + ".txt").read().split("\n")
    assert lines[0] == "1.0" and int(lines[1]) == n
    ell = np.loadtxt(prefix + ".txt", skiprows=2)
    want = amd.LAF.LAFs2ellT(res["LAFs"]).cpu().numpy().astype(np.float64)
    assert ell.shape == (n, 5) and np.abs(ell - want).max() <= 1e-9 * max(1.0, np.abs(want).max()) + 5.1e-11     # '%10.10f' rounding
    record_parity("C host program (examples/c_host/extract.c) vs the Python mirror, graf img1 2000 kp" + ("" if arith == "fp32" else " [arith %s]" % arith),
                  rows=int(n), byte_equal=True)
    # wrong argv -> usage + exit code 1 (hesaffnet.py:21-23)
    assert subprocess.run([exe], capture_output=True).returncode == 1


@pytest.mark.parametrize("arith", ARITH)
def test_config5_4k_deep_pyramid(amd, nets, weights, golden_dir, arith):
    """BASELINE.json configs[4]: 3840x2160, 8000 kp, 8 octaves.  Detector identities must equal the oracle's; LAFs and DESCRIPTORS
    within 1e-3; the batched path (bench.py --config5: 8 images per launch) bit-identical to single-image calls."""
    A, O, H = nets
    x = orc.synthetic_image(2160, 3840, 0)
    det = amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=8000, border=5, num_Baum_iters=1, AffNet=A, OriNet=O, arith=arith).to(DEV)
    res = det.run(x.to(DEV), do_ori=True, desc=H)
    assert len(det.scale_pyr) == 8 and res["LAFs"].shape == (8000, 2, 3) and res["descriptors"].shape == (8000, 128)
    ex, Lw, rw, Pw, Dw = _oracle_describe("synth 4K seed 0", x, 8000, weights)
    gi, wi, dl, dd, rec = _row_stats("configs[4]: 3840x2160 seed 0, 8000 kp" + ("" if arith == "fp32" else " [arith %s]" % arith), res["ids"].cpu().numpy(),
                                     res["LAFs"].cpu().numpy(), res["descriptors"].cpu().numpy(), res["responses"].cpu().numpy(), ex.keys.numpy(), Lw.numpy(),
                                     Dw.numpy(), rw.numpy(), ori_vec=ex.ori_vec.numpy(), ex=ex, hw=(2160, 3840), n_out=8000)
    assert rec["match_rate"] >= 0.995 and (dl < 1e-3 + 1e-6 * 3840).mean() >= 0.995, rec
    _assert_accounted(rec)
    assert rec["responses_equal"]
    assert rec["desc_rows_within_1e-3"] >= 0.995 and dd[dl < 1e-3].max() < 1e-3, rec
    # host-independent leg (round 6): the UNMODIFIED reference's own output for this image on the authoring host (tests/golden/make_golden_config3.py 4k;
    # descriptors stored as float16), rows matched through the response bit pattern.  At 4K a handful of frames of several hundred px with OriNet vectors of
    # length 1e-2 are ill-conditioned beyond 1e-3 px in ANY fp32 evaluation: >= 99.9 % of the rows within 1e-3 px, none outside 1e-2.
    g = np.load(os.path.join(golden_dir, "synth_2160x3840_s0_n8000.npz"))
    Lg, rg, Dg = res["LAFs"].cpu().numpy(), res["responses"].cpu().numpy(), res["descriptors"].cpu().numpy()
    g2, w2 = match_rows(rg, Lg, g["resp"], g["LAFs"])
    eg = np.abs(Lg[g2] - g["LAFs"][w2]).reshape(len(g2), -1).max(axis=1)
    dg = np.abs(Dg[g2] - g["desc"][w2].astype(np.float32)).max(axis=1)
    record_parity("configs[4]: 3840x2160 seed 0 vs the reference's golden output" + ("" if arith == "fp32" else " [arith %s]" % arith), golden_rows=int(len(g["resp"])), matched=int(len(g2)),
                  rows_outside_1e_3=int((eg >= 1e-3).sum()), laf_max_px=float(eg.max()), desc_max=float(dg.max()), desc_rows_outside_1e_3=int((dg >= 1e-3).sum()))
    assert len(g2) >= 0.995 * 8000 and (eg < 1e-3).mean() >= 0.999 and eg.max() < 1e-2, (len(g2), int((eg >= 1e-3).sum()), eg.max())
    assert (dg[eg < 1e-3] < 1e-3).all(), "descriptor of a geometrically matching row off by more than 1e-3 vs the golden output"
    # batched: 8 images per launch (seed 0 first and last so that one oracle run covers both positions)
    del det
    torch.cuda.empty_cache()
    xb = torch.cat([x] + [orc.synthetic_image(2160, 3840, s) for s in range(1, 7)] + [x], 0).to(DEV)
    det8 = amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=8000, border=5, num_Baum_iters=1, AffNet=A, OriNet=O, arith=arith).to(DEV)
    batched = det8.run_batch(xb, do_ori=True, desc=H)
    assert len(batched) == 8
    for i in (0, 7):
        for k in ("LAFs", "responses", "descriptors", "ids"):
            assert torch.equal(batched[i][k], res[k]), "4K batch image %d differs from the single-image call in %s" % (i, k)
    for b in batched:
        assert b["LAFs"].shape == (8000, 2, 3) and np.abs(np.linalg.norm(b["descriptors"].cpu().numpy(), axis=1) - 1.0).max() < 1e-4


def test_many_exact_ties_stay_cheap_and_deterministic(amd):
    """with far more ties at the top-k threshold than needed, the tie break scanned the whole candidate list once per
    tied row (O(n * ties)).  A 16 x 16 block tiled 20 x 15 times gives groups of ~300 identical responses: the cut is found by a radix
    select over the order keys instead - the selected responses equal the oracle's as a multiset, ties are taken in (octave, level,
    pixel) order, repeated runs agree, and the call stays in the millisecond range."""
    import time
    g = torch.Generator().manual_seed(5)
    blk = torch.rand(1, 1, 16, 16, generator=g) * 255.0
    x = blk.repeat(1, 1, 15, 20).contiguous()                             # 240 x 320
    for n in (50, 500):
        det = amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=n, border=5, num_Baum_iters=0).to(DEV)
        det.raw_div = 1          # the small octaves of this image are period-2 patterns: every second pixel is a maximum (default list capacity: h*w/4)
        L, r = det(x.to(DEV))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        L2, r2 = det(x.to(DEV))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert torch.equal(L, L2) and torch.equal(r, r2)
        ex = orc.OracleExtractor(mrSize=5.192, num_features=n, border=5, num_Baum_iters=0)
        Lw, rw = ex(x)
        got, want = np.sort(r.cpu().numpy()), np.sort(rw.numpy())
        assert got.shape == want.shape and np.array_equal(got, want), "selected responses differ from the oracle's as a multiset"
        u, c = np.unique(want, return_counts=True)
        keys = _keys(det.last_ids.cpu().numpy())
        last = r.cpu().numpy() == r.cpu().numpy().min()
        assert np.all(np.diff(keys[last]) > 0) or last.sum() == 1, "ties at the cut must be the first ones in key order"
        record_parity("many exact ties (16x16 block tiled, %d kp)" % n, largest_tie_group=int(c.max()), call_ms=dt * 1e3)
        assert c.max() >= 8 and dt < 0.5, (int(c.max()), dt)       # groups of equal responses inside the selection; the cut falls inside one


SPLIT_MODES = [("fp32_split3", 1), ("fp32_split2h", 2)]


@pytest.mark.parametrize("mode,code", SPLIT_MODES)
def test_split_modes_vs_the_exact_path_and_back(amd, nets, mode, code):
    """AFFNET_ARITH_FP32_SPLIT3 / AFFNET_ARITH_FP32_SPLIT2H (include/affnet_hip.h; the extractor's `arith` kwarg): the contractions of conv1 ..
    conv5 of the three trunks and of the HardNet head as six bf16 MFMAs (three bf16 terms per operand) or three fp16 MFMAs (two fp16 terms) per
    fp32 product, fp32 accumulate.  Distance of the whole path to the exact-fp32 path on one image, and the switch itself: the same extractor
    object, flipped to the split mode and back, returns the exact path's bits again."""
    A, O, H = nets
    x = orc.synthetic_image(240, 320, 1).to(DEV)
    det = amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=300, border=5, num_Baum_iters=1, AffNet=A, OriNet=O).to(DEV)
    exact = det.run(x, do_ori=True, desc=H)
    ctx0 = det._ctx
    det.arith = mode
    split = det.run(x, do_ori=True, desc=H)
    assert det._ctx is ctx0 and det._ctx.arith == code, "switching the arithmetic must not rebuild the context"
    det.arith = "fp32"
    again = det.run(x, do_ori=True, desc=H)
    for k in ("LAFs", "responses", "descriptors", "ids"):
        assert torch.equal(exact[k], again[k]), "switching back to fp32 must restore the exact path bit for bit (%s)" % k
    gi, wi = _match(split["ids"].cpu().numpy(), exact["ids"].cpu().numpy())
    dl = float((split["LAFs"][gi] - exact["LAFs"][wi]).abs().max())
    dd = float((split["descriptors"][gi] - exact["descriptors"][wi]).abs().max())
    record_parity("arith %s vs the exact-fp32 path, 320x240, 300 kp" % mode, matched=int(len(gi)), rows=int(exact["LAFs"].shape[0]), laf_max_px=dl, desc_max=dd)
    print("%s vs exact fp32 path: %d / %d rows, LAF max %.3g px, descriptor max %.3g" % (mode, len(gi), exact["LAFs"].shape[0], dl, dd))
    assert len(gi) >= 299 and dd > 0.0 and dd < 1e-4 and dl < 1e-3
    with pytest.raises(ValueError):
        amd.ScaleSpaceAffinePatchExtractor(arith="bf16")


@pytest.mark.parametrize("mode,code", SPLIT_MODES)
def test_split_trunks_vs_exact_trunks(amd, nets, mode, code):
    """Each net alone on random patches (incl. a ragged count), split arithmetic vs the exact fp32 MFMA path: the two differ like one fp32
    summation order from another.  Catches layout slips of the pre-split / half-patch layers (halo rows, the conv1 row kept for the second
    conv2 pass) and of the split head GEMM that a loose end-to-end bar could hide."""
    A, O, H = nets
    g = torch.Generator().manual_seed(7)
    try:
        for n in (1, 17, 257, 3000):
            p = (torch.rand(n, 1, 32, 32, generator=g) * 255).to(DEV)
            p[0, 0, :16] = 0.0                                            # a half-constant patch: exact zeros through ReLU in one half
            exact = [net(p).clone() for net in (A, O, H)]
            for net in (A, O, H):
                net.arith = mode
            split = [net(p).clone() for net in (A, O, H)]
            for net in (A, O, H):
                net.arith = "fp32"
            again = [net(p) for net in (A, O, H)]
            # OriNet returns the rotation of atan2(o): a short output vector o amplifies a 1e-7 difference (same effect as in the parity report)
            for nm, e, s_, a2, bar in zip(("AffNet", "OriNet", "HardNet"), exact, split, again, (2e-5, 5e-3, 1e-5)):
                d = float((e - s_).abs().max())
                record_parity("arith %s vs exact: %s alone, %d random patches" % (mode, nm, n), max_abs=d)
                assert torch.equal(e, a2), "switching the arithmetic back must restore the exact path bit for bit"
                assert 0.0 < d < bar or (n == 1 and d < bar), (nm, n, d)
    finally:
        for net in (A, O, H):
            net.arith = "fp32"


def test_split2h_activation_range_is_loud_not_silent(amd):
    """include/affnet_hip.h, AFFNET_ARITH_FP32_SPLIT2H: activations are carried as fp16 pairs, so |a| < 65504 is required.  The reference's nets are
    BatchNorm-ed (|a| < 50 on every test input); a net whose activations leave the range must NOT return plausible numbers: with conv1's weights times 1e6
    (activations ~1e6 from conv1 on) the exact and the three-bf16-term modes still return finite unit descriptors, the two-fp16-term mode returns non-finite
    values for every patch - visible to any caller - and is unaffected once the weights are back."""
    sd = amd.synthetic_hardnet_state(0)
    big = {k: v.clone() for k, v in sd.items()}
    big["features.3.weight"] *= 1.0e6
    p = (torch.rand(64, 1, 32, 32, generator=torch.Generator().manual_seed(3)) * 255).to(DEV)
    H = amd.HardNet(); H.load_state_dict(big); H.to(DEV)
    try:
        for mode, finite in (("fp32", True), ("fp32_split3", True), ("fp32_split2h", False)):
            H.arith = mode
            d = H(p)
            ok = torch.isfinite(d).all(dim=1)
            if finite:
                assert bool(ok.all()) and float((d.norm(dim=1) - 1).abs().max()) < 1e-4, mode
            else:
                assert not bool(ok.any()), "out-of-range activations must surface as non-finite descriptors in arith %s (%d of %d rows finite)" % (mode, int(ok.sum()), ok.numel())
    finally:
        H.arith = "fp32"
    H2 = amd.HardNet(); H2.load_state_dict(sd); H2.to(DEV)
    H2.arith = "fp32_split2h"
    try:
        assert bool(torch.isfinite(H2(p)).all())
    finally:
        H2.arith = "fp32"


@pytest.mark.parametrize("arith", ARITH)
def test_cnn_raw_outputs_vs_the_reference_jit_traces(amd, nets, weights, golden_dir, arith):
    """Second CNN oracle: the reference's own TorchScript traces convertJIT/AffNetJIT.pt / OriNetJIT.pt return the RAW
    network outputs - AffNet's (1 + x0, x1, 1 + x2) before the rectification, OriNet's (y, x) before atan2 (the quantity the parity outliers
    hinge on).  tests/golden/cnn_jit_raw.npz holds the traces' unmodified outputs (tests/golden/make_golden_jit.py).  The HIP trunks' raw
    values are taken from the per-wave head partials the trunk kernel leaves in the caller's scratch buffer (affnet_cnn32_forward: AffNet
    [patch][wave 8][4], OriNet [patch][wave 8][2 outputs x 9 taps]) exactly as the finish kernels combine them."""
    from affnet_amd import engine, _lib
    A, O, H = nets
    g, j = np.load(os.path.join(golden_dir, "cnn_random_patches.npz")), np.load(os.path.join(golden_dir, "cnn_jit_raw.npz"))
    p = torch.from_numpy(g["patches"]).to(DEV)
    n = p.size(0)
    # AffNet
    scr = torch.zeros(n * 144, device=DEV)
    out = engine.cnn_forward(_lib.NET_AFFNET, A.packed_weights(torch.device(DEV)), p, scratch=scr, arith=arith)
    part = scr[: n * 32].view(n, 8, 4).double().cpu().numpy()
    hb = weights["AffNet"]["features.19.bias"].double().numpy()
    raw = np.tanh(part.sum(axis=1)[:, :3] + hb) + np.array([1.0, 0.0, 1.0])
    da = _report("AffNet raw (1 + x0, x1, 1 + x2) vs AffNetJIT.pt [arith %s]" % arith, raw, j["affnet_raw"])
    assert da.max() < 2e-5
    want = orc.rectify_up_is_up(torch.from_numpy(np.stack([np.stack([j["affnet_raw"][:, 0], 0 * j["affnet_raw"][:, 0]], 1),
                                                           np.stack([j["affnet_raw"][:, 1], j["affnet_raw"][:, 2]], 1)], 1)).float()).numpy()
    assert np.abs(out.cpu().numpy() - want).max() < 2e-5, "rectified HIP output vs rectify(AffNetJIT)"
    # OriNet
    scr = torch.zeros(n * 144, device=DEV)
    out = engine.cnn_forward(_lib.NET_ORINET, O.packed_weights(torch.device(DEV)), p, scratch=scr, arith=arith)
    part = scr.view(n, 8, 2, 9).double().cpu().numpy()
    hb = weights["OriNet"]["features.19.bias"].double().numpy()
    raw = np.tanh(part.sum(axis=1) + hb[None, :, None]).mean(axis=2)                     # (n, 2) = (y, x)
    do = _report("OriNet raw (y, x) vs OriNetJIT.pt [arith %s]" % arith, raw, j["orinet_raw"])
    assert do.max() < 2e-5
    ang = np.arctan2(j["orinet_raw"][:, 0] + 1e-8, j["orinet_raw"][:, 1] + 1e-8)
    got = out.cpu().numpy()
    short = np.linalg.norm(j["orinet_raw"], axis=1)
    assert np.all(np.abs(got[:, 0, 0] - np.cos(ang)) < 2e-5 + 5e-6 / short) and np.all(np.abs(got[:, 0, 1] - np.sin(ang)) < 2e-5 + 5e-6 / short)


@pytest.mark.parametrize("ranks,gather", [(2, "all"), (3, "rank0")])
def test_bench_n_rank_gather_with_real_kernels(ranks, gather):
    """The N-rank path of bench.py with REAL kernels: N self-spawned ranks share this one device (gloo for the
    exchange - RCCL cannot put two ranks on one GPU), every rank computes its images, the records travel through all_gather / the
    gather to rank 0, and rank 0 re-computes EVERY record of the last step with a single-image call on the same seed
    (--verify-gather all): counts, LAFs, responses and descriptors bit-equal, global order image i -> rank i % world."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"AFFNET_BENCH_ONE_DEVICE": "1", "AFFNET_BENCH_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(ranks), "--steps", "2", "--warmup", "1", "--batch", "4", "--chunk", "2",
                        "--no-secondary", "--gather", gather, "--verify-gather", "all"], env=env, capture_output=True, text=True, timeout=900)
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, p.stderr[-2000:]
    d = lines[0]
    gc = d["gather_check"]
    record_parity("bench.py --gpus %d (one device, gloo, %s): gathered records vs single-image recomputation on rank 0" % (ranks, gather), **gc)
    assert d["n_gpus"] == ranks and gc["records"] == 4 * ranks and gc["checked"] == 4 * ranks and gc["identical"] is True, gc
    assert len(p.stdout.rstrip("\n").splitlines()[-1]) < 4096, "the N-rank line must be compact too"
    assert d["exchange"]["bytes_per_step"] == (4 + 540 * 2000) * 4 * ranks * (ranks if gather == "all" else 1)
    assert d["exchange"]["mode"] == ("all_gather" if gather == "all" else "gather_rank0") and d["exchange"]["gather_ms"] > 0
    assert set(d["ms_per_step_per_rank"]) == {"min", "max"} and d["ms_per_step_per_rank"]["max"] >= d["ms_per_step_per_rank"]["min"] > 0
    full = json.load(open(os.path.join(root, d["detail"])))                      # the full record (rank 0 writes it next to bench.py)
    assert len(full["ms_per_step_per_rank"]["all"]) == ranks and full["gather_check"]["identical"] is True


def test_bench_line_is_the_last_stdout_line_with_rccl():
    """The driver parses the LAST stdout line.  RCCL writes a version banner ("RCCL version : ... Librccl path : ...") to the C stdout, which - buffered -
    landed AFTER the JSON line when stdout is a file (round 6 evidence run).  bench.py destroys the process group first, flushes the C buffers and prints its
    line last; ranks other than 0 send their stdout to stderr.  The 1-rank RCCL flavour of the N-rank path (AFFNET_BENCH_SELF_GATHER=1) on this box."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"AFFNET_BENCH_SELF_GATHER": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "1", "--batch", "4", "--chunk", "2", "--no-cpu-baseline",
                        "--no-secondary", "--no-other-configs", "--no-split3", "--verify-gather", "all"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    last = p.stdout.rstrip("\n").splitlines()[-1]
    d = json.loads(last)                      # fails on a banner line
    assert len(last) < 4096 and d["n_gpus"] == 1 and d["value"] > 0 and d["exchange"]["mode"] == "gather_rank0" and d["gather_check"]["identical"] is True
