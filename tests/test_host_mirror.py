"""CPU: the C-ABI library loads and exports every declared symbol; the host-side mirror
(tap tables, pyramid plan, weight packing, affine_grid base coordinates) agrees with the oracle."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import affnet_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from affnet_amd import _lib
    raw = C.CDLL(_lib.LIB_PATH)
    # the boundary header and the debug / tuning header (kept apart: only the first is the drop-in ABI)
    for fname, table in (("affnet_hip.h", _lib.SYMBOLS), ("affnet_hip_debug.h", _lib.DEBUG_SYMBOLS)):
        hdr = open(os.path.join(ROOT, "include", fname)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        declared = set(re.findall(r"\b(affnet_[a-z0-9_]+)\s*\(", hdr))
        assert declared, "no declarations parsed in " + fname
        assert declared == set(table), (fname, declared ^ set(table))
        for name in declared:
            assert hasattr(raw, name), name
    assert not [n for n in _lib.SYMBOLS if "debug" in n or "probe" in n or "selftest" in n], "debug entry points leaked into the boundary"
    # the probe kernels of the tuning tools (include/affnet_hip_probes.h) are NOT in the shipped library - only in libaffnet_hip_probes.so
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "affnet_hip_probes.h")).read(), flags=re.S)
    probes = set(re.findall(r"\b(affnet_[a-z0-9_]+)\s*\(", hdr))
    assert probes == set(_lib.PROBE_SYMBOLS)
    product = C.CDLL(os.path.join(ROOT, "affnet_amd", "libaffnet_hip.so"))
    assert not [n for n in probes if hasattr(product, n)], "probe entry points in the product library"
    if os.path.isfile(_lib.PROBES_LIB_PATH):
        pl = C.CDLL(_lib.PROBES_LIB_PATH)
        for name in list(probes) + list(_lib.SYMBOLS) + list(_lib.DEBUG_SYMBOLS):
            assert hasattr(pl, name), name
    assert b"gfx950" in _lib.lib.affnet_version()
    assert C.sizeof(_lib.Config) > 30000  # struct mirrors the 8 x 31x31 tap tables


def test_base_grid_matches_torch_affine_grid():
    from affnet_amd import engine
    for ps in list(range(2, 65)):
        ref = (torch.linspace(-1, 1, ps) * (ps - 1) / ps).numpy()
        assert np.array_equal(engine.base_grid(ps), ref), ps
    theta = torch.tensor([[[1.0, 0, 0], [0, 1.0, 0]]])
    g = F.affine_grid(theta, torch.Size((1, 1, 32, 32)), align_corners=False)
    assert np.array_equal(g[0, 0, :, 0].numpy(), engine.base_grid(32))


@pytest.mark.parametrize("hw", [(768, 1024), (640, 800), (2160, 3840), (240, 320), (65, 97)])
def test_pyramid_plan_and_taps_match_oracle(hw):
    from affnet_amd.host_plan import PyramidPlan, gaussian_taps
    plan = PyramidPlan(hw[0], hw[1], 3, 1.6, 5)
    ref = orc.pyramid_plan(hw[0], hw[1], 3, 1.6, 5)
    assert plan.sizes == [(o["h"], o["w"]) for o in ref["octaves"]]
    assert plan.sigmas == [o["level_sigmas"] for o in ref["octaves"]]
    assert plan.blur_sigmas == ref["octaves"][0]["blur_sigmas"]
    for s in [plan.first_blur_sigma] + plan.blur_sigmas:
        assert np.array_equal(gaussian_taps(s), orc.gauss_kernel_2d(s)[0])
    cfg = plan.fill_config(5.192, 0.0, 2000, 3000)
    assert cfg.n_octaves == len(ref["octaves"]) and cfg.level_blur_taps[1] == 9 and cfg.level_blur_taps[4] == 15
    assert cfg.first_blur_taps == 11
    assert np.float32(cfg.level_sigma4[0][2]) == np.float32(plan.sigmas[0][2] ** 4)


def test_c_side_config_fill_equals_the_python_plan_byte_for_byte():
    """affnet_config_fill (csrc/config_fill.hip) = the C-side way to fill affnet_config (HandCraftedModules.py:14-56, Utils.py:92-114,155-161
    restated with doubles, numpy's linspace and pairwise np.sum): the struct must be byte-identical to the one the Python mirror builds from
    the reference's numpy formulas (affnet_amd/host_plan.py) - octave sizes, sigmas, sigma^4, every float32 tap of every Gaussian."""
    from affnet_amd import _lib
    from affnet_amd.host_plan import PyramidPlan
    rng = np.random.RandomState(7)
    cases = [(h, w, nl, s, b) for (h, w) in [(768, 1024), (640, 800), (2160, 3840), (240, 320), (481, 641), (598, 1000), (33, 47), (1000, 563)]
             for nl in (1, 2, 3, 4, 5, 6) for s in (1.6, 0.4, 0.5, 1.0, 2.2) for b in (5, 15, 0)]
    cases += [(768, 1024, 3, float(s), 5) for s in rng.uniform(0.3, 3.0, size=60)]
    refused = 0
    for h, w, nl, s, b in cases:
        try:
            want = PyramidPlan(h, w, nl, s, b).fill_config(5.192, 0.25, 2000, 3000, batch=2, baum_iters=1)
        except ValueError:
            want = None                                  # Gaussian wider than AFFNET_MAX_TAPS / pyramid too deep
        got = _lib.Config()
        rc = _lib.lib.affnet_config_fill(C.byref(got), h, w, nl, s, b, 5.192, 0.25, 2000, 3000, 2, 1)
        if want is None:
            assert rc == _lib.ERR_INVALID, (h, w, nl, s, b)
            refused += 1
            continue
        assert rc == _lib.OK, (h, w, nl, s, b)
        assert bytes(got) == bytes(want), "affnet_config_fill differs from host_plan for %s" % ((h, w, nl, s, b),)
    assert refused > 0, "the sweep must include configurations both sides refuse"
    assert _lib.lib.affnet_config_fill(C.byref(_lib.Config()), 100, 100, 7, 1.6, 5, 3.0, 0.0, 10, 10, 1, 0) == _lib.ERR_INVALID       # n_levels + 2 > AFFNET_MAX_LEVELS
    assert _lib.lib.affnet_config_fill(None, 100, 100, 3, 1.6, 5, 3.0, 0.0, 10, 10, 1, 0) == _lib.ERR_INVALID


def test_context_layout_no_gpu_needed():
    from affnet_amd import _lib
    from affnet_amd.host_plan import PyramidPlan
    cfg = PyramidPlan(768, 1024).fill_config(5.192, 0.0, 2000, 3000)
    h = C.c_void_p()
    assert _lib.lib.affnet_ctx_create(C.byref(h), 0, C.byref(cfg)) == 0
    nbytes = _lib.lib.affnet_workspace_bytes(h)
    assert 20e6 * 4 < nbytes < 400e6
    assert _lib.lib.affnet_capacity_prefilter(h) == 3000 and _lib.lib.affnet_capacity_final(h) == 2000
    o00, o01, o10 = (_lib.lib.affnet_pyramid_level_offset(h, *a) for a in [(0, 0), (0, 1), (1, 0)])
    assert o01 - o00 == 768 * 1024 and o10 - o00 == 5 * 768 * 1024
    assert _lib.lib.affnet_pyramid_level_offset(h, 6, 0) == -1
    # error behaviour: unbound workspace -> ERR_INVALID + message
    rc = _lib.lib.affnet_pyramid_build(h, None, None)
    assert rc == _lib.ERR_INVALID and b"not bound" in _lib.lib.affnet_last_error(h)
    bad = PyramidPlan(768, 1024).fill_config(5.192, 0.0, 2000, 3000)
    bad.oct_h[1] = 100
    h2 = C.c_void_p()
    assert _lib.lib.affnet_ctx_create(C.byref(h2), 0, C.byref(bad)) == _lib.ERR_INVALID
    _lib.lib.affnet_ctx_destroy(h)
    _lib.lib.affnet_ctx_destroy(h2)


def test_arith_mode_is_part_of_the_boundary():
    """include/affnet_hip.h: AFFNET_ARITH_FP32_MFMA (default) / AFFNET_ARITH_FP32_SPLIT3 through affnet_config.arith and affnet_set_arith,
    mirrored by the `arith` kwarg of the extractors and the nets' `.arith` attribute; unknown modes are errors, not silent defaults."""
    import affnet_amd
    from affnet_amd import _lib
    from affnet_amd.host_plan import PyramidPlan
    hdr = open(os.path.join(ROOT, "include", "affnet_hip.h")).read()
    assert re.search(r"#define AFFNET_ARITH_FP32_MFMA 0\b", hdr) and re.search(r"#define AFFNET_ARITH_FP32_SPLIT3 1\b", hdr)
    assert re.search(r"#define AFFNET_ARITH_FP32_SPLIT2H 2\b", hdr)
    assert (_lib.ARITH_FP32_MFMA, _lib.ARITH_FP32_SPLIT3, _lib.ARITH_FP32_SPLIT2H) == (0, 1, 2)
    h = C.c_void_p()
    assert _lib.lib.affnet_ctx_create(C.byref(h), 0, C.byref(PyramidPlan(240, 320).fill_config(5.192, 0.0, 300, 450, arith=2))) == 0
    assert _lib.lib.affnet_get_arith(h) == 2 and _lib.lib.affnet_set_arith(h, 0) == 0 and _lib.lib.affnet_set_arith(h, 2) == 0 and _lib.lib.affnet_get_arith(h) == 2
    _lib.lib.affnet_ctx_destroy(h)
    for cfg_arith in (0, 1):
        cfg = PyramidPlan(240, 320).fill_config(5.192, 0.0, 300, 450, arith=cfg_arith)
        h = C.c_void_p()
        assert _lib.lib.affnet_ctx_create(C.byref(h), 0, C.byref(cfg)) == 0
        assert _lib.lib.affnet_get_arith(h) == cfg_arith
        assert _lib.lib.affnet_set_arith(h, 1 - cfg_arith) == 0 and _lib.lib.affnet_get_arith(h) == 1 - cfg_arith
        assert _lib.lib.affnet_set_arith(h, 7) == _lib.ERR_INVALID and b"arith" in _lib.lib.affnet_last_error(h)
        assert _lib.lib.affnet_get_arith(h) == 1 - cfg_arith
        _lib.lib.affnet_ctx_destroy(h)
    bad = PyramidPlan(240, 320).fill_config(5.192, 0.0, 300, 450)
    bad.arith = 3
    h = C.c_void_p()
    assert _lib.lib.affnet_ctx_create(C.byref(h), 0, C.byref(bad)) == _lib.ERR_INVALID
    _lib.lib.affnet_ctx_destroy(h)
    u = C.c_void_p()                                                      # utility context (cfg == NULL): default exact fp32
    assert _lib.lib.affnet_ctx_create(C.byref(u), 0, None) == 0 and _lib.lib.affnet_get_arith(u) == 0
    _lib.lib.affnet_ctx_destroy(u)
    assert _lib.arith_code("fp32") == 0 and _lib.arith_code("fp32_split3") == 1 and _lib.arith_code("fp32_split2h") == 2 and _lib.arith_code(None) == 0
    with pytest.raises(ValueError):
        _lib.arith_code("bf16")
    with pytest.raises(ValueError):
        affnet_amd.ScaleSpaceAffinePatchExtractor(arith="tf32")
    assert affnet_amd.ScaleSpaceAffinePatchExtractor().arith == "fp32" and affnet_amd.AffNetFast().arith == "fp32"


def test_no_product_kernel_spills_registers():
    """Read from the built objects (tools/kernel_resources.py -> the code objects' metadata notes): no kernel of the library may spill
    VGPRs or use scratch memory.  Round 3 shipped hessian_nms_kernel<5> with 5 spilled registers and 24 B of scratch per thread because
    its launch bound asked for an occupancy its register count missed by five; loop-invariant addresses held across the tile loop were
    the cause."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources as kr
    ks = kr.all_kernels()
    assert len(ks) > 100, "metadata of the built objects not found (run __graft_entry__.build())"
    # (the STAMPS = true instantiations of the trunk kernel - third template argument - exist only for tools/*_phase_timing.py and the
    # layer dumps of the parity tests; the probe kernels of the tuning tools are not in these objects at all since round 5)
    tooling = re.compile(r"cnn32_trunk_kernelILi\dELi8ELb1E")
    assert not [k["name"] for k in ks.values() if re.search(r"probe_kernel|split3_rate|split3_gemm|stream_read_kernel|stream_write_kernel|tile_read_kernel", k["name"])], \
        "probe kernels in the product objects (they belong to libaffnet_hip_probes.so, AFFNET_PROBES=1)"
    bad = {k["name"]: (k.get("vgpr_spill_count", 0), k.get("sgpr_spill_count", 0), k.get("private_segment_fixed_size", 0)) for k in ks.values()
           if (k.get("vgpr_spill_count", 0) or k.get("private_segment_fixed_size", 0)) and not tooling.search(k["name"])}
    assert not bad, bad
    h5 = [k for k in ks.values() if "hessian_nms_kernelILi5E" in k["name"]][0]
    assert h5["group_segment_fixed_size"] == 52520 and kr.workgroups_per_cu(h5) == 3, h5      # three workgroups per CU by LDS AND by registers
    trunk = [k for k in ks.values() if "cnn32_trunk_kernelILi2ELi8ELb0ELi0E" in k["name"]][0]
    assert trunk["group_segment_fixed_size"] <= 160 * 1024 and kr.workgroups_per_cu(trunk) == 1


def test_design_kernel_figures_match_the_built_objects():
    """DESIGN.md section 4 quotes VGPR / LDS / scratch / occupancy figures per kernel; the table between the `kernel-resources` markers is the output
    of tools/kernel_resources.py --design.  Round 3's DESIGN said "no scratch" for a kernel that shipped with 5 spilled registers: the
    document and the built objects must not drift apart again."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources as kr
    doc = open(os.path.join(ROOT, "DESIGN.md")).read()
    m = re.search(r"<!-- kernel-resources:begin -->\n(.*?)\n<!-- kernel-resources:end -->", doc, flags=re.S)
    assert m, "DESIGN.md lost its generated kernel-resources table"
    want = kr.design_table()
    assert "(not built)" not in want, want
    assert m.group(1).strip() == want.strip(), "DESIGN.md section 4 differs from the built objects - regenerate with tools/kernel_resources.py --design:\n" + want


def _unpack_trunk(kind, blob):
    """Rebuilds conv weights / biases from the packed blob (the layout cnn32.hip documents)."""
    cb = 32 if kind == 2 else 16
    ch = [1, cb, cb, 2 * cb, 2 * cb, 4 * cb, 4 * cb]
    off, layers = 0, []
    for i in range(6):
        ci, co = ch[i], ch[i + 1]
        kk = 12 if i == 0 else 9 * ci        # conv0: K = 9 taps zero-padded to 12 (three 16x16x4 MFMA k-steps)
        w = blob[off:off + kk * co]
        off += kk * co
        b = blob[off:off + co]
        off += co
        off = (off + 3) & ~3
        if i == 0:
            assert float(w.view(12, co)[9:].abs().max()) == 0.0
            W = w.view(12, co)[:9].t().reshape(co, 1, 3, 3)
        else:
            # [tap][G = c/16][kq = (c/4)%4][n][j = c%4]  (channel-interleaved by 4: one dwordx4 / ds_read_b128 = 4 k-steps)
            W = w.view(9, ci // 16, 4, co, 4).permute(3, 1, 2, 4, 0).reshape(co, ci, 3, 3)
        layers.append((W.contiguous(), b.clone()))
    return layers, off


@pytest.mark.parametrize("kind,name", [(0, "AffNet"), (1, "OriNet"), (2, "HardNet")])
def test_packed_weights_reproduce_the_network(kind, name, weights):
    """BN folding + packing order: a torch-CPU forward driven ONLY by the packed blob equals the oracle."""
    from affnet_amd import engine
    sd = weights[name]
    blob = engine.pack_state_dict(kind, sd)
    layers, off = _unpack_trunk(kind, blob)
    g = torch.Generator().manual_seed(3)
    p = torch.rand(6, 1, 32, 32, generator=g) * 255
    x = orc.input_norm(p)
    for i, (W, b) in enumerate(layers):
        x = F.relu(F.conv2d(x, W, b, stride=2 if i in (2, 4) else 1, padding=1))
    if kind == 2:
        Bw = blob[off:off + 8192 * 128].view(512, 4, 128, 4).permute(0, 1, 3, 2).reshape(8192, 128)   # [k/16][(k/4)%4][n][k%4] -> [k][n]
        bias = blob[off + 8192 * 128: off + 8192 * 128 + 128]
        y = x.permute(0, 2, 3, 1).reshape(6, -1) @ Bw + bias      # head K order = [pixel][channel] (the trunk kernel's store order)
        got = y / torch.sqrt((y * y).sum(1, keepdim=True) + 1e-8)
        want = orc.hardnet_forward(sd, p)
    elif kind == 0:
        hw = blob[off:off + 3 * 4096].view(3, 4096)                       # [o][pixel][channel]
        hb = blob[off + 3 * 4096: off + 3 * 4096 + 3]
        t = torch.tanh(x.permute(0, 2, 3, 1).reshape(6, -1) @ hw.t() + hb)
        A = torch.zeros(6, 2, 2)
        A[:, 0, 0], A[:, 1, 0], A[:, 1, 1] = 1 + t[:, 0], t[:, 1], 1 + t[:, 2]
        got, want = orc.rectify_up_is_up(A), orc.affnet_forward(sd, p)
    else:
        hw = blob[off:off + 2 * 4096].view(2, 8, 8, 64).permute(0, 3, 1, 2).contiguous()     # [o][ky][kx][c] -> (o, c, ky, kx)
        hb = blob[off + 2 * 4096: off + 2 * 4096 + 2]
        t = torch.tanh(F.conv2d(x, hw, hb, padding=1)).mean(dim=(2, 3))
        got = orc.rotation_matrix(torch.atan2(t[:, 0] + 1e-8, t[:, 1] + 1e-8))
        want = orc.orinet_forward(sd, p)
    assert float((got - want).abs().max()) < 2e-5


def test_reference_checkpoints_load_into_the_mirrors(weights):
    import affnet_amd
    A = affnet_amd.AffNetFast(PS=32)
    A.load_state_dict(weights["AffNet"])
    O = affnet_amd.OriNetFast(PS=32)
    O.load_state_dict(weights["OriNet"])
    Hn = affnet_amd.HardNet()
    Hn.load_state_dict(weights["HardNet"])
    assert A.PS == 32 and O.PS == 32 and sorted(k for k in A.state_dict() if "num_batches" not in k) == sorted(weights["AffNet"])


def test_no_cpu_fallback_and_unbuilt_slots_fail_loudly(weights):
    import affnet_amd
    A = affnet_amd.AffNetFast(PS=32)
    with pytest.raises(RuntimeError, match="no CPU path"):
        A(torch.zeros(2, 1, 32, 32))
    det = affnet_amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=100, border=5, num_Baum_iters=1, AffNet=A)
    with pytest.raises(RuntimeError, match="no CPU path"):
        det(torch.zeros(1, 1, 64, 64))
    dflt = affnet_amd.ScaleSpaceAffinePatchExtractor(num_Baum_iters=1)    # default slots as in SparseImgRepresenter.py:42-49
    assert type(dflt.AffNet).__name__ == "AffineShapeEstimator" and type(dflt.OriNet).__name__ == "OrientationDetector"
    assert dflt.AffNet.PS == 19 and dflt.OriNet.PS == 19
    with pytest.raises(NotImplementedError):
        affnet_amd.HandCraftedModules.OrientationDetector()                   # reference default PS = 32: kernels are for 19
    assert affnet_amd.ScaleSpaceAffinePatchExtractor(RespNet=lambda x, s: x).RespNet is not None      # slot accepted (GPU test runs it)
    assert affnet_amd.ScaleSpaceAffinePatchExtractor(nlevels=4).nlevels == 4                           # 1..6 levels supported
    with pytest.raises(NotImplementedError):
        affnet_amd.ScaleSpaceAffinePatchExtractor(nlevels=7)
    d = affnet_amd.ScaleSpaceAffinePatchExtractor(th=28.41)
    assert d.num == -1                                                        # SparseImgRepresenter.py:33-35
    with pytest.raises(RuntimeError, match="forward"):
        d.extract_patches_from_pyr(torch.zeros(1, 2, 3))


def test_handcrafted_windows_match_the_reference_formula():
    """The Gaussian windows of the hand-crafted slots are host tables: they must equal the oracle's CircularGaussKernel."""
    from affnet_amd.HandCraftedModules import OrientationDetector, AffineShapeEstimator
    w = np.array(list(OrientationDetector(patch_size=19).window()), dtype=np.float32)
    assert np.array_equal(w, (10.0 * orc.circular_gauss_kernel(kernlen=19)).astype(np.float32).reshape(-1))
    w = np.array(list(AffineShapeEstimator(patch_size=19).window()), dtype=np.float32)
    assert np.array_equal(w, orc.circular_gauss_kernel(kernlen=19, sigma=(19 / 2) / 3.0).astype(np.float32).reshape(-1))


def test_exact_arithmetic_spec_of_the_blur():
    """Documents WHY the HIP blur is bit-exact: torch's CPU conv2d == row-major sequential fmaf chain."""
    x = orc.synthetic_image(40, 56, 5)
    ker, pad = orc.gauss_kernel_2d(1.2262734984654078)
    ref = orc.gaussian_blur(x, 1.2262734984654078)[0, 0].numpy()
    xp = np.pad(x[0, 0].numpy(), pad, mode="edge")
    acc = np.zeros_like(ref)
    k = ker.shape[0]
    for i in range(k):
        for j in range(k):
            acc = (xp[i:i + 40, j:j + 56].astype(np.float64) * np.float64(ker[i, j]) + acc.astype(np.float64)).astype(np.float32)
    assert np.array_equal(acc, ref)


def test_headers_are_plain_c():
    """The drop-in boundary is a C ABI: every header must compile as C99 without warnings (no C++ constructs, no torch / HIP types); and the plain-C
    host program of examples/c_host builds against them with gcc -Wall -Wextra (examples/c_host/build.sh, run by __graft_entry__.build())."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    for h in ("affnet_hip.h", "affnet_hip_debug.h", "affnet_hip_probes.h"):
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", h)])
        src = open(os.path.join(ROOT, "include", h)).read()
        assert "hip/" not in src and "torch" not in re.sub(r"/\*.*?\*/", "", src, flags=re.S), h


def test_bench_finds_the_counter_traffic_of_its_dominant_kernel():
    """bench.py fills roofline.traffic from the committed rocprofv3 --pmc passes; the trunk kernel's name gained template arguments over the
    rounds, and a lookup under an old name silently produced `traffic: null` once (round 3).  Both instantiations must be found in profiles/."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    argv = sys.argv
    try:
        sys.argv = ["bench.py"]
        spec.loader.exec_module(bench)
    finally:
        sys.argv = argv
    exact, note = bench.pmc_traffic(32)
    assert exact and 2.0e9 < exact < 6.0e9 and "traffic.json" in note, (exact, note)        # ~3.8 GB per 32-image HardNet trunk launch
    split, note3 = bench.pmc_traffic(32, split=True)
    assert split and 2.0e9 < split < 8.0e9 and "traffic.json" in note3, (split, note3)
    half, _ = bench.pmc_traffic(16)
    assert abs(half * 2 - exact) < 1e-3 * exact


@pytest.mark.parametrize("kind,name", [(0, "AffNet"), (1, "OriNet"), (2, "HardNet")])
def test_split_weight_copies_are_exact(kind, name, weights):
    """AFFNET_ARITH_FP32_SPLIT3 (include/affnet_hip.h): the packed blob carries conv1 .. conv5 once more as three bf16 terms in the
    bf16 MFMA's fragment order.  The three terms of a weight must add up to the BN-folded fp32 weight of the exact path EXACTLY (24-bit
    significand = 3 x 8 bits, exact remainders) - the split path's only error is then the 2^-25 of the six-product truncation - and sit
    where the kernels read them: [tap][cin/32][term][kq][cout][8] (cin >= 32), [step][term][kq = 2 (tap & 1) + c / 8][cout][8] (cin = 16)."""
    from affnet_amd import engine
    blob = engine.pack_state_dict(kind, weights[name])
    layers, off = _unpack_trunk(kind, blob)
    off += {0: 3 * 4096 + 4, 1: 2 * 4096 + 4, 2: 8192 * 128 + 128}[kind]          # behind the head weights and bias
    bits = blob.numpy().view(np.uint16)
    for i in range(1, 6):
        W = layers[i][0].numpy().astype(np.float64)                                   # (co, ci, 3, 3): what the fp32 path multiplies with
        co, ci = W.shape[:2]
        n_fl = (5 if ci == 16 else 9 * (ci // 32)) * 3 * 4 * co * 4
        raw = bits[2 * off: 2 * (off + n_fl)].astype(np.uint32) << 16
        terms = raw.view(np.float32).astype(np.float64)
        if ci == 16:
            t = terms.reshape(5, 3, 2, 2, co, 8).transpose(1, 4, 3, 5, 0, 2).reshape(3, co, 16, 10)     # (term, n, c = 8 cg + j, tap = 2 step + half)
            assert np.all(t[..., 9] == 0.0), "the pad half-step must multiply zeros"
            t = t[..., :9]
        else:
            t = terms.reshape(9, ci // 32, 3, 4, co, 8).transpose(2, 4, 1, 3, 5, 0).reshape(3, co, ci, 9)
        rec = t.sum(axis=0).reshape(co, ci, 3, 3)
        assert np.array_equal(rec, W), (name, i, float(np.abs(rec - W).max()))
        mag = np.abs(t)                                                               # term k is at most 2^-8 of term k - 1 (nearest-even split)
        assert np.all(mag[1] <= mag[0] * 2.0 ** -8 + 1e-45) and np.all(mag[2] <= mag[1] * 2.0 ** -8 + 1e-45)
        off += n_fl
    if kind == 2:
        # HardNet head (8192 x 128, BN folded): split copy [k / 32][term][kq = (k % 32) / 8][n][k % 8] against the fp32 copy the exact path
        # multiplies with ([k / 16][(k / 4) % 4][n][k % 4]); k = pixel * 128 + channel in both
        K = 8192
        head_off = off - sum((5 if layers[i][0].shape[1] == 16 else 9 * (layers[i][0].shape[1] // 32)) * 3 * 4 * layers[i][0].shape[0] * 4 for i in range(1, 6)) \
            - (K * 128 + 128)
        Wf = blob.numpy()[head_off: head_off + K * 128].reshape(K // 16, 4, 128, 4).transpose(0, 1, 3, 2).reshape(K, 128).astype(np.float64)
        n_fl = K * 128 * 3 // 2
        raw = bits[2 * off: 2 * (off + n_fl)].astype(np.uint32) << 16
        t = raw.view(np.float32).astype(np.float64).reshape(K // 32, 3, 4, 128, 8).transpose(1, 0, 2, 4, 3).reshape(3, K, 128)
        assert np.array_equal(t.sum(axis=0), Wf), float(np.abs(t.sum(axis=0) - Wf).max())
        mag = np.abs(t)
        assert np.all(mag[1] <= mag[0] * 2.0 ** -8 + 1e-45) and np.all(mag[2] <= mag[1] * 2.0 ** -8 + 1e-45)
        off += n_fl
    # AFFNET_ARITH_FP32_SPLIT2H: the same layers once more as TWO fp16 terms of 2^e w, e per layer such that the largest |w| lands in [2^13, 2^14);
    # same fragment orders with two terms, then 4 floats whose first is 2^-e.  hi + lo must reproduce 2^e w to 2^-23 relative (23 of fp32's
    # 24 significand bits), hi must be the nearest fp16 of 2^e w, lo at most half an ulp of hi (a remainder below 2^-14 is a subnormal fp16: absolute error 2^-25, which the scale
    # makes 2^-38 of the layer's largest weight).
    halves = blob.numpy().view(np.float16)

    def check_h2(W, t, inv_scale, what):
        scale = 1.0 / float(inv_scale)
        assert scale == 2.0 ** round(np.log2(scale)), (what, scale)
        Ws = W * scale
        assert 2.0 ** 13 <= np.abs(Ws).max() < 2.0 ** 14, (what, float(np.abs(Ws).max()))
        assert np.array_equal(t[0], Ws.astype(np.float32).astype(np.float16).astype(np.float64)), what + ": hi is not fp16(2^e w)"
        err = np.abs(t[0] + t[1] - Ws)
        assert np.all(err <= np.abs(Ws) * 2.0 ** -23 + 2.0 ** -25), (what, float((err / np.maximum(np.abs(Ws), 1e-30)).max()))
        assert np.all(np.abs(t[1]) <= np.abs(t[0]) * 2.0 ** -11 + 2.0 ** -24), what    # lo is at most half an ulp of hi

    for i in range(1, 6):
        W = layers[i][0].numpy().astype(np.float64)
        co, ci = W.shape[:2]
        n_fl = (5 if ci == 16 else 9 * (ci // 32)) * 2 * 4 * co * 4
        terms = halves[2 * off: 2 * (off + n_fl)].astype(np.float64)
        if ci == 16:
            t = terms.reshape(5, 2, 2, 2, co, 8).transpose(1, 4, 3, 5, 0, 2).reshape(2, co, 16, 10)
            assert np.all(t[..., 9] == 0.0), "the pad half-step must multiply zeros"
            t = t[..., :9]
        else:
            t = terms.reshape(9, ci // 32, 2, 4, co, 8).transpose(2, 4, 1, 3, 5, 0).reshape(2, co, ci, 9)
        check_h2(W, t.reshape(2, co, ci, 3, 3), blob.numpy()[off + n_fl], "%s conv%d" % (name, i))
        off += n_fl + 4
    if kind == 2:
        n_fl = K * 128
        t = halves[2 * off: 2 * (off + n_fl)].astype(np.float64).reshape(K // 32, 2, 4, 128, 8).transpose(1, 0, 2, 4, 3).reshape(2, K, 128)
        check_h2(Wf, t, blob.numpy()[off + n_fl], "HardNet head")
        off += n_fl + 4
    assert off == blob.numel(), (off, blob.numel())


def test_roofline_table_regenerates_from_the_tracked_evidence():
    """tools/roofline_table.py on the newest full evidence set in profiles/: runs, prices the exact-fp32 trunks against the fp32 MFMA peak
    with no fraction above 1 (round 2 printed 204 % for the lazily evaluated AffNet), and labels the split-operand instantiations."""
    import glob
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sets = sorted(f[:-len("_kernel_stats.csv")] for f in glob.glob(os.path.join(root, "profiles", "r*_s*_kernel_stats.csv"))
                  if "split3" not in f and "split2h" not in f and "config" not in f and "onepass" not in f)
    sets = [s for s in sets if os.path.exists(s + "_traffic.json") and (os.path.exists(s + "_bench_default.json") or os.path.exists(s + "_bench_driver_detail.json"))]
    assert sets, "no full evidence set in profiles/"
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "roofline_table.py"), sets[-1]], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-400:]
    fracs = [float(m) for m in re.findall(r"= ([0-9.]+) % of 157\.3", out.stdout)]
    assert len(fracs) >= 4 and max(fracs) < 100.0 and max(fracs) > 80.0, fracs
    assert "bench line:" in out.stdout

def test_lds_bank_model_of_the_split_layouts():
    """tools/lds_bank_model.py (service groups and bank rules of MI355X_MICROARCH.md, section LDS) on the layouts the split-operand trunks use: every
    fragment read of the loops is conflict free (4 LDS cycles) except the ONE reader per net and mode the design knowingly leaves with 2-way conflicts
     - conv4 in LayQ, conv3 (HardNet) / conv4 (16-channel nets) in LayR, whose output buffer two readers with opposite group strides
    share.  A change of a row pitch or group stride in cnn32.hip that breaks this has to show up here (the model mirrors those parameters)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import lds_bank_model as bm
    got = {name: bm.read_cycles(L, stride, c16) for name, L, stride, c16 in bm.readers()}
    slow = {k for k, v in got.items() if v != 4}
    assert slow == {"HardNet fp32_split3 conv4 (stride 2, 16-wide)", "AffNet / OriNet fp32_split3 conv4 (stride 2, 16-wide)",
                    "HardNet fp32_split2h conv3 (stride 1, 16-wide)", "AffNet / OriNet fp32_split2h conv4 (stride 2, 16-wide)"}, got
    assert all(got[k] == 8 for k in slow), got
    # the parameters the model mirrors are the ones in the kernel source
    src = open(os.path.join(ROOT, "affnet_amd", "csrc", "cnn32.hip")).read()
    for frag in ("LayR<32, 32, 34, CB, 0>", "LayR<32, 32, 34, CB, 16>", "LayR<16, 16, 20, 2 * CB, 16>, LayQ<16, 16, 18, 2 * CB, 0, 3>", "LayR<16, 16, 20, 2 * CB, 0>, LayQ<16, 16, 18, 2 * CB, 0, 3>",
                 "LayR<8, 8, 12, 4 * CB, 0>, LayQ<8, 8, 16, 4 * CB, 128, 3>", "LayQ<16, 32, 34, CB, 0, 3>", "LayQ<16, 32, 34, CB, 16, 3>"):
        assert frag in src, frag
    assert open(os.path.join(ROOT, "profiles", "r04_lds_bank_model.txt")).read().count(" 4\n") >= 16

def test_split_arithmetics_error_class_cpu_emulation():
    """tools/split2h_numerics.py: conv-like dot products (K = 288, post-ReLU activations x N(0, 0.05) weights) against fp64 - the three-bf16-term and the
    (weight-scaled) two-fp16-term summations are not less accurate than the fp32 fmaf chain of the exact mode; without the weight scale the two-term form is
    (the reason the copies are packed times 2^e).  What include/affnet_hip.h states about AFFNET_ARITH_FP32_SPLIT3 / _SPLIT2H."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import split2h_numerics as sn
    r = sn.errors(288, M=1024)
    assert r["bf16x3"][0] <= r["fp32_chain"][0] and r["fp16x2"][0] <= r["fp32_chain"][0], r
    assert r["fp16x2"][1] <= 1.2 * r["fp32_chain"][1] and r["fp16x2_unscaled"][0] > 1.5 * r["fp16x2"][0], r
    assert r["fp32_chain"][0] < 1e-7 and r["fp16x2"][0] < 5e-8


def test_split2h_operand_error_bound_by_magnitude():
    """include/affnet_hip.h on AFFNET_ARITH_FP32_SPLIT2H: x ~ fp16(x) + fp16(x - fp16(x)) holds to 2^-23 RELATIVE only for
    |x| >= 2^-2; activations are not scaled, so below that the low term is a subnormal fp16 and the error is <= 2^-25 ABSOLUTE (2^-22 |x|
    at 2^-3, 2^-17 |x| at 2^-8).  A layer of uniformly small activations therefore loses relative precision in this mode - a dot product
    of such a layer against fp64 shows it, while the same layer at O(1) magnitudes does not."""
    rng = np.random.RandomState(0)

    def split_err(x):
        h = x.astype(np.float16)
        low = (x - h.astype(np.float32)).astype(np.float16)
        return np.abs(x.astype(np.float64) - h.astype(np.float64) - low.astype(np.float64))

    for lo, hi, rel_bound in ((-2, 15, 2.0 ** -23), (-3, -2, 2.0 ** -22), (-8, -5, 2.0 ** -17)):
        x = (2.0 ** rng.uniform(lo, hi, 400000) * rng.choice([-1, 1], 400000)).astype(np.float32)
        e = split_err(x)
        assert (e / np.abs(x)).max() <= rel_bound * (1 + 1e-9), (lo, hi)
        if hi <= -2:
            assert e.max() <= 2.0 ** -25 * (1 + 1e-9) and (e / np.abs(x)).max() > 2.0 ** -23      # the absolute floor, beyond the relative bound
    # a K = 288 dot product with two-term activations (weights exact here): O(1) activations vs the same values x 2^-10
    w = rng.normal(0, 0.05, (64, 288)).astype(np.float32)
    a = np.maximum(rng.normal(0, 1, (288,)), 0).astype(np.float32) + np.float32(0.3)
    out = []
    for scale in (1.0, 2.0 ** -10):
        x = (a * np.float32(scale)).astype(np.float32)
        two = x.astype(np.float16).astype(np.float64) + (x - x.astype(np.float16).astype(np.float32)).astype(np.float16).astype(np.float64)
        exact = w.astype(np.float64) @ x.astype(np.float64)
        out.append(np.abs(w.astype(np.float64) @ two - exact).max() / np.abs(w.astype(np.float64)).dot(np.abs(x.astype(np.float64))).max())
    assert out[0] < 2.0 ** -23 and out[1] > 8 * out[0], out


def test_reference_centroid_conv_order_the_detector_kernel_assumes():
    """csrc/detect.hip sums the detector's 27-tap centroid in the order of the reference's CPU conv2d: (level, ky, kx) for maps of at most
    20480 / 3 pixels (ATen's native im2col + sgemm path), (ky, kx, level) above (oneDNN) - round 5, DESIGN section 2.  This pins that
    assumption against the torch build of THIS host (tools/probes/cpu_conv_order.py): a torch / oneDNN upgrade that changes the order shows
    here, not as 1e-3 px LAF outliers on the GPU.  The large-map order holds on every host seen; the small-map order is a sequential chain on
    the authoring host (Intel MKL) and NOT on an AMD host (MKL's sgemm there sums differently: reference-side host variation) - then only
    the large-map half is asserted."""
    sys.path.insert(0, os.path.join(ROOT, "tools", "probes"))
    import cpu_conv_order as cco
    assert cco.orders_matching(83, 83) == ["klc"] and cco.orders_matching(96, 128) == ["klc"], "oneDNN's direct 3-channel convolution no longer sums in (ky, kx, level) order"
    small = cco.orders_matching(82, 83)
    if small:
        assert small == ["ckl"], small
    else:
        pytest.skip("this host's small-map sgemm is not a sequential fmaf chain (AMD host): the live oracle differs from tests/golden by an ulp in the small octaves")
