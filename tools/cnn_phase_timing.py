#!/usr/bin/env python
"""Per-phase cycle breakdown of the fused CNN trunk kernels (s_memtime stamps, tuning aid).
Runs each net on 2048 random patches and prints mean cycles per phase per wave."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import affnet_amd
from affnet_amd._lib import lib, ptr
from affnet_amd import engine

dev = torch.device("cuda:0")
n = int(os.environ.get("PHASE_PATCHES", "2048"))      # 256 = one workgroup per CU: the phases without a co-resident workgroup
p = (torch.rand(n, 1, 32, 32) * 255).to(dev)
A = affnet_amd.AffNetFast(); A.load_state_dict(torch.load(os.path.join(ROOT, "pretrained/AffNet.pth"), map_location="cpu", weights_only=False)["state_dict"]); A.to(dev)
H = affnet_amd.HardNet(); H.load_state_dict(affnet_amd.synthetic_hardnet_state(0)); H.to(dev)
names = ["input+norm", "conv0", "conv1 mfma", "conv1 store", "conv2 mfma", "conv2 store", "conv3 mfma", "conv3 store",
         "conv4 mfma", "conv4 store", "conv5 mfma", "conv5 store"]
for net, nm, nw in [(A, "AffNet", 8), (H, "HardNet", 8)]:
    net(p); torch.cuda.synchronize()
    st = torch.zeros(n * nw * 32, dtype=torch.int64, device=dev)
    lib.affnet_cnn32_debug_timing(engine.utility_ctx(dev), ptr(st))
    net(p); torch.cuda.synchronize()
    lib.affnet_cnn32_debug_timing(engine.utility_ctx(dev), None)
    t = st.cpu().numpy().reshape(n, nw, 32).astype(np.float64)
    nst = 12
    d = np.diff(t[:, :, :nst], axis=2)              # cycles per phase per wave
    wg = t[:, :, :nst].max(axis=1) - t[:, :, 0:1].min(axis=1)   # per patch: boundary times relative to WG start
    print("== %s (%d waves): mean phase cycles per wave (s_memtime ticks, 100 MHz const clock?)" % (nm, nw))
    tot = (t[:, :, nst - 1].max(axis=1) - t[:, :, 0].min(axis=1)).mean()
    span = (t[:, :, nst - 1].max() - t[:, :, 0].min())
    print("  kernel span %.0f ticks for %d patches -> %.0f ticks / patch / CU-slot (256 CUs)" % (span, n, span / (n / 256.0)))
    for i in range(nst - 1):
        print("  %-12s mean %9.0f  max-over-waves %9.0f" % (names[i], d[:, :, i].mean(), d[:, :, i].max(axis=1).mean()))
    print("  total per patch (WG) %.0f ticks" % tot)
    simd = (st.cpu().numpy().reshape(n, nw, 32)[:, :, 14] >> 4) & 3
    for i in (2, 6, 10):
        print("  %-12s per wave:" % names[i], " ".join("%6.0f" % v for v in d[:, :, i].mean(axis=0)))
    # the two waves of a workgroup that share a SIMD: how far apart do they finish a loop?
    first, second = [], []
    for k in range(0, n, max(1, n // 128)):
        for sd in range(4):
            w = np.where(simd[k] == sd)[0]
            if len(w) == 2:
                a, b = d[k, w[0], 2], d[k, w[1], 2]
                first.append(min(a, b)); second.append(max(a, b))
    if first:
        print("  conv1 mfma, SIMD-sharing pairs: faster wave %.0f, slower wave %.0f ticks; SIMD ids of waves 0..%d in patch 0: %s" %
              (np.mean(first), np.mean(second), nw - 1, simd[0].tolist()))
    ti = st.cpu().numpy().reshape(n, nw, 32)
    start, end = ti[:, :, 0].min(axis=1).astype(np.float64), ti[:, :, 13].max(axis=1).astype(np.float64)
    print("  last phase stamp -> kernel end (head / global store): %.0f ticks" % (ti[:, :, 13] - ti[:, :, nst - 1]).mean())
    hw, xcc = ti[:, 0, 14], ti[:, 0, 15] & 0xF
    cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)
    util, gaps = [], []
    for c in np.unique(cu):
        idx = np.where(cu == c)[0]
        s_, e_ = start[idx], end[idx]
        o = np.argsort(s_)
        s_, e_ = s_[o], e_[o]
        span = e_.max() - s_.min()
        util.append((e_ - s_).sum() / span)
        # gap between a WG ending and the next WG starting on this CU (slot-agnostic: sorted ends vs later starts)
        ends = np.sort(e_)
        for k in range(len(s_)):
            prev = ends[ends <= s_[k]]
            if len(prev):
                gaps.append(s_[k] - prev.max())
    sub = t[:, :, [0, 16, 17, 18, 1, 19, 20, 2]]
    ds = np.diff(sub, axis=2).mean(axis=(0, 1))
    print("  input: load %.0f | halo+sum1 %.0f | sum2 %.0f | patch write+barrier %.0f || conv0: mfma %.0f | stores %.0f | barrier %.0f" % tuple(ds))
    sub = t[:, :, [3, 21, 22, 4]]
    ds = np.diff(sub, axis=2).mean(axis=(0, 1))
    print("  conv1 epilogue: barrier1 %.0f | halo+stores %.0f | barrier2 %.0f" % tuple(ds))
    print("  CUs seen %d; mean resident WGs per CU %.2f; median end->next-start gap %.0f ticks (p90 %.0f)" %
          (len(util), np.mean(util), np.median(gaps), np.percentile(gaps, 90)))
    # wall time of the same launch without stamps
    lib.affnet_cnn32_debug_timing(engine.utility_ctx(dev), None)
    big = (torch.rand(48000, 1, 32, 32) * 255).to(dev)
    net(big); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); net(big); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    fl = {"AffNet": 19193856.0, "HardNet": 78184448.0}[nm]
    print("  48000 contiguous patches: %.3f ms -> %.1f TFLOP/s (incl. head / allocation)" % (ms, 48000 * fl / ms / 1e9))
