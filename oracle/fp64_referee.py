"""fp64 referee of the post-detector stages + an accounting of the keypoints only one side returns.

TEST INFRASTRUCTURE - NOT PRODUCT CODE (same rule as affnet_oracle.py: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline / parity_check leg import it).

Why it exists.  The reference's CPU path and the HIP path evaluate the same fp32 formulas in different summation orders.  Two questions
remain after matching rows by their integer key (octave, level, pixel):

1. A matched LAF row differs by more than 1e-3 px (1 - 2 rows of 2000: large frames or short OriNet vectors).  Whose rounding is that?
   The referee evaluates the SAME stages - patch sampling (LAF.py:313-372), AffNetFast (architectures.py:204-252), shape composition
   (SparseImgRepresenter.py:121-146), OriNetFast (architectures.py:33-82), rotation (LAF.py:276-283, SparseImgRepresenter.py:173-177) and
   denormalisation (LAF.py:407-417) - in float64 with the weights `.double()`, on the fp32 detector output (which is bit-identical on both
   sides).  What is asserted for a row outside 1e-3 px has no constant that was chosen after looking at results (round 6; round
   5's third clause - a quantile of the reference's own errors, moved from the median to the 90th percentile after a failure - is gone):
     (a) |GPU - fp64| <= |CPU-fp32 reference - fp64| + 1e-3 px: the GPU row is as close to the exact result as the reference's own fp32
         row, up to the BASELINE tolerance; or
     (b) the reference's own row is ILL-CONDITIONED by its own measure - |CPU - fp64| >= 1e-3 px: no two fp32 evaluations agree on it to
         the tolerance - AND the GPU's error is bounded by it: |GPU - fp64| <= ILL_FACTOR |CPU - fp64|.
   A row that meets neither is counted in `rows_outside_1e-3_beyond_referee` and LISTED; the gate is a count budget stated here, before any
   run: at most BEYOND_BUDGET_PER_4000 such row per 4000 matched rows (rounded up per image) against an oracle run LIVE on a foreign host
   (whose own rounding differs from the golden host's in a few operators); the golden vectors (tests/golden/, authoring host) are compared at
   the plain tolerance with no referee at all.  Above both stands an UNCONDITIONAL ceiling: no matched row may differ by ABS_CEILING_PX =
   1e-2 px (`rows_outside_1e-2`), whatever its conditioning.  (Both sides' fp32 errors are the same class: over all rows of graf img1 the GPU
   is closer to fp64 in 820 - 901 rows, the CPU in 862 - 931, p50 / p99 / max of both within 5 %.)  Round 5 found with it that
   every such row at <= 1024 x 768 was a REAL discrepancy - the detector's 27-tap centroid summed in another order than the reference's
   conv2d for maps above 6826 px, one ulp of a sub-pixel centre, amplified by the patch sampling - and fixed it (csrc/detect.hip).
2. A key is returned by one side only (6 of 4000).  The shape filter (SparseImgRepresenter.py:147-162) takes HARD decisions on AffNet
   outputs: `d1 > 0` and `1/6 < |l1 / (l2 + 1e-8)| < 6` (Utils.py:168-175), all four frame corners inside [0, 1]^2 (LAF.py:98-104),
   then `topk(resp * good, N)`.  `explain_unmatched` traces every such key to the candidate it belongs to and requires that (a) the
   decision that differs is BORDERLINE in the fp64 evaluation - eigen-ratio within 1e-4 relative of {1/6, 6}, discriminant within the
   fp32 rounding error of tr^2, or a corner within 1e-3 px of the image boundary - or (b) the row was displaced at the top-N cut by
   such a row (or ties the cut response).  Anything else counts as `unmatched_unexplained` and fails the test.
"""
import numpy as np
import torch
import torch.nn.functional as F

import affnet_oracle as orc

RATIO_TOL = 1e-4              # relative distance of |l1 / l2| from 6 or 1/6 that counts as borderline
CORNER_TOL_PX = 1e-3          # distance of a frame corner from the image boundary (px) that counts as borderline
ILL_FACTOR = 4.0              # clause (b): an ill-conditioned row's GPU error may be this multiple of the CPU reference's own error vs float64 (fixed in round 6 BEFORE any run)
ABS_CEILING_PX = 1e-2         # unconditional: no matched row may differ from the reference's row by this much
BEYOND_BUDGET_PER_4000 = 1    # rows meeting neither clause that a comparison with a LIVE oracle on a foreign host may contain, per 4000 matched rows
DISC_TOL = 4.0 * 2.0 ** -24   # |tr^2 - 4 det| <= DISC_TOL * tr^2: the sign of the fp32 discriminant is decided by the rounding of tr^2
                              # (Utils.py:170: three fp32 roundings of quantities of size tr^2, 2^-24 relative each, + one of slack)


def beyond_budget(matched):
    """Count budget of `rows_outside_1e-3_beyond_referee` for `matched` rows (stated up front, module docstring)."""
    return BEYOND_BUDGET_PER_4000 * max(1, -(-int(matched) // 4000))


def _double_sd(sd):
    return {k: (v.double() if torch.is_floating_point(v) else v) for k, v in sd.items()}


def conv64(x, w, bias=None, stride=1, padding=1):
    """F.conv2d semantics (cross-correlation) as im2col + one dgemm: ATen's float64 convolution is the same algorithm, three times slower."""
    n, _, h, wd = x.shape
    co, _, k, _ = w.shape
    cols = F.unfold(x, (k, k), padding=padding, stride=stride)
    out = torch.matmul(w.reshape(co, -1), cols)
    ho = (h + 2 * padding - k) // stride + 1
    out = out.view(n, co, ho, -1)
    return out if bias is None else out + bias.view(1, -1, 1, 1)


def input_norm64(x):
    """architectures.py:235-239: per-patch mean, unbiased std + 1e-7."""
    flat = x.reshape(x.size(0), -1)
    return (x - flat.mean(dim=1).view(-1, 1, 1, 1)) / (flat.std(dim=1) + 1e-7).view(-1, 1, 1, 1)


def trunk64(sd, x):
    """architectures.py:207-224 == :36-53 == HardNet.py:67-84: six conv3x3 + BatchNorm(eval, no affine, eps 1e-5) + ReLU."""
    for ci, bi, st in orc._TRUNK:
        x = conv64(x, sd["features.%d.weight" % ci], None, st, 1)
        m, v = sd["features.%d.running_mean" % bi], sd["features.%d.running_var" % bi]
        x = torch.relu((x - m.view(1, -1, 1, 1)) / torch.sqrt(v.view(1, -1, 1, 1) + 1e-5))
    return x


def affnet64(sd, patches):
    """architectures.py:204-252 in float64 -> rectified (n, 2, 2)."""
    y = torch.tanh(conv64(trunk64(sd, input_norm64(patches)), sd["features.19.weight"], sd["features.19.bias"], 1, 0)).view(-1, 3)
    A = torch.zeros(y.size(0), 2, 2, dtype=torch.float64)
    A[:, 0, 0] = 1.0 + y[:, 0]
    A[:, 1, 0] = y[:, 1]
    A[:, 1, 1] = 1.0 + y[:, 2]
    return orc.rectify_up_is_up(A)          # LAF.py:285-291: dtype-agnostic arithmetic


def orinet_vec64(sd, patches):
    """architectures.py:33-80 in float64: the (n, 2) vector fed to atan2."""
    y = torch.tanh(conv64(trunk64(sd, input_norm64(patches)), sd["features.19.weight"], sd["features.19.bias"], 1, 1))
    return y.mean(dim=(2, 3))


def hardnet64(sd, patches):
    """HardNet.py:61-101 in float64."""
    y = conv64(trunk64(sd, input_norm64(patches)), sd["features.19.weight"], None, 1, 0).view(patches.size(0), -1)
    y = (y - sd["features.20.running_mean"].view(1, -1)) / torch.sqrt(sd["features.20.running_var"].view(1, -1) + 1e-5)
    return y / torch.sqrt((y * y).sum(dim=1, keepdim=True) + 1e-8)


def extract_patches64(img, lafs, ps):
    """LAF.py:313-372 in float64 (affine_grid + grid_sample, bilinear, zeros padding, align_corners=False)."""
    h, w = img.size(2), img.size(3)
    n = lafs.size(0)
    if n == 0:
        return torch.zeros(0, 1, ps, ps, dtype=torch.float64)
    m = float(min(h, w))
    coef = torch.tensor([[m, m, float(w)], [m, m, float(h)]], dtype=torch.float64)
    grid = F.affine_grid(lafs * coef.unsqueeze(0), torch.Size((n, 1, ps, ps)), align_corners=False)
    grid = torch.stack([2.0 * grid[..., 0] / float(w) - 1.0, 2.0 * grid[..., 1] / float(h) - 1.0], dim=-1)
    out = torch.zeros(n, 1, ps, ps, dtype=torch.float64)
    for s in range(0, n, 64):
        e = min(n, s + 64)
        out[s:e] = F.grid_sample(img.expand(e - s, 1, h, w), grid[s:e], mode="bilinear", padding_mode="zeros", align_corners=False)
    return out


def keys_of(octs, levs, pixs):
    return np.asarray(octs, dtype=np.int64) * (1 << 40) + np.asarray(levs, dtype=np.int64) * (1 << 32) + np.asarray(pixs, dtype=np.int64)


class Referee(object):
    """fp64 evaluation of the post-detector stages for chosen CANDIDATES of an OracleExtractor that has run on an image
    (ex.detected = the C = 1.5 N detector rows, normalised LAFs already x mrSize; ex.scale_pyr = the fp32 pyramid)."""

    def __init__(self, ex, width, height, ps=32):
        assert ex.aff is not None, "the referee covers the AffNetFast / OriNetFast slots"
        self.ex, self.w, self.h, self.ps = ex, int(width), int(height), ps
        self.aff = _double_sd(ex.aff)
        self.ori = None if ex.ori is None else _double_sd(ex.ori)
        d = ex.detected
        self.octs, self.levs, self.pixs = d["oct"].numpy(), d["lev"].numpy(), d["pix"].numpy()
        self.resp = d["resp"].numpy()
        self.cand_lafs = d["lafs"].double()
        self.cand_keys = keys_of(self.octs, self.levs, self.pixs)
        self.pos = {int(k): i for i, k in enumerate(self.cand_keys)}
        self._pyr = {}
        self._shape = {}          # candidate index -> (A (2,2), frame (2,3)) float64
        self._laf = {}            # candidate index -> pixel LAF (2,3) float64
        self._ovec = {}           # candidate index -> OriNet vector (2,) float64 (before atan2)

    def level(self, o, l):
        if (o, l) not in self._pyr:
            self._pyr[(o, l)] = self.ex.scale_pyr[o][l].double()
        return self._pyr[(o, l)]

    def _sample(self, idx, lafs):
        out = torch.zeros(len(idx), 1, self.ps, self.ps, dtype=torch.float64)
        ol = self.octs[idx] * 64 + self.levs[idx]
        for code in np.unique(ol):
            sel = np.nonzero(ol == code)[0]
            out[sel] = extract_patches64(self.level(int(code) // 64, int(code) % 64), lafs[sel], self.ps)
        return out

    def shapes(self, idx):
        """Candidates idx -> (A (k,2,2), frames (k,2,3) normalised) in float64: SparseImgRepresenter.py:121-146 with num_Baum_iters = 1."""
        idx = np.asarray(idx, dtype=np.int64)
        todo = np.array([i for i in idx if int(i) not in self._shape], dtype=np.int64)
        for s in range(0, len(todo), 512):
            part = todo[s:s + 512]
            lafs = self.cand_lafs[part]
            A = affnet64(self.aff, self._sample(part, lafs))
            fr = torch.cat([torch.bmm(A, lafs[:, :, :2]), lafs[:, :, 2:]], dim=2)
            for j, i in enumerate(part):
                self._shape[int(i)] = (A[j], fr[j])
        if len(idx) == 0:
            return torch.zeros(0, 2, 2, dtype=torch.float64), torch.zeros(0, 2, 3, dtype=torch.float64)
        return torch.stack([self._shape[int(i)][0] for i in idx]), torch.stack([self._shape[int(i)][1] for i in idx])

    def decisions(self, idx, A=None, frames=None):
        """The quantities the shape filter decides on (Utils.py:168-175, LAF.py:98-104), evaluated in float64 - on the referee's own fp64
        AffNet outputs, or on given (A, frames) of either side cast to float64.  Per candidate: ratio, its relative distance to the nearer of
        {1/6, 6}, the discriminant relative to tr^2, the smallest corner-to-boundary distance in px (negative = outside), and the decision."""
        if A is None:
            A, frames = self.shapes(idx)
        A, frames = A.double(), frames.double()
        tr = A[:, 0, 0] + A[:, 1, 1]
        d1 = tr * tr - 4.0 * (A[:, 0, 0] * A[:, 1, 1] - A[:, 1, 0] * A[:, 0, 1])
        ok = d1 > 0
        sq = torch.sqrt(torch.abs(d1))
        l1 = torch.where(ok, (tr + sq) / 2.0, torch.full_like(tr, 1000.0))
        l2 = torch.where(ok, (tr - sq) / 2.0, torch.full_like(tr, 1e-4))
        ratio = torch.abs(l1 / (l2 + 1e-8))
        rmargin = torch.minimum(torch.abs(ratio - 6.0) / 6.0, torch.abs(ratio - 1.0 / 6.0) * 6.0)
        c = orc.frame_corners(frames)                                           # (k, 2, 4) normalised: x by W, y by H
        scale = torch.tensor([float(self.w), float(self.h)], dtype=torch.float64).view(1, 2, 1)
        cmargin = (torch.minimum(c, 1.0 - c) * scale).reshape(len(A), -1).min(dim=1).values
        good = ok & (ratio < 6.0) & (ratio > 1.0 / 6.0) & (cmargin >= 0)
        return {"ratio": ratio.numpy(), "ratio_margin_rel": rmargin.numpy(), "disc_over_tr2": (d1 / (tr * tr)).numpy(),
                "corner_margin_px": cmargin.numpy(), "good": good.numpy()}

    def borderline(self, dec, k):
        """Which hard decision of candidate row k of `dec` sits within rounding distance of its threshold (None if none does)."""
        if abs(dec["disc_over_tr2"][k]) <= DISC_TOL:
            return "discriminant"
        if dec["ratio_margin_rel"][k] <= RATIO_TOL:
            return "eigen_ratio"
        if abs(dec["corner_margin_px"][k]) <= CORNER_TOL_PX:
            return "corner"
        return None

    def lafs_px(self, idx):
        """Candidates idx -> the pixel LAFs (k,2,3) the path returns for them with do_ori=True, all in float64:
        SparseImgRepresenter.py:163-180, 199-203."""
        idx = np.asarray(idx, dtype=np.int64)
        todo = np.array([i for i in idx if int(i) not in self._laf], dtype=np.int64)
        m = float(min(self.w, self.h))
        coef = torch.tensor([[m, m, float(self.w)], [m, m, float(self.h)]], dtype=torch.float64)
        for s in range(0, len(todo), 512):
            part = todo[s:s + 512]
            _, fr = self.shapes(part)
            if self.ori is not None:
                v = orinet_vec64(self.ori, self._sample(part, fr))
                for j, i in enumerate(part):
                    self._ovec[int(i)] = v[j]
                ang = torch.atan2(v[:, 0] + 1e-8, v[:, 1] + 1e-8)
                fr = torch.cat([torch.bmm(fr[:, :, :2], orc.rotation_matrix(ang)), fr[:, :, 2:]], dim=2)
            fr = fr * coef
            for j, i in enumerate(part):
                self._laf[int(i)] = fr[j]
        if len(idx) == 0:
            return torch.zeros(0, 2, 3, dtype=torch.float64)
        return torch.stack([self._laf[int(i)] for i in idx])


def equivalent_output_error(ref, cidx, err_px):
    """A LAF row error in px expressed as the error of the CNN output that produces it: a rotation by d_angle moves a frame of scale
    S = sqrt|det A| px by S d_angle, and d_angle = |d_o| / |o| for OriNet's vector o (architectures.py:78-81) - so err |o| / S is the size of the
    OriNet-output error that explains err (an upper bound when part of the error is shape, not rotation).  S and |o| from the fp64 evaluation."""
    cidx = np.asarray(cidx, dtype=np.int64)
    L64 = ref.lafs_px(cidx).numpy()
    S = np.sqrt(np.abs(L64[:, 0, 0] * L64[:, 1, 1] - L64[:, 0, 1] * L64[:, 1, 0]))
    on = np.array([float(torch.linalg.vector_norm(ref._ovec[int(c)])) for c in cidx]) if ref.ori is not None else np.ones(len(cidx))
    return np.asarray(err_px, dtype=np.float64) * np.minimum(on, 1.0) / np.maximum(S, 1e-30), S, on


def explain_unmatched(ref, keys_gpu, n_out):
    """Accounts for every key that only one side returns.  ref: Referee of the oracle run (ex.shape_stage holds the CPU-fp32 decisions);
    keys_gpu: (n, 3) (octave, level, pixel) of the rows the HIP path returned; n_out: the N of topk (<= 0: threshold mode, no cut).
    Returns a dict whose "unmatched_unexplained" must be 0; every unmatched key is listed with the decision that differs and its margins."""
    ex = ref.ex
    st = ex.shape_stage
    good_cpu = st["good"].numpy().astype(bool)
    kg = keys_of(*np.asarray(keys_gpu, dtype=np.int64).T) if len(keys_gpu) else np.zeros(0, dtype=np.int64)
    kc = keys_of(*ex.keys.numpy().T) if len(ex.keys) else np.zeros(0, dtype=np.int64)
    set_g, set_c = set(int(k) for k in kg), set(int(k) for k in kc)
    only_g, only_c = sorted(set_g - set_c), sorted(set_c - set_g)
    rows, unexplained = [], 0
    flips = {}                                      # candidate index -> the GPU's decision, for the borderline ones
    pending = []
    for side, ks in (("gpu_only", only_g), ("cpu_only", only_c)):
        for k in ks:
            c = ref.pos.get(k)
            if c is None:
                rows.append({"key_octave_level_pixel": [int(k >> 40), int((k >> 32) & 255), int(k & 0xFFFFFFFF)], "side": side, "why": "NOT A DETECTOR CANDIDATE"})
                unexplained += 1
                continue
            pending.append((side, k, c))
    if pending:
        cidx = np.array([c for _, _, c in pending], dtype=np.int64)
        d64 = ref.decisions(cidx)
        d32 = ref.decisions(cidx, st["A"][cidx], st["frames"][cidx])     # the CPU reference's fp32 values, decision quantities in fp64
        for j, (side, k, c) in enumerate(pending):
            why = ref.borderline(d64, j) or ref.borderline(d32, j)
            row = {"key_octave_level_pixel": [int(ref.octs[c]), int(ref.levs[c]), int(ref.pixs[c])], "side": side, "candidate_rank": int(c),
                   "response": float(ref.resp[c]), "cpu_good": bool(good_cpu[c]), "fp64_good": bool(d64["good"][j]),
                   "ratio_fp64": float(d64["ratio"][j]), "ratio_margin_rel": float(min(d64["ratio_margin_rel"][j], d32["ratio_margin_rel"][j])),
                   "disc_over_tr2_fp64": float(d64["disc_over_tr2"][j]), "disc_over_tr2_cpu": float(d32["disc_over_tr2"][j]),
                   "corner_margin_px": float(d64["corner_margin_px"][j]), "corner_margin_px_cpu": float(d32["corner_margin_px"][j])}
            gpu_good = side == "gpu_only"
            if why is not None and gpu_good != bool(good_cpu[c]):
                row["why"] = "borderline " + why
                flips[c] = gpu_good
            else:
                row["why"] = None                   # decided below: displaced at the top-N cut, or unexplained
            rows.append(row)
    # replay the cut with the CPU decisions + the borderline flips: the result must be the GPU's key set (SparseImgRepresenter.py:152-158)
    good_sim = good_cpu.copy()
    for c, g in flips.items():
        good_sim[c] = g
    if n_out > 0 and good_sim.sum() > n_out:
        val = ref.resp * good_sim
        order = np.argsort(-val, kind="stable")
        cut = val[order[n_out - 1]]
        sim = set(int(k) for k in ref.cand_keys[order[:n_out]])
        tied = set(int(k) for k in ref.cand_keys[val == cut])
    else:
        sim = set(int(k) for k in ref.cand_keys[good_sim])
        cut, tied = None, set()
    leftover = (sim ^ set_g) - tied
    for row in rows:
        if row.get("why") is None:
            k = keys_of(*row["key_octave_level_pixel"])
            if int(k) not in leftover:
                row["why"] = "tie at the top-N cut" if int(k) in tied else "displaced at the top-N cut by a borderline row"
            else:
                row["why"] = "UNEXPLAINED"
                unexplained += 1
    extra = [k for k in leftover if k not in set(only_g) | set(only_c)]          # the replay disagrees on a key both sides agree on
    unexplained += len(extra)
    return {"gpu_only": len(only_g), "cpu_only": len(only_c), "borderline_flips": len(flips), "unmatched_unexplained": int(unexplained),
            "cut_response": None if cut is None else float(cut), "rows": rows}


def referee_rows(ref, keys, L_gpu, L_cpu, rows=None):
    """For matched rows (same key on both sides; L_gpu / L_cpu (n,2,3) px in the same order, keys (n,3)): the max-entry distance of either
    side to the fp64 evaluation.  rows: subset to evaluate (default all).  Returns (rows, err_gpu_vs_fp64, err_cpu_vs_fp64, L64)."""
    keys = np.asarray(keys, dtype=np.int64)
    rows = np.arange(len(keys)) if rows is None else np.asarray(rows, dtype=np.int64)
    if len(rows) == 0:
        z = np.zeros(0)
        return rows, z, z, np.zeros((0, 2, 3))
    cidx = np.array([ref.pos[int(k)] for k in keys_of(*keys[rows].T)], dtype=np.int64)
    L64 = ref.lafs_px(cidx).numpy()
    eg = np.abs(np.asarray(L_gpu, dtype=np.float64)[rows] - L64).reshape(len(rows), -1).max(axis=1)
    ec = np.abs(np.asarray(L_cpu, dtype=np.float64)[rows] - L64).reshape(len(rows), -1).max(axis=1)
    return rows, eg, ec, L64


def parity_account(ref, ids_gpu, L_gpu, n_out, full=False):
    """Everything the parity statement needs for one image: ref = Referee of the oracle run (ex.keys / the LAFs it returned are the CPU
    rows), ids_gpu (n,3) / L_gpu (n,2,3) px = the HIP path's rows.  Returns a JSON-able record:
      unmatched_unexplained        keys returned by one side only that are neither a borderline decision nor displaced at the cut (must be 0)
      rows_outside_1e-3            matched rows whose GPU and CPU LAFs differ by >= 1e-3 px (max entry)
      rows_worse_than_cpu_vs_fp64  of those, rows with |GPU - fp64| > |CPU - fp64| + 1e-3 px
      rows_outside_1e-3_beyond_referee  rows outside 1e-3 px that meet neither clause (a) nor clause (b) of the module docstring; gate:
                                   <= beyond_budget(matched) against a live oracle.  (`rows_outside_1e-3_unexplained` is the same number under
                                   its round-5 name, kept for the recorded reports.)
      rows_outside_1e-2            matched rows that differ by >= ABS_CEILING_PX, unconditionally (must be 0)
      rows_outside_5e-3_unexplained  rows that differ by >= 5e-3 px although the CPU reference's own row is within 5e-3 px of fp64 (must be 0)
    full=True evaluates the referee on EVERY matched row (seconds per 2000 rows) and adds the distributions of both sides' distance to fp64."""
    ex = ref.ex
    ids_gpu = np.asarray(ids_gpu, dtype=np.int64)
    L_gpu = np.asarray(L_gpu, dtype=np.float64)
    L_cpu = np.asarray(ex.last_lafs_px, dtype=np.float64)          # the rows OracleExtractor.forward returned (px)
    kg, kc = keys_of(*ids_gpu.T), keys_of(*ex.keys.numpy().T)
    pos = {int(k): i for i, k in enumerate(kc)}
    gi = np.array([i for i, k in enumerate(kg) if int(k) in pos], dtype=np.int64)
    wi = np.array([pos[int(kg[i])] for i in gi], dtype=np.int64)
    dl = np.abs(L_gpu[gi] - L_cpu[wi]).reshape(len(gi), -1).max(axis=1) if len(gi) else np.zeros(0)
    out = np.nonzero(dl >= 1e-3)[0]
    rows = np.arange(len(gi)) if full else out
    rows, eg, ec, _ = referee_rows(ref, ids_gpu[gi], L_gpu[gi], L_cpu[wi], rows)
    at = {int(r): j for j, r in enumerate(rows)}
    worse, beyond, beyond_ceiling, listed = 0, 0, 0, []
    for r in out:
        j = at[int(r)]
        c = ref.pos[int(kg[gi[r]])]
        bad = bool(eg[j] > ec[j] + 1e-3)                                  # fails clause (a)
        ill = bool(ec[j] >= 1e-3)                                         # the CPU reference's own fp32 row misses the float64 result by the tolerance
        bounded = bool(ill and eg[j] <= ILL_FACTOR * ec[j])               # clause (b)
        u, S, on = equivalent_output_error(ref, [c], [eg[j]])
        row = {"key_octave_level_pixel": [int(v) for v in ids_gpu[gi[r]]], "gpu_vs_cpu_px": float(dl[r]), "gpu_vs_fp64_px": float(eg[j]),
               "cpu_vs_fp64_px": float(ec[j]), "gpu_closer_to_fp64_than_cpu": bool(eg[j] <= ec[j]), "worse_than_cpu_by_more_than_1e-3": bad,
               "reference_row_itself_1e-3_from_fp64": ill, "beyond_referee": bool(bad and not bounded), "frame_scale_px": float(S[0]), "orinet_norm": float(on[0]),
               "equivalent_output_error": float(u[0])}
        worse += bad
        beyond += bool(bad and not bounded)
        beyond_ceiling += bool(dl[r] >= 5e-3 and ec[j] < 5e-3)       # a row may differ by 5e-3 px only where the reference's own row is that far from fp64
        listed.append(row)
    exp = explain_unmatched(ref, ids_gpu, n_out)
    rec = {"keypoints_cpu": int(len(kc)), "keypoints_gpu": int(len(kg)), "matched": int(len(gi)),
           "unmatched_keys": exp["gpu_only"] + exp["cpu_only"], "unmatched_borderline_flips": exp["borderline_flips"],
           "unmatched_unexplained": exp["unmatched_unexplained"], "unmatched_rows": exp["rows"],
           "rows_outside_1e-3": int(len(out)), "rows_worse_than_cpu_vs_fp64": int(worse), "rows_outside_1e-3_beyond_referee": int(beyond),
           "rows_outside_1e-3_unexplained": int(beyond), "beyond_budget": beyond_budget(len(gi)), "rows_outside_1e-2": int((dl >= ABS_CEILING_PX).sum()),
           "rows_outside_5e-3_unexplained": int(beyond_ceiling), "rows_outside_1e-3_vs_fp64": listed,
           "laf_max_px_gpu_vs_cpu": float(dl.max()) if len(dl) else 0.0}
    if full and len(rows):
        q = lambda a: [float(np.percentile(a, p)) for p in (50, 99, 100)]
        rec["referee_all_rows"] = {"rows": int(len(rows)), "gpu_vs_fp64_px_p50_p99_max": q(eg), "cpu_vs_fp64_px_p50_p99_max": q(ec),
                                   "gpu_rows_beyond_1e-3_of_fp64": int((eg > 1e-3).sum()), "cpu_rows_beyond_1e-3_of_fp64": int((ec > 1e-3).sum()),
                                   "rows_where_gpu_is_closer": int((eg < ec).sum()), "rows_where_cpu_is_closer": int((ec < eg).sum())}
    return rec
