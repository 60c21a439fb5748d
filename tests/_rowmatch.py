"""Row matching against golden vectors of the UNMODIFIED reference, which carry no integer keypoint keys: rows are matched through
the bit pattern of the response (responses are bit-identical between the reference and the HIP path) and, where several rows carry
the SAME response (exact ties do occur: two maxima of a symmetric structure), through the frame centre.  A dict keyed on the response
alone maps both rows of a tie to one golden row and reports their distance (50 px in round 2's report) as a LAF error."""
import numpy as np


def match_rows(resp_got, lafs_got, resp_ref, lafs_ref):
    """Returns (gi, wi): rows gi of `got` paired with rows wi of `ref`.  Rows whose response bit pattern does not occur in the other
    set stay unmatched; inside a group of equal responses rows are paired greedily by centre distance (closest first)."""
    rg = np.ascontiguousarray(resp_got, dtype=np.float32).view(np.uint32)
    rr = np.ascontiguousarray(resp_ref, dtype=np.float32).view(np.uint32)
    cg = np.asarray(lafs_got, dtype=np.float64)[:, :, 2]
    cr = np.asarray(lafs_ref, dtype=np.float64)[:, :, 2]
    groups = {}
    for i, v in enumerate(rr):
        groups.setdefault(int(v), ([], []))[1].append(i)
    for i, v in enumerate(rg):
        if int(v) in groups:
            groups[int(v)][0].append(i)
    gi, wi = [], []
    for g_rows, r_rows in groups.values():
        if not g_rows:
            continue
        if len(g_rows) == 1 and len(r_rows) == 1:
            gi.append(g_rows[0]); wi.append(r_rows[0])
            continue
        pairs = sorted(((float(np.abs(cg[a] - cr[b]).max()), a, b) for a in g_rows for b in r_rows))
        used_a, used_b = set(), set()
        for _, a, b in pairs:
            if a not in used_a and b not in used_b:
                used_a.add(a); used_b.add(b)
                gi.append(a); wi.append(b)
    order = np.argsort(np.array(gi, dtype=np.int64), kind="stable") if gi else np.zeros(0, dtype=np.int64)
    return np.array(gi, dtype=np.int64)[order], np.array(wi, dtype=np.int64)[order]


def tie_groups(resp):
    """Number of rows that share their response bit pattern with another row."""
    r = np.ascontiguousarray(resp, dtype=np.float32).view(np.uint32)
    _, counts = np.unique(r, return_counts=True)
    return int(counts[counts > 1].sum())
