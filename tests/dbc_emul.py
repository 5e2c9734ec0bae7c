"""TEST INFRASTRUCTURE: numpy emulation of the tables `launch_dbc` (csrc/kernels_tail.hip) compacts on
the device, so that the host consumer `ctd_db_boxes_compact` can be checked without a GPU (and the
device kernels against this emulation with one)."""
import ctypes as C

import numpy as np

from oracle import postproc_ref as R


def dbc_tables(prob: np.ndarray, bitmap: np.ndarray):
    H, W = bitmap.shape
    nf, lab_f, st_f = R.connected_components_with_stats(bitmap.astype(np.uint8), 8)
    nb, lab_b, st_b = R.connected_components_with_stats((~bitmap.astype(bool)).astype(np.uint8), 4)
    nf, nb = nf - 1, nb - 1
    st_f, st_b = st_f[1:].astype(np.int32).reshape(-1, 5), st_b[1:].astype(np.int32).reshape(-1, 5)
    flat_f, flat_b = lab_f.ravel(), lab_b.ravel()

    def first_pixels(flat, n):
        first = np.full(n + 1, -1, np.int64)
        idx = np.nonzero(flat)[0]
        if idx.size == 0:
            return first[1:].astype(np.int32)
        # first occurrence per label
        lab = flat[idx]
        order = np.argsort(lab, kind="stable")
        l_sorted, i_sorted = lab[order], idx[order]
        keep = np.r_[True, l_sorted[1:] != l_sorted[:-1]]
        first[l_sorted[keep]] = i_sorted[keep]
        return first[1:].astype(np.int32)

    first_f, first_b = first_pixels(flat_f, nf), first_pixels(flat_b, nb)
    par_f = np.array([flat_b[p - 1] if p % W > 0 else 0 for p in first_f], np.int32).reshape(-1)
    hole = (st_b[:, 0] > 0) & (st_b[:, 1] > 0) & (st_b[:, 0] + st_b[:, 2] < W) & (st_b[:, 1] + st_b[:, 3] < H) if nb else np.zeros(0, bool)
    par_b = np.array([flat_f[p - 1] if hl else 0 for p, hl in zip(first_b, hole)], np.int32).reshape(-1)
    rows_f = st_f[:, 3] if nf else np.zeros(0, np.int32)
    rows_b = np.where(hole, st_b[:, 3] + 2, 0) if nb else np.zeros(0, np.int32)
    offs = np.r_[0, np.cumsum(np.r_[rows_f, rows_b])].astype(np.int32)
    off_f, off_b = offs[:nf].copy(), offs[nf: nf + nb].copy()
    total = int(offs[-1])
    row_lo = np.full(max(total, 1), 0x7fffffff, np.int32)
    row_hi = np.full(max(total, 1), -1, np.int32)
    p64 = prob.astype(np.float64)
    sum_f = np.bincount(flat_f, weights=p64.ravel(), minlength=nf + 1)[1:]
    sum_b = np.bincount(flat_b, weights=p64.ravel(), minlength=nb + 1)[1:] * hole
    ys, xs = np.nonzero(lab_f)
    for y, x in zip(ys, xs):
        l = lab_f[y, x] - 1
        r = off_f[l] + y - st_f[l, 1]
        row_lo[r] = min(row_lo[r], x)
        row_hi[r] = max(row_hi[r], x)
    ring_sum = np.zeros(nb, np.float64)
    ring_cnt = np.zeros(nb, np.int32)
    for y, x in zip(ys, xs):
        seen = set()
        for dy, dx in ((0, -1), (0, 1), (-1, 0), (1, 0)):
            yy, xx = y + dy, x + dx
            if not (0 <= yy < H and 0 <= xx < W):
                continue
            hb = lab_b[yy, xx]
            if hb <= 0 or par_b[hb - 1] != lab_f[y, x] or hb in seen:
                continue
            seen.add(hb)
            ring_sum[hb - 1] += p64[y, x]
            ring_cnt[hb - 1] += 1
            r = off_b[hb - 1] + y - (st_b[hb - 1, 1] - 1)
            row_lo[r] = min(row_lo[r], x)
            row_hi[r] = max(row_hi[r], x)
    return dict(W=W, H=H, n_f=nf, st_f=st_f, first_f=first_f, par_f=par_f, off_f=off_f, sum_f=sum_f.astype(np.float64),
                n_b=nb, st_b=st_b, first_b=first_b, par_b=par_b, off_b=off_b, sum_b=sum_b.astype(np.float64),
                ring_sum=ring_sum, ring_cnt=ring_cnt, row_lo=row_lo, row_hi=row_hi, lab_f=lab_f, lab_b=lab_b)


def boxes_from_tables(pkg, t, max_candidates=1000, unclip_ratio=1.5):
    L = pkg._lib
    lib = L.lib()
    boxes = np.zeros((max(max_candidates, 1), 4, 2), np.int16)
    scores = np.zeros((max(max_candidates, 1),), np.float32)
    n = C.c_int32(0)
    keep = {k: np.ascontiguousarray(t[k]) for k in ("st_f", "first_f", "par_f", "off_f", "sum_f", "st_b", "first_b", "par_b",
                                                     "off_b", "sum_b", "ring_sum", "ring_cnt", "row_lo", "row_hi")}
    ptr = lambda k: keep[k].ctypes.data if keep[k].size else None
    L.check(lib.ctd_db_boxes_compact(t["W"], t["H"], t["n_f"], ptr("st_f"), ptr("first_f"), ptr("par_f"), ptr("off_f"),
                                     ptr("sum_f"), t["n_b"], ptr("st_b"), ptr("first_b"), ptr("par_b"), ptr("off_b"),
                                     ptr("sum_b"), ptr("ring_sum"), ptr("ring_cnt"), keep["row_lo"].ctypes.data,
                                     keep["row_hi"].ctypes.data, max_candidates, unclip_ratio, boxes.ctypes.data,
                                     scores.ctypes.data, C.byref(n)), "ctd_db_boxes_compact")
    return boxes[: n.value], scores[: n.value]
