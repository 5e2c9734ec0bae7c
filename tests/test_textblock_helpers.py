"""`TextBlock`'s numpy helpers (min_rect, bounding_rect, aspect_ratio, alignment, text / colour accessors) against the
reference's OWN class (utils/textblock.py:110-265) in the build container, and against golden values of it
(tests/golden/textblock_helpers.json, written by this file's `--regen`) everywhere else."""
import importlib
import json
import os
import sys
import warnings

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "textblock_helpers.json")
sys.path.insert(0, os.path.dirname(HERE))


def pkg():
    return importlib.import_module("comic-text-detector_amd")


def cases():
    rng = np.random.RandomState(11)
    out = []
    for k in range(24):
        n = int(rng.randint(1, 6)) if k else 0
        angle = int(rng.choice([0, 0, 7, -12, 33, 90, -45]))
        x0, y0 = rng.randint(10, 600, size=2)
        lines = []
        for i in range(n):
            w, h = rng.randint(20, 200), rng.randint(8, 40)
            x, y = x0 + rng.randint(-15, 15), y0 + i * 45
            lines.append([[int(x), int(y)], [int(x + w), int(y + rng.randint(-3, 3))], [int(x + w), int(y + h)], [int(x), int(y + h)]])
        xy = np.array(lines).reshape(-1, 2) if n else np.array([[x0, y0], [x0 + 50, y0 + 30]])
        out.append(dict(xyxy=[int(xy[:, 0].min()), int(xy[:, 1].min()), int(xy[:, 0].max()), int(xy[:, 1].max())], lines=lines,
                        angle=angle, vertical=bool(k % 5 == 3), alignment=int(rng.choice([-1, -1, -1, 0, 2])),
                        text=["ab", " cd "] if k % 2 else "whole string", frgb=[int(v) for v in rng.randint(0, 255, 3)],
                        srgb=[int(v) for v in rng.randint(0, 255, 3)], accumulate=bool(k % 3)))
    return out


def evaluate(cls, c):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        b = cls(c["xyxy"], lines=[list(map(list, ln)) for ln in c["lines"]], angle=c["angle"], vertical=c["vertical"],
                alignment=c["alignment"], text=c["text"])
        r = {}
        if c["lines"]:
            r["min_rect"] = np.asarray(b.min_rect()).tolist()
            r["min_rect_noback"] = np.asarray(b.min_rect(rotate_back=False)).tolist()
            r["bounding_rect"] = [int(v) for v in b.bounding_rect()]
            ar = float(b.aspect_ratio())
            r["aspect_ratio"] = None if not np.isfinite(ar) else ar
        r["alignment"] = int(b.alignment())
        r["text"] = b.get_text()
        b.set_font_colors(c["frgb"], c["srgb"], accumulate=c["accumulate"])
        r["stored"] = [int(v) for v in (b.fg_r, b.fg_g, b.fg_b, b.bg_r, b.bg_g, b.bg_b)]
        for bgr in (False, True):
            f, s = b.get_font_colors(bgr=bgr)
            r[f"colors_{int(bgr)}"] = [[int(v) for v in f], [int(v) for v in s]]
        r["stroke_width"] = float(b.stroke_width)
        r["len"] = len(b)
        r["xywh"] = [int(v) for v in b.xywh()]
    return r


def reference_class():
    try:
        from oracle import ref_post_import as RP
        return RP.load_reference_post().TB.TextBlock
    except Exception:
        return None


def test_helpers_match_golden_values_of_the_reference_class():
    want = json.load(open(GOLD))
    T = pkg().textblock.TextBlock
    got = [evaluate(T, c) for c in cases()]
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g == w


def test_helpers_match_the_reference_class_itself():
    R = reference_class()
    if R is None:
        pytest.skip("/root/reference is not importable here (GPU box): covered by the golden values")
    T = pkg().textblock.TextBlock
    for c in cases():
        assert evaluate(T, c) == evaluate(R, c)


def test_bounding_rect_prefers_a_stored_rectangle_and_target_lang_reads_the_field():
    T = pkg().textblock.TextBlock
    b = T([0, 0, 10, 10], lines=[[[0, 0], [10, 0], [10, 10], [0, 10]]], _bounding_rect=[1, 2, 3, 4], target_lang="en")
    assert b.bounding_rect() == [1, 2, 3, 4] and b.target_lang() == "en"


if __name__ == "__main__" and "--regen" in sys.argv:
    R = reference_class()
    assert R is not None, "needs /root/reference"
    json.dump([evaluate(R, c) for c in cases()], open(GOLD, "w"), indent=0)
    print("wrote", GOLD)


def _random_records(rng, n_pages, max_blocks):
    """Batch-wide native records as `ctd_tail_batch_fetch` lays them out: pages back to back, per-page offsets inside."""
    TB = pkg().textblock
    counts, recs, lines, dist = [], [], [], []
    for _ in range(n_pages):
        nb = int(rng.randint(0, max_blocks + 1))
        r = np.zeros((nb,), TB.BLK_DTYPE)
        nl = rng.randint(0, 4, nb)
        nd = nl.copy()
        r["xyxy"] = rng.randint(0, 2000, (nb, 4))
        r["language"] = rng.randint(0, 3, nb)
        r["vertical"] = rng.randint(0, 2, nb)
        r["angle"] = rng.randint(-90, 91, nb)
        r["font_is_float"] = rng.randint(0, 2, nb)
        r["font_size"] = rng.rand(nb) * 60 - 5                       # int(font_size) truncates toward zero, also below 0
        r["vec"] = rng.randn(nb, 2)
        r["norm"] = rng.rand(nb) * 100
        r["weight"] = rng.rand(nb)
        r["merged"] = rng.randint(0, 2, nb)
        r["n_lines"], r["n_dist"] = nl, nd
        r["line_off"] = np.concatenate(([0], np.cumsum(nl)))[:-1] if nb else 0
        r["dist_off"] = np.concatenate(([0], np.cumsum(nd)))[:-1] if nb else 0
        counts.append((nb, int(nl.sum()), int(nd.sum())))
        recs.append(r)
        lines.append(rng.randint(-5, 3000, (int(nl.sum()), 8)).astype(np.int32))
        d = rng.rand(int(nd.sum()), 3) * 2.4 - 1.2                   # cosines beyond [-1, 1]: arccos gives nan, like numpy's
        dist.append(d)
    cat = lambda xs, shape, dt: np.concatenate(xs) if sum(len(x) for x in xs) else np.zeros(shape, dt)   # noqa: E731
    return (np.concatenate(recs) if sum(len(r) for r in recs) else np.zeros((0,), TB.BLK_DTYPE),
            cat(lines, (0, 8), np.int32), cat(dist, (0, 3), np.float64), np.array(counts, np.int64).reshape(-1, 3))


def _same_objects(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        dx, dy = vars(x), vars(y)
        assert list(dx) == list(dy)                                  # attribute order = the key order of to_dict()
        for k in dx:
            assert type(dx[k]) is type(dy[k]), (k, type(dx[k]), type(dy[k]))
            if isinstance(dx[k], np.ndarray):
                assert dx[k].dtype == dy[k].dtype and np.array_equal(dx[k], dy[k], equal_nan=True), k
            else:
                assert dx[k] == dy[k] or (dx[k] != dx[k] and dy[k] != dy[k]), k


def test_textblocks_built_in_c_equal_the_python_loop():
    """csrc/pyblocks.c (`_ctd_pyblocks.build_blocks`: one C loop over the native records' columns) against the Python loop it
    replaces on the tail workers -- same attribute order, same Python types (bool / int / float / list / np.float64 /
    ndarray), same values incl. `int(font_size)`, the nan distances and fresh `text` lists; and the batch-wide conversion
    (`blocks_from_batch`: page-relative offsets shifted into the batch's pools) against page-by-page conversion."""
    TB = pkg().textblock
    if TB._PYB is None:                                              # a checkout that has not been built yet
        pkg()._lib.build()
        TB = importlib.reload(TB)
    assert TB._PYB is not None, "csrc/pyblocks.c was not built (make -C comic-text-detector_amd/csrc)"
    rng = np.random.RandomState(3)
    for n_pages, max_blocks in ((1, 0), (1, 7), (5, 40), (9, 3)):
        recs, lines, dist, counts = _random_records(rng, n_pages, max_blocks)
        ob = np.concatenate(([0], np.cumsum(counts[:, 0])))
        ol = np.concatenate(([0], np.cumsum(counts[:, 1])))
        od = np.concatenate(([0], np.cumsum(counts[:, 2])))
        per_page = [(recs[ob[b]: ob[b + 1]], lines[ol[b]: ol[b + 1]], dist[od[b]: od[b + 1]]) for b in range(n_pages)]
        py = [TB.blocks_from_records(*p, native=False) for p in per_page]
        cc = [TB.blocks_from_records(*p, native=True) for p in per_page]
        batch = TB.blocks_from_batch(recs, lines, dist, counts)
        assert len(batch) == n_pages
        for b in range(n_pages):
            _same_objects(cc[b], py[b])
            _same_objects(batch[b], py[b])
            assert all(t.text == [] for t in batch[b]) and len({id(t.text) for t in batch[b]}) == len(batch[b])
            assert isinstance(batch[b], list)
        for blk in (x for pg in batch for x in pg):                  # the objects are ordinary TextBlocks
            assert isinstance(blk, TB.TextBlock) and isinstance(blk.to_dict(), dict)
