"""The window-local refine kernel's algorithm (tests/twlds_emul.py = csrc/kernels_twlds.hip word for word) against the
oracle's pixel-level merge_mask_list (reference utils/textmask.py:73-132): bit-exact on speckle, strokes, blobs, windows
narrower than a word, exactly one word wide, one pixel wide / high, and with components of 1 / 2 pixels in every position
(the `w * h < 3` rule as a bit pattern, also across word boundaries)."""
import numpy as np
import pytest

from oracle import postproc_ref as R
import twlds_emul as E


def smooth_mask(rng, H, W):
    """a 0..255 prediction with blobs (values on both sides of 60)"""
    m = np.zeros((H, W), np.float32)
    for _ in range(max(1, H * W // 400)):
        cy, cx, r = rng.integers(0, H), rng.integers(0, W), rng.integers(2, 9)
        yy, xx = np.ogrid[:H, :W]
        m += 200 * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2.0 * r * r))
    return np.clip(m + rng.integers(0, 70, (H, W)), 0, 255).astype(np.uint8)


def candidates(rng, H, W, kind):
    out = []
    for k in range(int(rng.integers(1, 5))):
        if kind == "speckle":
            c = rng.random((H, W)) < rng.choice([0.05, 0.3, 0.5, 0.7])
        elif kind == "strokes":
            c = np.zeros((H, W), bool)
            for _ in range(max(1, H * W // 150)):
                y, x = rng.integers(0, H), rng.integers(0, W)
                if rng.random() < 0.5:
                    c[y, x: x + rng.integers(1, 12)] = True
                else:
                    c[y: y + rng.integers(1, 12), x] = True
        else:
            c = smooth_mask(rng, H, W) > rng.integers(40, 200)
        out.append([(c * 255).astype(np.uint8), int(rng.integers(0, 1000))])
    return out


SHAPES = [(1, 1), (1, 40), (40, 1), (5, 3), (7, 31), (9, 32), (12, 33), (20, 64), (17, 65), (30, 100), (41, 97), (64, 64)]


@pytest.mark.parametrize("kind", ["speckle", "strokes", "blobs"])
@pytest.mark.parametrize("mode", [0, 1])
def test_bit_plane_merge_equals_the_oracle(kind, mode):
    rng = np.random.default_rng(hash((kind, mode)) % 2 ** 31)
    for H, W in SHAPES:
        for rep in range(2):
            pm = smooth_mask(rng, H, W)
            ml = candidates(rng, H, W, kind)
            want = R.merge_mask_list([[c.copy(), s] for c, s in ml], pm.copy(), refine_mode=mode)
            order = sorted(range(len(ml)), key=lambda i: ml[i][1])           # list.sort is stable, like this
            got, _ = E.merge_mask_list([ml[i][0] for i in order], pm, refine_mode=mode)
            np.testing.assert_array_equal(got, want, err_msg=f"{kind} {H}x{W} mode {mode}")


def test_small_component_rule_in_every_position_of_a_word_boundary():
    """1x1, 2x1, 1x2 components are skipped (w * h < 3), diagonal pairs (2x2 box) and triples are not -- around bit 31 | 0"""
    H, W = 9, 70
    pm = np.full((H, W), 255, np.uint8)                           # everything predicted text: every allowed component is accepted
    for x0 in (0, 29, 30, 31, 32, 33, 62, 67, 68, 69):
        for shape in ("dot", "h2", "v2", "diag", "antidiag", "h3", "v3", "L"):
            c = np.zeros((H, W), np.uint8)
            pts = {"dot": [(4, 0)], "h2": [(4, 0), (4, 1)], "v2": [(4, 0), (5, 0)], "diag": [(4, 0), (5, 1)],
                   "antidiag": [(5, 0), (4, 1)], "h3": [(4, 0), (4, 1), (4, 2)], "v3": [(3, 0), (4, 0), (5, 0)],
                   "L": [(4, 0), (5, 0), (5, 1)]}[shape]
            for y, dx in pts:
                if x0 + dx < W:
                    c[y, x0 + dx] = 255
            want = R.merge_mask_list([[c.copy(), 0]], pm.copy(), refine_mode=1)
            got, _ = E.merge_mask_list([c], pm, refine_mode=1)
            np.testing.assert_array_equal(got, want, err_msg=f"{shape} at x={x0}")
