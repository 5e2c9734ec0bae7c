"""-m gpu: NMS and connected-components kernels vs the oracle restatements."""
import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import postproc_ref as R

pytestmark = pytest.mark.gpu


def random_blks(rng, B, rows, no=7, frac=0.02, size=1024):
    b = np.zeros((B, rows, no), np.float32)
    b[..., 0:2] = rng.uniform(0, size, (B, rows, 2))
    b[..., 2:4] = rng.uniform(8, 300, (B, rows, 2))
    b[..., 4] = np.where(rng.uniform(size=(B, rows)) < frac, rng.uniform(0.4, 1.0, (B, rows)),
                         rng.uniform(0, 0.4, (B, rows)))
    b[..., 5:] = rng.uniform(0, 1, (B, rows, no - 5))
    return b


# candidates per page after the confidence filter: ~30, ~370, ~1240 (in registers, csrc/kernels_post.hip NMS_REG), ~5850
# (beyond 2048: the loop over HBM), ~38600 (beyond max_nms = 30000: the score cut first), none
@pytest.mark.parametrize("rows,frac", [(1008, 0.05), (64512, 0.01), (4032, 0.5), (16128, 0.6), (64512, 1.0), (300, 0.0)])
def test_nms_matches_reference_restatement(rows, frac):
    rng = np.random.RandomState(rows + int(frac * 100))
    blks = random_blks(rng, 3, rows, frac=frac)
    dets, counts = pkg().backend.nms(torch.from_numpy(blks).cuda(), 0.4, 0.35)
    torch.cuda.synchronize()
    ref = R.non_max_suppression(blks, 0.4, 0.35)
    for b in range(3):
        n = int(counts[b])
        assert n == len(ref[b]), (n, len(ref[b]))
        got = dets[b, :n].cpu().numpy()
        # identical float32 arithmetic => identical boxes/scores, same order
        np.testing.assert_array_equal(got, ref[b])


def test_nms_clustered_duplicates_and_ties():
    """Heavily overlapping boxes with equal scores: tie-break = lower row first."""
    rng = np.random.RandomState(0)
    base = random_blks(rng, 1, 64, frac=1.0)
    blks = np.repeat(base, 8, axis=1)             # every box 8 times, identical scores
    blks[..., 0:2] += rng.uniform(-1, 1, blks[..., 0:2].shape).astype(np.float32)
    dets, counts = pkg().backend.nms(torch.from_numpy(blks).cuda(), 0.4, 0.35)
    ref = R.non_max_suppression(blks, 0.4, 0.35)
    n = int(counts[0])
    assert n == len(ref[0])
    np.testing.assert_array_equal(dets[0, :n].cpu().numpy(), ref[0])


def test_nms_rejects_bad_threshold():
    p = pkg()
    with pytest.raises(p._lib.CtdError):
        p.backend.nms(torch.zeros(1, 10, 7, device="cuda"), 1.5, 0.35)


def blobs(rng, h, w, n):
    img = np.zeros((h, w), np.uint8)
    for _ in range(n):
        y, x = rng.randint(0, h), rng.randint(0, w)
        hh, ww = rng.randint(1, 24), rng.randint(1, 24)
        img[y:y + hh, x:x + ww] = rng.randint(1, 256)
    # diagonal touches + thin lines exercise 4- vs 8-connectivity
    for i in range(0, min(h, w) - 1, 7):
        img[i, i] = 255
        img[i + 1, i + 1] = 255
    return img


@pytest.mark.parametrize("conn", [4, 8])
@pytest.mark.parametrize("shape", [(64, 64), (257, 131), (1024, 1024)])
def test_ccl_matches_scipy(conn, shape):
    rng = np.random.RandomState(shape[0] * 10 + conn)
    imgs = np.stack([blobs(rng, *shape, 60), (rng.uniform(size=shape) < 0.45).astype(np.uint8) * 255,
                     np.zeros(shape, np.uint8), np.full(shape, 255, np.uint8)])
    labels, n, stats = pkg().backend.connected_components(torch.from_numpy(imgs).cuda(), 0, conn, max_labels=1 << 17)
    torch.cuda.synchronize()
    for b in range(len(imgs)):
        nref, lref, sref = R.connected_components_with_stats(imgs[b], conn)
        assert int(n[b]) == nref - 1
        np.testing.assert_array_equal(labels[b].cpu().numpy(), lref)
        np.testing.assert_array_equal(stats[b, : nref - 1].cpu().numpy(), sref[1:])


@pytest.mark.parametrize("conn", [4, 8])
@pytest.mark.parametrize("shape", [(200, 333), (512, 512)])
def test_ccl_dense_noise(conn, shape):
    """Noise from sparse to nearly full: long runs across tile borders are where the border kernels link
    only one pixel per pair of overlapping runs."""
    rng = np.random.RandomState(shape[1] + conn)
    imgs = np.stack([(rng.uniform(size=shape) < d).astype(np.uint8) * 255 for d in (0.3, 0.6, 0.75, 0.92, 0.99)])
    imgs[4, ::32] = 0          # full tiles cut apart exactly on the tile rows / columns
    imgs[3, :, 31::32] = 0
    labels, n, stats = pkg().backend.connected_components(torch.from_numpy(imgs).cuda(), 0, conn, max_labels=1 << 17)
    torch.cuda.synchronize()
    for b in range(len(imgs)):
        nref, lref, sref = R.connected_components_with_stats(imgs[b], conn)
        assert int(n[b]) == nref - 1
        np.testing.assert_array_equal(labels[b].cpu().numpy(), lref)
        np.testing.assert_array_equal(stats[b, : nref - 1].cpu().numpy(), sref[1:])


@pytest.mark.parametrize("shape", [(64, 64), (257, 131), (1024, 1024)])
def test_ccl_dual_equals_the_two_separate_labellings(shape):
    """`ctd_ccl_dual`: 8-connected foreground and 4-connected background components in one union-find, label for
    label what scipy gives for the image and for its complement (ids, stats, first pixels, raster order)."""
    rng = np.random.RandomState(shape[1])
    noise = [(rng.uniform(size=shape) < d).astype(np.uint8) * 255 for d in (0.1, 0.45, 0.8)]
    noise[2][::32] = 255
    noise[1][:, 31::32] = 0
    imgs = np.stack([blobs(rng, *shape, 60)] + noise + [np.zeros(shape, np.uint8), np.full(shape, 255, np.uint8)])
    cap = 1 << 17
    labels, (n_f, n_b), (st_f, st_b), (fi_f, fi_b) = pkg().backend.connected_components_dual(
        torch.from_numpy(imgs).cuda(), 0, max_labels=cap)
    torch.cuda.synchronize()
    lab = labels.cpu().numpy()
    for b in range(len(imgs)):
        for sign, img, conn, n, st, fi in ((1, imgs[b], 8, n_f, st_f, fi_f), (-1, np.where(imgs[b] > 0, 0, 255).astype(np.uint8), 4, n_b, st_b, fi_b)):
            nref, lref, sref = R.connected_components_with_stats(img, conn)
            assert int(n[b]) == nref - 1
            np.testing.assert_array_equal(np.maximum(sign * lab[b], 0), lref)
            np.testing.assert_array_equal(st[b, : nref - 1].cpu().numpy(), sref[1:])
            firsts = fi[b, : nref - 1].cpu().numpy()
            flat = lref.ravel()
            assert np.array_equal(flat[firsts], np.arange(1, nref))                     # a pixel of every component ...
            assert all(not (flat[:f] == l + 1).any() for l, f in list(enumerate(firsts))[:50])   # ... and its first one


def test_ccl_threshold_semantics():
    """foreground = img > thresh (reference textmask.py:137: threshold(mask, 30, 255, BINARY) then CC)."""
    rng = np.random.RandomState(5)
    img = rng.randint(0, 256, (96, 160)).astype(np.uint8)
    labels, n, stats = pkg().backend.connected_components(torch.from_numpy(img).cuda(), 30, 4)
    nref, lref, sref = R.connected_components_with_stats((img > 30).astype(np.uint8) * 255, 4)
    assert int(n[0]) == nref - 1
    np.testing.assert_array_equal(labels[0].cpu().numpy(), lref)
