"""-m gpu: the standalone kernel self-test binary (csrc/selftest.hip) for the multi-layer kernels and the head tails
on the MFMA -- each compared with the kernels it replaces: the C3 block / stem + layer 1 / SPPF pools BIT FOR BIT with
the per-layer launches (134 M values per shape at B=32; here B=4), `db_up_mfma_kernel` / `seg_final_mfma_kernel` within
3e-3 / 2e-5 of their VALU versions.  Same binary, larger batch: profiles/r02_selftest_fused_b32.txt."""
import os
import subprocess

import pytest

from conftest import pkg

pytestmark = pytest.mark.gpu


def test_selftest_of_the_fused_and_mfma_head_tail_kernels():
    L = pkg()._lib
    assert os.path.isfile(L.SELFTEST_PATH), "ctd_selftest is built by __graft_entry__.build()"
    r = subprocess.run([L.SELFTEST_PATH, "4"], env={**os.environ, "ST_ONLY_C3": "1"}, capture_output=True, text=True,
                       timeout=600)
    out = r.stdout
    assert r.returncode == 0 and "selftest: PASSED (0 failures)" in out, out[-4000:]
    lines = out.splitlines()
    exact = [ln for ln in lines if ln.startswith(("[c3]", "[stem2]", "[sppf]"))]
    assert len(exact) >= 12
    for ln in exact:
        assert "ok(bit-exact)" in ln or "unsupported (falls back)" in ln, ln
    tol = [ln for ln in lines if ln.startswith(("[dbup]", "[segfinal]"))]
    assert len(tol) >= 6 and all("| ok:" in ln for ln in tol), "\n".join(tol)


def test_selftest_of_the_split_operand_kernels():
    """`ST_SPLIT=1`: the fp32s engine's kernels (kernels_split*.hip) against the f32-operand MFMA kernel and a float64 host
    reference -- fp32 tensors, split-plane tensors on the 128-pixel and on the haloed-patch kernel, mixed layouts, ragged
    maps; the first layer straight from the page against INPUT + the generic kernel.  B = 32 lines: profiles/r03_split_selftest.txt."""
    L = pkg()._lib
    r = subprocess.run([L.SELFTEST_PATH, "2"], env={**os.environ, "ST_SPLIT": "1", "ST_CASES": "3,10,16,18"}, capture_output=True,
                       text=True, timeout=600)
    out = r.stdout
    assert r.returncode == 0 and "selftest: PASSED (0 failures)" in out, out[-4000:]
    lines = out.splitlines()
    sp = [ln for ln in lines if ln.startswith("[split]")]
    assert len(sp) >= 12 and not any("FAIL" in ln for ln in sp), "\n".join(sp)
    assert sum("split(planes,halo): ok" in ln for ln in sp) >= 5           # ragged halo / PAIR cases + the selected network shapes
    assert sum("f32->planes ok, planes->f32 ok" in ln for ln in sp) >= 8
    st = [ln for ln in lines if ln.startswith("[stem-split]")]
    assert len(st) == 2 and all(ln.count(": ok") == 3 for ln in st), "\n".join(st)


def test_selftest_of_the_big_tile_convt_kernels():
    """ConvTranspose 512 -> 256, 256 -> 128, 128 -> 64 at B = 16 (1024 ... 8192 tiles): the product dispatch (kernels_halo3.hip)
    against the exact direct kernel, bit for bit against kernels_halo2.hip and kernels_halo.hip, and six repeated launches
    with identical bits (the LDS hand-offs of both big-tile kernels are ordered by counted waits and barriers only)."""
    L = pkg()._lib
    r = subprocess.run([L.SELFTEST_PATH, "16"], env={**os.environ, "ST_CASES": "16,17,18", "ST_NO_C3": "1"}, capture_output=True,
                       text=True, timeout=600)
    out = r.stdout
    assert r.returncode == 0 and "selftest: PASSED (0 failures)" in out, out[-4000:]
    cases = [ln for ln in out.splitlines() if ln.startswith("[case] convT4")]
    assert len(cases) == 3
    for ln in cases:
        assert "[x6 repeat: stable]" in ln and "default: ok" in ln, ln
        assert "[default vs halo2: bit-identical]" in ln and "[default vs halo1: bit-identical]" in ln, ln
