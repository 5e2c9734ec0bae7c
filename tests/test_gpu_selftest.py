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
