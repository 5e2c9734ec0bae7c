"""The C-ABI shared library loads and exports every symbol include/ctd_hip.h
declares; the ctypes struct mirrors have the C layout (no compute calls here)."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import pytest

from conftest import ROOT, pkg

HEADER = os.path.join(ROOT, "include", "ctd_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ctd_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = pkg()._lib
    if not os.path.isfile(L.LIB_PATH):
        L.build()
    lib = L.lib()
    names = declared_functions()
    assert len(names) >= 16
    for n in names:
        assert hasattr(lib, n), f"{n} declared in ctd_hip.h but not exported"
    assert sorted(L.SYMBOLS) == names, "python binding table and header disagree"
    assert lib.ctd_abi_version() == L.ABI_VERSION


def test_struct_layout_matches_c():
    L = pkg()._lib
    prog = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "ctd_hip.h"
    int main(void){ printf("%zu %zu %zu %zu %zu %zu\n", sizeof(ctd_tensor), sizeof(ctd_op),
        offsetof(ctd_op, w_off), offsetof(ctd_op, b_off), offsetof(ctd_op, aux), offsetof(ctd_op, faux)); return 0; }
    '''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(prog)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        vals = [int(v) for v in subprocess.check_output([exe]).split()]
    assert vals == [C.sizeof(L.CtdTensor), C.sizeof(L.CtdOp), L.CtdOp.w_off.offset, L.CtdOp.b_off.offset,
                    L.CtdOp.aux.offset, L.CtdOp.faux.offset]


def test_engine_create_rejects_bad_program_without_gpu_compute():
    """Error convention: negative rc + message, no exception across the ABI."""
    L = pkg()._lib
    lib = L.lib()
    h = C.c_void_p()
    rc = lib.ctd_engine_create(C.byref(h), None, 0, None, 0, None, 0, L.PREC_F16, 0)
    assert rc == -1 and b"null" in lib.ctd_last_error()


def test_backend_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = pkg()
    with pytest.raises(p._lib.CtdError):
        p.backend.HipTextDetBackend(p.synth.make_checkpoint(0))
