import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pkg():
    return importlib.import_module("comic-text-detector_amd")


_CKPT = {}


def checkpoint(seed: int = 0):
    if seed not in _CKPT:
        _CKPT[seed] = pkg().synth.make_checkpoint(seed)
    return _CKPT[seed]


def load_golden(name: str):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def ckpt_checksum(ckpt) -> float:
    s = 0.0
    for sd in (ckpt["blk_det"]["weights"], ckpt["text_seg"], ckpt["text_det"]):
        for k in sorted(sd):
            s += float(sd[k].double().sum())
    return s


@pytest.fixture(scope="session")
def ckpt0():
    return checkpoint(0)


def has_gpu() -> bool:
    import torch
    return torch.cuda.is_available()
