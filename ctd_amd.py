"""Alias: `import ctd_amd` (and `import ctd_amd.detector`, `from ctd_amd.backend import ...`) == the
hyphen-named package `comic-text-detector_amd/`.  Sub-modules resolve to the SAME module objects as
under the real name (one `TextBlock` class, one library handle), not to second copies."""
import importlib
import importlib.abc
import importlib.util
import sys

_REAL = "comic-text-detector_amd"


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname == __name__ or fullname.startswith(__name__ + "."):
            return importlib.util.spec_from_loader(fullname, self)
        return None

    def create_module(self, spec):
        return importlib.import_module(_REAL + spec.name[len(__name__):])

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, _AliasFinder())
sys.modules[__name__] = importlib.import_module(_REAL)
