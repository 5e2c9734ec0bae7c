"""Alias: `import ctd_amd` == the hyphen-named package `comic-text-detector_amd/`."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("comic-text-detector_amd")
