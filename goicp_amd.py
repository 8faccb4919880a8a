"""Importable alias of the package directory `go-icp-protein-cavities_amd/` (whose name is not an identifier)."""
import importlib
import sys

_pkg = importlib.import_module("go-icp-protein-cavities_amd")
sys.modules[__name__] = _pkg
