"""Import shim: the package directory is named `rust-raytracer_amd` (with a hyphen),
which the `import` statement cannot spell. `import rtx_amd` gives the same module."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("rust-raytracer_amd")
sys.modules[__name__] = _pkg
