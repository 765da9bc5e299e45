"""Import alias: ``import scpose`` == the package in ./spacecraft-pose-estimation_amd/
(a hyphenated directory name cannot be written in an import statement)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("spacecraft-pose-estimation_amd")
sys.modules[__name__] = _pkg
