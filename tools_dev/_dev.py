"""Imported first by the developer scripts: the library honours its SCPOSE_* development switches only when
SCPOSE_DEV=1 is set (csrc/util.cpp), so set it whenever one of them is present in the environment."""
import os

if any(k.startswith("SCPOSE_") and k not in ("SCPOSE_DEV", "SCPOSE_MODEL") for k in os.environ):
    os.environ.setdefault("SCPOSE_DEV", "1")
