"""Loader of the product package for tests (directory name has a hyphen)."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
cg = importlib.import_module("collaborative-circom_amd")


def ensure_built():
    if not os.path.exists(cg.LIB_PATH):
        cg.build()
    return cg
