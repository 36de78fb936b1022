"""Loader of the product package for tests (directory name has a hyphen)."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
cg = importlib.import_module("collaborative-circom_amd")


_made = False


def ensure_built():
    """ALWAYS runs make (a no-op when the in-tree libraries are newer than every source): a stale .so that travelled to the GPU box
    would otherwise be tested without complaint.  Once per process."""
    global _made
    if not _made:
        cg.build()
        _made = True
    return cg
