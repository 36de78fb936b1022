import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "variant: opt-in protocol variants that are NOT the reference's message sequence (CGH_SESSION_ADDITIVE_H); they need a GPU and run "
                                       "only when asked for: pytest -m variant")


def pytest_collection_modifyitems(config, items):
    """`variant` tests stay out of both default runs (-m gpu on the GPU box, -m "not gpu" on CPU): selected only by an -m expression that names them"""
    if "variant" in (config.getoption("-m") or ""):
        return
    keep, drop = [], []
    for it in items:
        (drop if it.get_closest_marker("variant") else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


GOLDEN = os.path.join(HERE, "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def host_option():
    """set(option, value): a process-wide option of the host library (cgh_set_option) for the duration of one test"""
    from product import cg
    saved = {}

    def set_(option, value):
        if option not in saved:
            saved[option] = cg.host_get_option(option)
        cg.host_set_option(option, value)
    yield set_
    for k, v in saved.items():
        cg.host_set_option(k, v)


@pytest.fixture
def lib_option():
    """set(option, value): a process-wide option of the hip library (cg_set_option) for the duration of one test"""
    from product import cg
    saved = {}

    def set_(option, value):
        if option not in saved:
            saved[option] = cg.get_option(option)
        cg.set_option(option, value)
    yield set_
    for k, v in saved.items():
        cg.set_option(k, v)
