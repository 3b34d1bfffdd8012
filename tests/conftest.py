import os
import sys
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _baseline_modes():
    """Every test starts from the fp32-storage trunk and fp32 GEMMs (round 1's kernels: the reference point the parity tests of every other
    mode were written against) and selects the mode it is about explicitly; the PACKAGE default -- split precision, _lib.DEFAULT_MODE -- is
    what tests/test_host_cpu.py::test_package_default_is_the_split_precision_mode pins, in a fresh interpreter."""
    try:
        from aesrc2020_amd import _lib
        _lib.lib()
    except Exception:                                                # no library built (collection on a bare checkout): nothing to reset
        yield
        return
    _lib.set_trunk_dtype("fp32"); _lib.set_gemm_math("fp32"); _lib.set_bn_deterministic(False)
    yield
    _lib.set_trunk_dtype("fp32"); _lib.set_gemm_math("fp32"); _lib.set_bn_deterministic(False)
