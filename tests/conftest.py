import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


try:  # property tests run the same examples on every machine (the long randomised runs are done offline)
    from hypothesis import settings as _hyp_settings

    _hyp_settings.register_profile("repeatable", derandomize=True)
    _hyp_settings.load_profile("repeatable")
except ImportError:  # hypothesis is only needed by the property tests themselves
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure): built on demand from oracle/gpk_oracle.c."""
    from oracle import pyoracle

    pyoracle.build()
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def gpk():
    """The product library; GPU tests fail loudly (never skip) when it cannot run."""
    from geopolars_amd import _abi

    _abi.lib()
    name, cus = _abi.device_info()  # raises GeopolarsHipError without a gfx950
    return _abi
