import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.build()
    return oracle


def _refresh_tuning():
    """The library reads its LLE_* tuning overrides once per process (lle_tuning_refresh reads them again)."""
    try:
        from lle_amd import _capi
        _capi.refresh_tuning()
    except Exception:  # noqa: BLE001  (library not built: the tests that need it fail on their own)
        pass


class _TuningMonkeyPatch(pytest.MonkeyPatch):
    """monkeypatch whose setenv / delenv of an LLE_* name also makes the library re-read its overrides."""

    def setenv(self, name, value, prepend=None):
        super().setenv(name, value, prepend)
        if name.startswith("LLE_"):
            _refresh_tuning()

    def delenv(self, name, raising=True):
        super().delenv(name, raising)
        if name.startswith("LLE_"):
            _refresh_tuning()


@pytest.fixture
def monkeypatch():
    mp = _TuningMonkeyPatch()
    yield mp
    mp.undo()
    _refresh_tuning()
