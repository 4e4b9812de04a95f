import os
import sys

import pytest

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for path in (REPO_ROOT, os.path.join(REPO_ROOT, "tests")):
    if path not in sys.path:
        sys.path.insert(0, path)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The library reads its MDB_* switches once per process; the tests set them per test (monkeypatch.setenv): every
    # call through the binding asks the library to read the environment again first.
    from modelardb_rs_amd import _abi
    _abi.RELOAD_OPTIONS_BEFORE_EVERY_CALL = True


@pytest.fixture(scope="session")
def hip():
    """The HIP product library bound to cuda:0. Fails loudly when it is missing: no fallback."""
    import modelardb_rs_amd as mdb
    from modelardb_rs_amd import api
    if os.environ.get("MDB_HOST_LIBRARY_UNDER_TEST"):
        # tests/test_host_sanitizers_cpu.py: the host operators over the canned-answer stand-in for
        # libmdb_hip (tests/stub/mdb_stub.cpp); the only thing they need of a context is its handle.
        import ctypes
        from modelardb_rs_amd import host

        class StubContext:
            handle = ctypes.c_void_p()

        assert host.lib().mdb_init(0, ctypes.byref(StubContext.handle)) == 0
        return StubContext()
    return api.Context(0)
