import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# the scans show the reference's tqdm bar by default; the suite keeps its logs clean (tests of the bar pass progress=True)
os.environ.setdefault("CELLREGMAP_AMD_PROGRESS", "0")

# A test that takes the process down (a GPU memory-access fault makes ROCr call abort(), a C++ exception
# crossing the C-ABI ends in std::terminate) must leave its name behind: every test writes its nodeid to
# stderr before it starts and to a trace file that survives the process (CRM_TEST_TRACE, default
# gpurun_out/pytest_trace.log when that directory exists).
_trace_fh = None
_echo = True   # nodeids on stderr: always, except in the CPU-only selection (-m "not gpu"), which cannot take a GPU fault


def _trace_path():
    p = os.environ.get("CRM_TEST_TRACE")
    if p:
        return p
    d = os.path.join(ROOT, "gpurun_out")
    return os.path.join(d, "pytest_trace.log") if os.path.isdir(d) else None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    global _trace_fh, _echo
    _echo = "not gpu" not in (config.getoption("markexpr", "") or "") or bool(os.environ.get("CRM_TEST_TRACE"))
    p = _trace_path()
    if p:
        try:
            _trace_fh = open(p, "a", buffering=1)
            _trace_fh.write("=== session pid %d argv %s\n" % (os.getpid(), " ".join(sys.argv[1:])))
        except OSError:
            _trace_fh = None


def pytest_runtest_logstart(nodeid, location):
    msg = "[crm-test] start %s\n" % nodeid
    if _echo:
        try:
            os.write(2, msg.encode())
        except OSError:
            pass
    if _trace_fh is not None:
        _trace_fh.write(msg)
        _trace_fh.flush()
        os.fsync(_trace_fh.fileno())


def pytest_runtest_logfinish(nodeid, location):
    if _trace_fh is not None:
        _trace_fh.write("[crm-test] done  %s\n" % nodeid)
        _trace_fh.flush()


@pytest.fixture(scope="session")
def math_golden():
    import numpy as np

    return np.load(os.path.join(GOLDEN, "math_golden.npz"))


def pytest_sessionfinish(session, exitstatus):
    """Under CRM_POISON=1 every device buffer carries a red zone that is inspected when the buffer is released: a session
    in which some kernel wrote past the end of its buffer fails here even if every assertion held."""
    if os.environ.get("CRM_POISON", "0") not in ("", "0") and "cellregmap_amd._lib" in sys.modules:
        lib = sys.modules["cellregmap_amd._lib"]._lib
        if lib is not None:
            import gc

            eng = sys.modules.get("cellregmap_amd._engine")
            if eng is not None:
                eng._bg_cache.clear()
            gc.collect()      # release what the tests left behind, so that their red zones are looked at too
            if eng is not None:
                for h in eng._contexts.values():     # the contexts' own work buffers stay alive: inspect them in place
                    lib.crm_test_check_context(h)
            bad = lib.crm_test_overruns() - int(os.environ.get("CRM_TEST_EXPECTED_OVERRUNS", "0"))   # (the detector's self-test)
            sys.stderr.write("[crm-test] CRM_POISON: %d device buffer overrun(s) detected\n" % bad)
            if bad:
                session.exitstatus = 3


def pytest_unconfigure(config):
    global _trace_fh
    if _trace_fh is not None:
        _trace_fh.close()
        _trace_fh = None


@pytest.fixture
def kernel_form():
    """Switch a kernel form of the library for the duration of a test (include/crm_hip_test.h: crm_test_set_form):
    ``kernel_form("gram_staged", 1)``; every form set through the fixture returns to its default afterwards."""
    from cellregmap_amd import _lib

    lib = _lib.load()
    touched = []

    def set_form(name, value=1, reset=False):
        _lib.check(lib.crm_test_set_form(name.encode(), int(value), 1 if reset else 0))
        touched.append(name)

    yield set_form
    for name in touched:
        lib.crm_test_set_form(name.encode(), 0, 1)
