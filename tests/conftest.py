import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


# Several test modules (re)build the same harness binaries and the host library with `make` / `g++` from their fixtures.  Under pytest-xdist two
# workers doing that at once relink a binary a third is executing ("Text file busy", "Permission denied"): right after a rebuild of the tree the
# first run failed two tests that way and the second passed.  The builds of a session — all workers — therefore take turns on a lock file.
import fcntl
import subprocess

_real_run = subprocess.run


def _run_builds_in_turn(cmd, *args, **kwargs):
    if isinstance(cmd, (list, tuple)) and cmd and os.path.basename(str(cmd[0])) in ("make", "g++", "gcc", "hipcc"):
        with open(os.path.join(ROOT, "tests", "harness", ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            return _real_run(cmd, *args, **kwargs)
    return _real_run(cmd, *args, **kwargs)


subprocess.run = _run_builds_in_turn


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except OSError:
        return False


HAS_GPU = _has_gpu()

# On a GPU box the suite runs with the library's guarded device allocations (include/pandepth_amd_dev.h: pd_guard_check): every
# buffer the engine allocates is surrounded by canaries, the executable exits with 97 when one of its kernels wrote outside a
# buffer, and every gpu test is followed by a check of the buffers still alive.  PANDEPTH_GUARD=0 switches it off.
if HAS_GPU:
    os.environ.setdefault("PANDEPTH_GUARD", "1")


def pytest_collection_modifyitems(config, items):
    # A gpu-marked test on a box without a GPU is a hard error on purpose when selected with
    # -m gpu (no silent skips on the GPU box); in an unfiltered run here it is skipped.
    if HAS_GPU:
        return
    if "gpu" in (config.getoption("-m") or "") and "not gpu" not in (config.getoption("-m") or ""):
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(autouse=True)
def _no_out_of_bounds_device_writes(request):
    yield
    if not HAS_GPU or "gpu" not in request.keywords or os.environ.get("PANDEPTH_GUARD", "0") in ("", "0"):
        return
    import ctypes
    import pandepth_amd as pda
    L = pda.load()
    msg = ctypes.create_string_buffer(400)
    before = getattr(_no_out_of_bounds_device_writes, "seen", 0)
    n = L.pd_guard_check(msg, 400)
    _no_out_of_bounds_device_writes.seen = n
    assert n == before, "a kernel wrote outside a device buffer: " + msg.value.decode()


@pytest.fixture(scope="session", autouse=True)
def _oracle_backed_cli():
    """The product's host code on the oracle-backed engine (tests/harness/pandepth_oracle_cli) is what most CPU tests execute; several
    of them take it for granted.  It is made here once per session — per pytest-xdist worker, in turn on the build lock — so that no
    test depends on another one's fixture having run first (a fresh tree under `-n 4` failed the fuzz test that way)."""
    if not HAS_GPU and not os.path.exists(os.path.join(ROOT, "tests", "harness", "pandepth_oracle_cli")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "pandepth_amd"), "libpandepth_host.a"], check=False, stdout=subprocess.DEVNULL)
        subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "harness"), "pandepth_oracle_cli"], check=False, stdout=subprocess.DEVNULL)
    yield


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
