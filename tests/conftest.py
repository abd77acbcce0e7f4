import os
import sys

import pytest

# An abort must name itself in the log of the ordinary run: glibc's fatal messages ("malloc(): corrupted
# top size", "stack smashing detected", "double free") go to /dev/tty unless LIBC_FATAL_STDERR_ is set -- a run whose output
# is redirected to a file loses them; DCTFP_CRASH_BACKTRACE makes libdctfp.so print the native frames of a
# SIGABRT / SIGSEGV before Python's faulthandler prints the Python ones.  Set before anything native is loaded.
os.environ.setdefault('LIBC_FATAL_STDERR_', '1')
os.environ.setdefault('PYTHONFAULTHANDLER', '1')
os.environ.setdefault('DCTFP_CRASH_BACKTRACE', '1')

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)


def _ensure_native_built():
    """The suites import dctdomain_amd, which refuses to load without its HIP library.  Normally
    __graft_entry__.build() has produced it; build it here when a fresh checkout runs pytest first
    (hipcc cross-compiles without a GPU).  The CPU checkers of oracle/ are built the same way."""
    import subprocess
    import build_ext
    if not (os.path.exists(build_ext.LIB_PATH) and os.path.exists(build_ext.RECCUT_LIB_PATH)
            and os.path.exists(build_ext.EXPERIMENTS_LIB_PATH)):
        build_ext.build_all()
    if not os.path.exists(os.path.join(ROOT, 'oracle', '_build', 'liboracle.so')):
        subprocess.run(['make', '-C', os.path.join(ROOT, 'oracle'), 'all'], check=True)


def pytest_configure(config):
    _ensure_native_built()
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_report_header(config):
    """On a GPU box: which HIP / HSA runtime files libdctfp.so (hipcc 7.2) is bound to, their versions, library hashes."""
    if not _gpu_available():
        return None
    try:
        from dctdomain_amd import _lib
        return ['dctfp runtime: ' + line for line in _lib.runtime_report().splitlines()]
    except Exception as e:      # noqa: BLE001 -- a header must not stop the run; the tests will say what is wrong
        return [f'dctfp runtime report failed: {e!r}']


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
