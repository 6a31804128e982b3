import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


# ---- dispatch check of a whole pytest session ------------------------------------------------------------------------
# tests/test_switches_gpu.py (and the other suites that re-run parity sets in a child process with an A/B switch set) name the
# kernels the child MUST / MUST NOT launch: INTEL_EXPECT_KERNELS / INTEL_FORBID_KERNELS (comma-separated base names).  The
# child then runs every GPU test under the library's built-in profiler (csrc/prof.cpp), collects the kernel names test by
# test and fails the session when an expected kernel never ran or a forbidden one did -- a switch that silently stopped
# selecting its code path turns the child red instead of re-testing the default path.
_EXPECT = [k for k in os.environ.get('INTEL_EXPECT_KERNELS', '').split(',') if k]
_FORBID = [k for k in os.environ.get('INTEL_FORBID_KERNELS', '').split(',') if k]
_SEEN = set()


@pytest.fixture(autouse=True)
def _dispatch_trace(request):
    if not (_EXPECT or _FORBID) or request.node.get_closest_marker('gpu') is None:
        yield
        return
    from intel_sigir2023_amd import _lib
    from tests.helpers import kernel_base
    lib = _lib.lib()
    lib.intel_prof_timeline()
    lib.intel_prof_enable(1)
    yield
    import torch
    torch.cuda.synchronize()
    recs = json.loads(lib.intel_prof_timeline().decode())
    lib.intel_prof_enable(0)
    _SEEN.update(kernel_base(r['name']) for r in recs)


def pytest_sessionfinish(session, exitstatus):
    if not (_EXPECT or _FORBID):
        return
    if not _SEEN:
        # nothing was recorded: no gpu-marked test ran in this child, or the profiler saw no launch -- exactly the silent case the check exists for
        if _EXPECT:
            print('\nDISPATCH CHECK FAILED: expected kernels %s but the session recorded no launch at all' % _EXPECT)
            session.exitstatus = 1
        return
    missing = [k for k in _EXPECT if k not in _SEEN]
    extra = [k for k in _FORBID if k in _SEEN]
    if missing or extra:
        print('\nDISPATCH CHECK FAILED: expected kernels that never ran %s; forbidden kernels that ran %s; ran: %s' % (missing, extra, sorted(_SEEN)))
        session.exitstatus = 1
