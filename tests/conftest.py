import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session', autouse=True)
def _native_libraries():
    """Build the HIP library / the C oracle when a checkout has not been built yet (hipcc and gcc are in the
    image; on the GPU box the prebuilt files travel with the snapshot and nothing is rebuilt)."""
    from cpflow_amd import build as hip_build
    if not os.path.exists(hip_build.OUT):
        hip_build.build(force=True)
    if not os.path.exists(os.path.join(ROOT, 'oracle', '_build', 'libcpf_oracle.so')):
        import oracle
        oracle.build()


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


class _Tuning:
    """Kernel-selection switches of the library (cpf_set_tuning, include/cpflow_amd.h), restored after the test."""
    def __init__(self):
        self._names = set()

    def set(self, name, value):
        from cpflow_amd import _ffi
        _ffi.set_tuning(name, value)
        self._names.add(name)

    def unset(self, name):
        from cpflow_amd import _ffi
        _ffi.set_tuning(name, None)

    def clear(self):
        for n in self._names:
            self.unset(n)


@pytest.fixture
def tune():
    t = _Tuning()
    yield t
    t.clear()
