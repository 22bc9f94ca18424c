"""pytest configuration: the `gpu` marker and import paths."""

import os
import sys

import pytest

_TESTS = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_TESTS)
for path in (_ROOT, _TESTS):
  if path not in sys.path:
    sys.path.insert(0, path)


# Some tests use torch next to the library (device memory bookkeeping, torch.distributed).  The
# PyTorch-ROCm wheel bundles its own HIP/HSA runtime, and that copy only finds the GPU when it is
# the FIRST HIP runtime loaded into the process (measured: libuisrnn_hip.so -- system ROCm --
# first, then torch.cuda.init() -> "No HIP GPUs are available"; the other order works).  A full
# run loads torch during collection anyway; this makes single-file runs behave the same.
try:
  import torch  # noqa: F401  pylint: disable=unused-import,wrong-import-position
except ImportError:
  pass


def pytest_configure(config):
  config.addinivalue_line(
      'markers', 'gpu: needs a real MI355X (run with `-m gpu` on the GPU box)')


@pytest.fixture(scope='session')
def oracle_lib():
  from oracle import oracle  # pylint: disable=import-outside-toplevel
  oracle.lib()
  return oracle
