"""pytest configuration: the `gpu` marker and import paths."""

import os
import sys

import pytest

_TESTS = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_TESTS)
for path in (_ROOT, _TESTS):
  if path not in sys.path:
    sys.path.insert(0, path)


def pytest_configure(config):
  config.addinivalue_line(
      'markers', 'gpu: needs a real MI355X (run with `-m gpu` on the GPU box)')


@pytest.fixture(scope='session')
def oracle_lib():
  from oracle import oracle  # pylint: disable=import-outside-toplevel
  oracle.lib()
  return oracle
