"""pytest configuration: `gpu` marker, import paths, shared fixtures."""

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
  if p not in sys.path:
    sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
  config.addinivalue_line(
      "markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


def golden(name):
  return dict(np.load(os.path.join(GOLDEN, name)))


@pytest.fixture(scope="session")
def handle():
  """A device handle; fails loudly (no skip) if the HIP path is unavailable."""
  from spectralcluster_amd import _lib
  return _lib.default_handle()
