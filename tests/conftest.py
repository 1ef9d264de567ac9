"""Shared pytest configuration.

* registers the ``gpu`` marker (tests that need a real MI355X and the built HIP library);
* puts the product package (``matrix-factorization-case-studies_amd/``) and the repo root on
  ``sys.path`` so ``import convex_dim_red`` resolves to the product mirror and
  ``import oracle`` to the CPU oracle (test infrastructure).
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "matrix-factorization-case-studies_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X and the built HIP library")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def golden():
    return load_golden
