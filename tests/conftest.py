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


# ---------------------------------------------------------------- yardsticks of the GPU tests
def ulp_perturbed(X, seed=5):
    """X with every entry moved by about one unit in the last place.

    The alternating solvers are not contractions: the BB step lengths of the SPG solvers, the
    max-norm first step (spg.py:178-189, :326-336), QPs that end at the function-evaluation
    cap and the support changes of the projections amplify rounding differences from one outer
    iteration to the next (measured on the C3 stand-in: a 1-ulp perturbation of X moves the
    ORACLE's own cost by 6e-7 relative and its dictionary by 1e-3 after 50 outer iterations;
    on the C2 stand-in by 4e-6 after 5).  Wherever a run is long enough for that to matter, the
    tolerance of the HIP-vs-oracle comparison is therefore tied to the oracle's own sensitivity:
    ``tol = max(floor, 20 x |oracle(X) - oracle(ulp_perturbed(X))|)`` -- the HIP path has to
    agree with the oracle as well as the oracle agrees with itself when its input moves by one
    ulp; short runs keep the plain rounding-level tolerances."""
    return X * (1.0 + 2e-16 * np.random.RandomState(seed).standard_normal(X.shape))


def f32_perturbed(X, seed=6):
    """X with every entry moved by about one float32 rounding (6e-8 relative): what storing the
    data in float32 does to the inputs of an otherwise exact run."""
    return X * (1.0 + 6e-8 * np.random.RandomState(seed).standard_normal(X.shape))


def oracle_twins(orc, run, X, dtype, operands=True):
    """The oracle's own runs on perturbed inputs: the yardstick of every leg that is not at
    rounding level.  float64 legs -- the data moved by one ulp (three draws).  float32 legs -- the
    data moved by one float32 rounding (three draws) AND the exact data with the non-data operands
    of the big contractions against X rounded to float32 (oracle.operand_rounding): the float32
    mode feeds the matrix cores float32 operands -- the dictionary / weights / search direction in
    the reduce-over-rows pass, C X and X'Z in the row-local pass -- and an archetype row C X is an
    average over many samples, so rounding IT moves it far more than float32-sized noise on the
    samples does.  `run(X)` returns whatever the caller compares.  `operands=False` leaves the
    last twin out: with a dictionary SPG run to ITS stopping rule (hundreds of inner iterations)
    the oracle's closures, which re-evaluate f and df from rounded operands at every trial point,
    no longer converge (the device keeps one consistent quadratic model per update instead), and
    the twin would take minutes."""
    if dtype == "float64":
        return [run(ulp_perturbed(X, s)) for s in (5, 6, 7)]
    out = [run(f32_perturbed(X, s)) for s in (6, 7, 8)]
    if operands:
        with orc.operand_rounding(np.float32):
            out.append(run(X))
    return out
