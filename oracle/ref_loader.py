"""Loader for the *real* reference package, used ONLY to generate golden vectors.

TEST INFRASTRUCTURE -- never imported by the product (``convex_dim_red``),
never shipped, never run on the GPU box (``/root/reference`` does not exist
there).  Only ``oracle/gen_golden.py`` and ``tests/test_oracle_golden.py``
(skipped when the reference is absent) use it.

The reference (``/root/reference/src/convex_dim_red``) is pure Python whose only
compiled dependency is numba (``@jit(nopython=True)`` / ``@guvectorize``), which
is not installed in this image and cannot be installed (no network).  The
decorators do not change the arithmetic: ``jit`` compiles the same Python body,
and the three ``guvectorize`` kernels are called with 2-D arrays whose whole
extent is ONE gufunc element, i.e. the decorated body runs
once over the full arrays with the output allocated by the caller.  So the
loader registers an in-memory module named ``numba`` whose decorators return
the undecorated function (``jit``) or a wrapper that allocates the single
output and calls the body (``guvectorize``), then imports the reference
untouched from where it lies.  Nothing of the reference is copied.

``np.NaN`` (used at spg.py:310) was removed in NumPy 2; it is aliased to
``np.nan`` for the duration of the import.
"""
import importlib
import os
import sys
import types

import numpy as np

REFERENCE_SRC = os.environ.get("AA_REFERENCE_SRC", "/root/reference/src")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_SRC, "convex_dim_red"))


def _make_numba_standin():
    m = types.ModuleType("numba")

    def jit(*args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return args[0]
        return lambda fn: fn

    def guvectorize(signatures, layout, **kwargs):
        ins, outs = layout.split("->")
        n_in = ins.count("(")
        out_sig = outs.strip()

        def deco(fn):
            def wrapper(*call_args):
                inputs = call_args[:n_in]
                if len(call_args) > n_in:          # output passed explicitly
                    fn(*call_args)
                    return call_args[n_in]
                # single output: "(m, n) -> (m, n)" takes the shape of arg 0,
                # "... -> (i, k)" takes the shape of arg 2 (initial weights).
                if out_sig.replace(" ", "") == "(m,n)":
                    like = np.asarray(inputs[0])
                else:
                    like = np.asarray(inputs[2])
                out = np.empty(like.shape, dtype=np.float64)
                fn(*inputs, out)
                return out
            wrapper.__name__ = fn.__name__
            wrapper.__doc__ = fn.__doc__
            return wrapper
        return deco

    m.jit = jit
    m.njit = jit
    m.guvectorize = guvectorize
    class _Type:                      # supports ``float64[:, :]`` in signatures
        def __getitem__(self, item):
            return self

    m.float64 = _Type()
    m.int32 = _Type()
    m.int64 = _Type()
    m.prange = range
    return m


_loaded = None


def load_reference():
    """Return the reference ``convex_dim_red`` package (imported under the
    private name ``_ref_convex_dim_red`` so it never shadows the product)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not reference_available():
        raise RuntimeError("reference sources not present at %s" % REFERENCE_SRC)

    if not hasattr(np, "NaN"):
        np.NaN = np.nan  # spg.py:310
    had_numba = "numba" in sys.modules
    if not had_numba:
        sys.modules["numba"] = _make_numba_standin()

    # Import under an alias so that ``import convex_dim_red`` elsewhere in the
    # same interpreter still resolves to the product package.
    saved = {k: v for k, v in sys.modules.items()
             if k == "convex_dim_red" or k.startswith("convex_dim_red.")}
    for k in saved:
        del sys.modules[k]
    sys.path.insert(0, REFERENCE_SRC)
    try:
        # kmeans.py imports sklearn/joblib only; fine.
        pkg = importlib.import_module("convex_dim_red")
        for sub in ("archetypal_analysis", "gpnh_convex_coding", "spg",
                    "simplex_projection", "furthest_sum",
                    "stochastic_matrices", "validation_utils"):
            importlib.import_module("convex_dim_red." + sub)
    finally:
        sys.path.remove(REFERENCE_SRC)
    ref_mods = {k: v for k, v in sys.modules.items()
                if k == "convex_dim_red" or k.startswith("convex_dim_red.")}
    for k, v in ref_mods.items():
        del sys.modules[k]
        sys.modules["_ref_" + k] = v
    sys.modules.update(saved)
    if not had_numba:
        del sys.modules["numba"]
    _loaded = pkg
    return pkg
