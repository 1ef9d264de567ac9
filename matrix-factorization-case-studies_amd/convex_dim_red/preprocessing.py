"""Driver-side preprocessing on the GPU.

What ``bin/run_hadisst_aa.py`` / ``bin/run_jra55_pca_gpnh.py`` do with NumPy before they fit
(reference ``run_hadisst_aa.py:112-146,196-209``): multiply the field by its latitude weights,
flatten the feature dimensions, drop every grid point (column) that is missing at any time, and
split the rows into a training and a validation block.  ``weight_and_flatten_on_device`` does the
same on the device from ONE upload of the raw field and returns a ``DeviceData`` that the
estimators accept wherever they accept a data matrix; the NetCDF / xarray I/O stays with the
driver, which passes ``da.values`` (sample dimension first) and the weights it computed.
"""
from __future__ import absolute_import, division

import numpy as np

from . import _backend


class DeviceData(object):
    """A preprocessed data matrix resident on the GPU.

    ``shape``  -- (n_samples, n_valid_features) of the block it holds;
    ``valid``  -- boolean mask over the flattened features (True: kept), what the drivers use to
                  put archetypes back on the grid;
    ``to_host()`` -- the matrix as float64 NumPy (downloaded once, then cached).

    Pass it as ``data`` to ``ArchetypalAnalysis.fit_transform / transform`` or
    ``GPNHConvexCoding.fit_transform``; ``close()`` (or ``with``) frees the device copy."""

    def __init__(self, ctx, shape, valid, original_shape):
        self._ctx = ctx
        self.shape = shape
        self.valid = valid
        self.original_shape = original_shape
        self.ndim = 2
        self._host = None

    @property
    def dtype(self):
        return np.dtype(np.float32 if self._ctx.dtype_code == _backend.AA_F32 else np.float64)

    def borrow(self):
        if self._ctx is None or not self._ctx.h:
            raise RuntimeError("DeviceData has been closed")
        return _backend._Borrowed(self._ctx)

    def to_host(self):
        if self._host is None:
            self._host = self._ctx.get_data()
        return self._host

    def close(self):
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __deepcopy__(self, memo):          # never copied into a model (estimators keep no data)
        return self


def weight_and_flatten_on_device(values, weights=None, rows=None, dtype=None, device=None):
    """``values``: array (n_samples, *feature_dims), NaN where data are missing (``da.values``
    with the sample dimension first, as ``weight_and_flatten_data`` arranges it);
    ``weights``: None or an array broadcastable to ``feature_dims`` (e.g. the latitude weights
    ``sqrt(cos(lat))[:, None]`` for (lat, lon) fields);
    ``rows``: None (all samples) or a ``slice`` / ``(start, stop)`` of the block to keep resident
    (``slice(0, n_training)`` for the training set, ``slice(n_training, None)`` for validation --
    the NaN mask is always taken over ALL samples, as the reference takes it before it splits).
    Returns a ``DeviceData``."""
    values = np.asarray(values)
    if values.ndim < 2:
        raise ValueError("expected an array with a sample dimension and at least one feature dimension")
    n_total = values.shape[0]
    feature_shape = values.shape[1:]
    flat = values.reshape(n_total, -1)
    col_w = None
    if weights is not None:
        col_w = np.ascontiguousarray(np.broadcast_to(np.asarray(weights, dtype=np.float64),
                                                     feature_shape)).reshape(-1)
    if rows is None:
        start, stop = 0, n_total
    elif isinstance(rows, slice):
        start, stop, step = rows.indices(n_total)
        if step != 1:
            raise ValueError("rows must be a contiguous block")
    else:
        start, stop = rows
    ctx = _backend.Context(dtype=dtype, device=device)
    try:
        valid = ctx.set_data_weighted(flat, col_w, start, stop - start)
    except Exception:
        ctx.close()
        raise
    return DeviceData(ctx, (stop - start, int(valid.sum())), valid, feature_shape)
