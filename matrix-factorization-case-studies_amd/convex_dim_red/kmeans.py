"""k-means gap statistic (reference src/convex_dim_red/kmeans.py:81-108).

Host-side scikit-learn code with no solver behind it -- it stays on the host exactly as in
the reference and is kept so the k-means drivers
(bin/run_*_kmeans.py) find ``gap_statistic`` under the same name, signature and random-number
order: ``n_trials`` distinct int32 seeds are drawn from ``random_state`` first, then each trial
builds its reference data set (feature-wise uniform box, or the same box in the leading PCA
coordinates) from its own seed and clusters it with KMeans(n_init=10).

The reference forwards ``n_jobs`` to ``KMeans``, an argument scikit-learn removed in 1.0; it
is dropped here (KMeans parallelises through its own thread pool), while the trials still
spread over ``n_jobs`` joblib workers.
"""
from __future__ import absolute_import, division

import numpy as np
from joblib import Parallel, delayed
from sklearn.cluster import KMeans
from sklearn.decomposition import TruncatedSVD
from sklearn.utils import check_random_state


def _box_sample(data, rng):
    """Uniform sample of data.shape from the feature-wise bounding box of ``data``."""
    lo, hi = data.min(axis=0), data.max(axis=0)
    return (hi - lo) * rng.uniform(size=data.shape) + lo


def _calculate_uniform_reference_wk(X, n_clusters, n_init=10, n_jobs=None, random_state=None):
    """Within-cluster dispersion of a uniform box sample (reference :18-36)."""
    rng = check_random_state(random_state)
    return KMeans(n_clusters=n_clusters, n_init=n_init, random_state=rng).fit(
        _box_sample(np.asarray(X), rng)).inertia_


def _calculate_pca_reference_wk(X, n_clusters, n_init=10, n_components=100, n_iter=10,
                                n_jobs=None, random_state=None):
    """The box is drawn in the coordinates of the leading right singular vectors and mapped
    back (reference :39-66)."""
    rng = check_random_state(random_state)
    X = np.asarray(X)
    svd = TruncatedSVD(n_components=n_components, n_iter=n_iter, random_state=rng).fit(X)
    axes = svd.components_
    sample = _box_sample(X.dot(axes.T), rng).dot(axes)
    return KMeans(n_clusters=n_clusters, n_init=n_init, random_state=rng).fit(sample).inertia_


def _calculate_reference_wk(X, n_components, reference='uniform', random_state=None):
    if reference == 'uniform':
        return _calculate_uniform_reference_wk(X, n_components, random_state=random_state)
    if reference == 'pca':
        return _calculate_pca_reference_wk(X, n_components, random_state=random_state)
    raise ValueError("unrecognized reference distribution '%s'" % reference)


def gap_statistic(X, Wk, n_components, n_trials=100, reference='uniform', n_jobs=1,
                  random_state=None):
    """Gap statistic of a k-means clustering with within-cluster dispersion ``Wk``:
    returns ``(gap, s_k)`` with gap = E*[log W_k] - log W_k over ``n_trials`` reference data
    sets and s_k = std * sqrt(1 + 1/n_trials) (reference :81-108)."""
    rng = check_random_state(random_state)
    seeds = []
    while len(seeds) < n_trials:                       # distinct seeds, in drawing order
        seed = rng.randint(np.iinfo(np.int32).max)
        if seed not in seeds:
            seeds.append(seed)
    dispersions = Parallel(n_jobs=n_jobs)(
        delayed(_calculate_reference_wk)(X, n_components, reference=reference, random_state=s)
        for s in seeds)
    log_ref = np.log(np.asarray(dispersions))
    sk = np.std(log_ref) * np.sqrt(1 + 1.0 / n_trials)
    return log_ref.mean() - np.log(Wk), sk
