"""k-means gap statistic (reference src/convex_dim_red/kmeans.py).

Out of scope of the MI355X hot path (a thin wrapper over
scikit-learn's KMeans that does not touch the AA/GPNH solver).  The name is kept so
``from convex_dim_red import gap_statistic`` resolves; calling it says where to go.
"""


def gap_statistic(*args, **kwargs):
    raise NotImplementedError(
        "gap_statistic is outside the MI355X solver path; use the reference "
        "implementation (scikit-learn KMeans) for k-means experiments")
