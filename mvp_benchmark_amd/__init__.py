"""mvp_benchmark_amd -- MI355X (gfx950) native point-cloud op layer.

Drop-in for the data-parallel hot path of paul007pl/MVP_Benchmark: the
``utils/metrics`` (Chamfer distance, F-score, auction EMD) and
``utils/mm3d_pn2`` (PointNet++ set-abstraction ops) operator API, backed by
hand-written HIP kernels in ``libmvpops.so`` (C ABI: ``include/mvpops.h``).

    from mvp_benchmark_amd.metrics import cd, fscore, emd
    from mvp_benchmark_amd.mm3d_pn2 import (furthest_point_sample,
        gather_points, grouping_operation, ball_query, knn, three_nn,
        three_interpolate)

There is no CPU fallback: every op raises if ``libmvpops.so`` is missing or
an input tensor is not on a GPU.
"""
from . import _lib  # noqa: F401

__version__ = "0.1.0"
