"""Furthest-point-sampling operators and the Points_Sampler facade (re-export)."""
from .furthest_point_sample import furthest_point_sample, furthest_point_sample_with_dist  # noqa: F401
from .points_sampler import Points_Sampler  # noqa: F401

__all__ = ["furthest_point_sample", "furthest_point_sample_with_dist", "Points_Sampler"]
