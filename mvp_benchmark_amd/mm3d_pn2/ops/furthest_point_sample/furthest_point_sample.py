"""Import path kept from the reference (ops/furthest_point_sample/
furthest_point_sample.py); the implementation lives in mm3d_pn2/functional.py."""
from ...functional import (FurthestPointSampling, FurthestPointSamplingWithDist,  # noqa: F401
                           furthest_point_sample, furthest_point_sample_with_dist)
