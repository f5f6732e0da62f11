"""Import path kept from the reference (ops/furthest_point_sample/
points_sampler.py); the implementation lives in mm3d_pn2/modules.py."""
from ...modules import (DFPS_Sampler, FFPS_Sampler, FS_Sampler, Points_Sampler,  # noqa: F401
                        get_sampler_type)
