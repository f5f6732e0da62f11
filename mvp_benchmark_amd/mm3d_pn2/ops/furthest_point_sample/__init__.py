from .furthest_point_sample import (furthest_point_sample,
                                    furthest_point_sample_with_dist)
from .points_sampler import Points_Sampler

__all__ = [
    'furthest_point_sample', 'furthest_point_sample_with_dist',
    'Points_Sampler'
]
