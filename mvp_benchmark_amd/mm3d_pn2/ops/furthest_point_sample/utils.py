"""Import path kept from the reference (ops/furthest_point_sample/utils.py)."""
from ...modules import calc_square_dist  # noqa: F401
