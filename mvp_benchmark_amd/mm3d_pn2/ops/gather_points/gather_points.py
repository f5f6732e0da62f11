"""Import path kept from the reference (ops/gather_points/gather_points.py); the
implementation lives in mm3d_pn2/functional.py."""
from ...functional import GatherPoints, gather_points  # noqa: F401
