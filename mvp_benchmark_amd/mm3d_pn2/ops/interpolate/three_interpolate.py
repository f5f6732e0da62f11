"""Import path kept from the reference (ops/interpolate/three_interpolate.py);
the implementation lives in mm3d_pn2/functional.py."""
from ...functional import ThreeInterpolate, three_interpolate  # noqa: F401
