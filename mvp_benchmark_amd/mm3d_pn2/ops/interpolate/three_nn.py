"""Import path kept from the reference (ops/interpolate/three_nn.py); the
implementation lives in mm3d_pn2/functional.py."""
from ...functional import ThreeNN, three_nn  # noqa: F401
