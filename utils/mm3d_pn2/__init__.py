"""Drop-in for the reference's ``utils/mm3d_pn2`` package
(``from mm3d_pn2 import furthest_point_sample, gather_points, ...``,
completion/model_utils.py:21).  Re-exports the MI355X-native operators."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from mvp_benchmark_amd.mm3d_pn2 import *  # noqa: E402,F401,F403
from mvp_benchmark_amd.mm3d_pn2 import __all__  # noqa: E402,F401
