"""Drop-in for the reference's ``utils/metrics`` package: the completion code
does ``sys.path.append("../utils"); from metrics import cd, fscore, emd``
(completion/model_utils.py:19-20).  Re-exports the MI355X-native operators."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from mvp_benchmark_amd.metrics import cd, fscore, emd  # noqa: E402

__all__ = ['cd', 'fscore', 'emd']
