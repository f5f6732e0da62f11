"""Counterpart of utils/metrics/CD/__init__.py:1-2."""
from .chamfer3D.dist_chamfer_3D import chamfer_3DDist as cd
from .fscore import fscore

__all__ = ['cd', 'fscore']
