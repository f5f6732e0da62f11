"""Counterpart of the reference's ``utils/metrics`` package
(utils/metrics/__init__.py:1-2): ``cd``, ``fscore``, ``emd``."""
from .CD import cd, fscore
from .EMD import emd

__all__ = ['cd', 'fscore', 'emd']
