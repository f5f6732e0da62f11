"""Counterpart of utils/metrics/EMD/__init__.py:1."""
from .emd_module import emdModule as emd

__all__ = ['emd']
