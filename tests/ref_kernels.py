"""The loader of the reference's own kernels (oracle/_ref) under the name the tests import: see oracle/ref_gpu.py."""
from oracle.ref_gpu import *  # noqa: F401,F403
from oracle.ref_gpu import _module, available  # noqa: F401
