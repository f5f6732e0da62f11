from .three_interpolate import three_interpolate
from .three_nn import three_nn

__all__ = ['three_nn', 'three_interpolate']
