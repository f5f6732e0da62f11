"""three_nn / three_interpolate operators (re-export)."""
from .three_interpolate import three_interpolate  # noqa: F401
from .three_nn import three_nn  # noqa: F401

__all__ = ["three_nn", "three_interpolate"]
