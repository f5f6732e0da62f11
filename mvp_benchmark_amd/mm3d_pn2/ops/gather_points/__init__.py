"""gather_points operator (re-export)."""
from .gather_points import gather_points  # noqa: F401

__all__ = ["gather_points"]
