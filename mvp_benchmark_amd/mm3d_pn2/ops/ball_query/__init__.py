"""ball_query operator (re-export)."""
from .ball_query import ball_query  # noqa: F401

__all__ = ["ball_query"]
