"""knn operator (re-export)."""
from .knn import knn  # noqa: F401

__all__ = ["knn"]
