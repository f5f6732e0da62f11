"""Grouping operator and the QueryAndGroup / GroupAll modules (re-export)."""
from .group_points import GroupAll, QueryAndGroup, grouping_operation  # noqa: F401

__all__ = ["QueryAndGroup", "GroupAll", "grouping_operation"]
