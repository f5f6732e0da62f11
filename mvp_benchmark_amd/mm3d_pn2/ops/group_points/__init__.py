from .group_points import GroupAll, QueryAndGroup, grouping_operation

__all__ = ['QueryAndGroup', 'GroupAll', 'grouping_operation']
