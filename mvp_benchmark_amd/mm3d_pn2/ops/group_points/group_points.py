"""Import path kept from the reference (ops/group_points/group_points.py); the
implementations live in mm3d_pn2/functional.py and mm3d_pn2/modules.py."""
from ...functional import GroupingOperation, grouping_operation  # noqa: F401
from ...modules import GroupAll, QueryAndGroup  # noqa: F401
