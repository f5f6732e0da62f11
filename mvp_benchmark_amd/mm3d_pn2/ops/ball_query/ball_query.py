"""Import path kept from the reference (ops/ball_query/ball_query.py); the
implementation lives in mm3d_pn2/functional.py."""
from ...functional import BallQuery, ball_query  # noqa: F401
