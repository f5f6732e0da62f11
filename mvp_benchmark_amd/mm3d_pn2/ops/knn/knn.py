"""Import path kept from the reference (ops/knn/knn.py); the implementation
lives in mm3d_pn2/functional.py."""
from ...functional import KNN, knn  # noqa: F401
