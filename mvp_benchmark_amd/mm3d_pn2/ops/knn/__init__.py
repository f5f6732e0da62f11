from .knn import knn

__all__ = ['knn']
