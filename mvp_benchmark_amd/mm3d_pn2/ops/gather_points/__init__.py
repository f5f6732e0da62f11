from .gather_points import gather_points

__all__ = ['gather_points']
