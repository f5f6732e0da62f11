from .ball_query import ball_query

__all__ = ['ball_query']
