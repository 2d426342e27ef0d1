from .masks import KNNAreaMaskBuilder

__all__ = ["KNNAreaMaskBuilder"]
