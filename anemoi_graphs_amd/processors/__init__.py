from .post_process import RemoveUnconnectedNodes

__all__ = ["RemoveUnconnectedNodes"]
