"""`lle.types` of the reference (python/lle/types.py): semantic aliases, no runtime behaviour."""
Position = tuple[int, int]   # (i, j) = (row, column)
AgentId = int
LaserId = int                # the beam carries the identifier of its source

__all__ = ["AgentId", "LaserId", "Position"]
