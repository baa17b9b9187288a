"""`lle.tiles` of the reference (python/lle/tiles/__init__.pyi): the tile value types, under the same import path."""
from .world import Direction, Gem, Laser, LaserSource

__all__ = ["Direction", "Gem", "Laser", "LaserSource"]
