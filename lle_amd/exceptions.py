"""`lle.exceptions` of the reference (python/lle/exceptions/__init__.pyi; src/bindings/pyexceptions.rs:8-34): all subclass ValueError."""
from .world import InvalidActionError, InvalidLevelError, InvalidWorldStateError, ParsingError

__all__ = ["InvalidActionError", "InvalidLevelError", "InvalidWorldStateError", "ParsingError"]
