from . import _lib, ops  # noqa: F401
