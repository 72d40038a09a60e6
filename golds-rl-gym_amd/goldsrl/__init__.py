"""goldsrl -- host-side mirror of the reference's fed_gym API surface for the PAAC actor loop,
backed by libgoldsrl.so (hand-written HIP for gfx950, C ABI in include/goldsrl.h)."""
from . import _ffi  # noqa: F401

__all__ = ["_ffi"]
