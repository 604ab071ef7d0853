"""tnerf — host-side binding of the MI355X TinyNeRF hot path (libtnerf_hip.so).

Import order matters for nothing; the shared library is loaded on first use and its absence is an
error (there is no CPU fallback)."""
from . import lib  # noqa: F401

__all__ = ["lib", "ops", "trainer", "dist"]
