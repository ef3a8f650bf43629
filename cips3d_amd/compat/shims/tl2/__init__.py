"""Minimal tl2 stand-in (cips3d_amd/compat/shims/README.md)."""
__cips3d_shim__ = True
