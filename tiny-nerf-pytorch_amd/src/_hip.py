"""Locate the host package (tiny-nerf-pytorch_amd/tnerf) for the flat drop-in modules in this directory."""
import os
import sys

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from tnerf import ops, lib, trainer, dist  # noqa: E402,F401
