"""nerf-navigation_amd: the MI355X (gfx950) Instant-NGP rendering core of nerf-navigation.

This directory is a path root, not a conventional package (its name has a hyphen).  Importing it -- e.g.
`importlib.import_module("nerf-navigation_amd")` with the repository root on sys.path -- puts the directory
itself on sys.path so that the reference's own import lines resolve to the drop-in packages here:

    import raymarching                      # nerf/renderer.py:9
    from gridencoder import GridEncoder     # encoding.py:60,64
    from shencoder import SHEncoder         # encoding.py:56
    from ffmlp import FFMLP                 # nerf/network_ff.py:7

All kernels live in lib/libngp_hip.so (built from csrc/ by hipcc for gfx950); see include/ngp_hip.h.
"""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import ngp_hip  # noqa: E402  (the ctypes binding; raises on use if the library has not been built)

__all__ = ["ROOT", "ngp_hip"]
