"""MI355X-native primary/shadow ray-trace path: drop-in for Raytracer-in-CPP's CPU ThreadPool render.

Only what the path needs lives here: csrc/ (HIP kernels + C ABI + GL-free host scene code), the ctypes binding
(capi), the Python mirror of the reference's Flyscene interface (flyscene) and the row-shard helpers (shard).
"""
from . import capi, hipmem, shard  # noqa: F401
from .capi import load_library  # noqa: F401
from .flyscene import (Context, Flyscene, FrameGraph, HostScene, default_camera, make_lights, make_params, set_sphere,  # noqa: F401
                       sphere_offsets)
