"""sm64rt-legacy-renderer_amd -- MI355X-native implementation of RT64's ray-traced render path.

The product is csrc/ (HIP kernels + the C-ABI shim) built into librt64.so; see include/rt64.h for the boundary and
DESIGN.md for the path.  The Python modules here are the host-side harness only:
  rt64.py          ctypes mirror of the RT64_LIBRARY function table (what a C host binds with dlsym)
  sample_scene.py  the reference's sample scene issued call by call through that table
  tiles.py         image-tile partition across ranks + RCCL gather of the composited framebuffer
Because the directory name contains '-', import it through __graft_entry__.load_package() (module name
`sm64rt_legacy_renderer_amd`).
"""
from . import rt64  # noqa: F401
