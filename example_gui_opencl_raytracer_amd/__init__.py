"""MI355X-native per-pixel Whitted trace behind the reference's ``cl_wrap_*`` API.

Package contents (only what the hot path needs):
  csrc/        HIP kernels (gfx950) + the C-ABI shim + host helpers
  api.py       ctypes mirror of include/opencl_wrap.h and include/hip_wrap_ext.h
  renderer.py  the raypng.c call protocol as an object; row-strip partitioning
  scene.py     scene wire format (render.map) and the benchmark scene generators
  textures.py  procedural stand-ins for the reference's PNG assets
  build.py     in-tree build of libopencl_wrap_hip.so
"""
from . import scene, textures  # noqa: F401  (pure numpy; the HIP library loads lazily via api)

CAMERA_RAYPNG = dict(origin=(0.8, 2.5, -8.0), look=(0.2, 0.0, 1.0), fov=90.0, focal=1.0)  # raypng.c:17-21
