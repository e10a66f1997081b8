"""falkordb_amd — MI355X-native traversal engine behind FalkorDB's Matrix / Delta_Matrix API.

The product is libfgpu.so (HIP kernels + C ABI, include/fgpu.h).  The Python here is
plumbing: ctypes handles (`engine`), the build recipe (`build`), the multi-GPU BFS driver
over torch.distributed (`dist`) and synthetic inputs for the bench.
"""
from ._ffi import FgpuError, load  # noqa: F401

__all__ = ["FgpuError", "load"]
