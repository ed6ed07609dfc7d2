"""elasticreconstruction_amd -- MI355X-native (gfx950) drop-in for the two data-parallel stages of
qianyizh/ElasticReconstruction: TSDF depth integration with control-grid warp (Integrate/) and
pairwise ICP refinement + correspondence building (BuildCorrespondence/).

The compute lives in hand-written HIP kernels behind a plain C ABI (include/er_hip.h ->
liber_hip.so); this package is the host-side mirror of the reference's classes over that ABI.
There is no CPU fallback: importing works anywhere, but every operation needs the built
extension and a HIP device.
"""
import os as _os

# One hardware queue per stream of the TSDF pipeline (csrc/er_common.cpp); read by the HIP runtime at its first call, so it
# only takes effect when this package is imported before the process touches the GPU.  Never overrides the user's value.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from ._ffi import ErError, LIB_PATH, lib  # noqa: E402,F401

__all__ = ["ErError", "LIB_PATH", "lib"]
