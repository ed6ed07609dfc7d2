"""elasticreconstruction_amd -- MI355X-native (gfx950) drop-in for the two data-parallel stages of
qianyizh/ElasticReconstruction: TSDF depth integration with control-grid warp (Integrate/) and
pairwise ICP refinement + correspondence building (BuildCorrespondence/).

The compute lives in hand-written HIP kernels behind a plain C ABI (include/er_hip.h ->
liber_hip.so); this package is the host-side mirror of the reference's classes over that ABI.
There is no CPU fallback: importing works anywhere, but every operation needs the built
extension and a HIP device.
"""
from ._ffi import ErError, LIB_PATH, lib  # noqa: F401

__all__ = ["ErError", "LIB_PATH", "lib"]
