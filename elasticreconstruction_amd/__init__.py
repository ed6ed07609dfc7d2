"""elasticreconstruction_amd -- MI355X-native (gfx950) drop-in for the two data-parallel stages of
qianyizh/ElasticReconstruction: TSDF depth integration with control-grid warp (Integrate/) and
pairwise ICP refinement + correspondence building (BuildCorrespondence/).

The compute lives in hand-written HIP kernels behind a plain C ABI (include/er_hip.h ->
liber_hip.so); this package is the host-side mirror of the reference's classes over that ABI.
There is no CPU fallback: importing works anywhere, but every operation needs the built
extension and a HIP device.
"""
from ._ffi import ErError, LIB_PATH, lib  # noqa: F401


def request_hw_queues(n=8):
    """OPT-IN (er_request_hw_queues, include/er_hip.h): one hardware queue per stream of the TSDF pipeline.  The HIP runtime reads
    GPU_MAX_HW_QUEUES at its first call, so this only takes effect before the process touches the GPU (before torch.cuda is
    used); it never overrides a value the user exported.  Importing the package does NOT set it; bench.py calls this first."""
    import os
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(int(n)))
    return int(os.environ["GPU_MAX_HW_QUEUES"])


__all__ = ["ErError", "LIB_PATH", "lib", "request_hw_queues"]
