"""footprints_amd -- MI355X-native (gfx950) implementation of the Footprints hot path.

Drop-in surface of nianticlabs/footprints for the network forward/backward, the multi-head loss and the
Adam step (reference: footprints/network.py, training/losses.py, training/train.py:150-156,
model_manager.py, predict_simple.py), executed by hand-written HIP kernels behind the C ABI declared in
include/footprints_hip.h.  No CPU compute path and no PyTorch-op fallback exist in this package.
"""
from .network import FootprintNetwork  # noqa: F401

__all__ = ["FootprintNetwork"]
