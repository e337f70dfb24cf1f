"""glnn_amd: MI355X-native hot path of Graph-less Neural Networks (GLNN) -- teacher neighbour
aggregation + dense projections and the MLP student distillation step -- behind the reference's own
Python surface (`Model`, `train_mini_batch`, ...).  All arithmetic is hand-written HIP for gfx950 in
libglnn_hip.so (C ABI: include/glnn_hip.h); see DESIGN.md."""
from ._lib import GlnnError, LIB_PATH, lib  # noqa: F401

__all__ = ["lib", "GlnnError", "LIB_PATH"]
