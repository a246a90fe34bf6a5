"""flame_amd -- MI355X-native NLTGV2-L1 graph regularizer (the one FLaME hot path this repo builds).

Layout:
  csrc/                 HIP kernels + C-ABI (include/flame_nltgv2.h) -> libflame_nltgv2_hip.so
  regularizer.py        ctypes mirror of flame::optimizers::nltgv2_l1_graph_regularizer
  stereo.py             ctypes mirror of Flame::updateFeatureIDepths / stereo::* (include/flame_stereo.h)
  synth.py              synthetic Delaunay-graph inputs for tests and bench
There is no CPU fallback in this package: without the HIP library / a GPU every compute call raises.
"""
from .regularizer import (  # noqa: F401
    NLTGV2Error,
    Params,
    Regularizer,
    delaunay,
    library_path,
    load_library,
)
