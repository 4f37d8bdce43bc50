"""tf-recomm_amd: MI355X-native SVD matrix-factorisation training step.

One hot path of jilljenn/TF-recomm (ops.inference_svd + ops.optimization + the
svd_train_val.py minibatch step + dataio.ShuffleIterator) as hand-written gfx950 HIP
kernels behind the C-ABI of include/tfrecomm.h.  Import as ``import tfrecomm_amd``
(the directory name has a hyphen; ``tfrecomm_amd.py`` at the repo root aliases it).
"""
from . import _lib
from ._lib import TfrError, OutOfRangeError
from .engine import SvdModel, device_copy_rate
from . import dataio, graph, ops, config, cats, adaptive_test
from .fm import FmModel
from .als import MangakiALS3

__all__ = ["SvdModel", "device_copy_rate", "TfrError", "OutOfRangeError", "_lib", "dataio", "graph", "ops", "config", "cats", "adaptive_test",
           "FmModel", "MangakiALS3"]
