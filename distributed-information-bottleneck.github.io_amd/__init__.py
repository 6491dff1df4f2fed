"""MI355X-native Distributed Information Bottleneck training path.

Drop-in for the reference's Python surface on this path (reference models.py / train.py):
    DistributedIBNet, compile, fit, InfoBottleneckAnnealingCallback, SaveCompressionMatricesCallback
backed by hand-written HIP kernels for gfx950 (csrc/) behind the C ABI in include/dib_hip.h.
"""
from . import chaos_data, ctw, data, dense, infonce, losses, models, optimizers, set_transformer, utils, visualization  # noqa: F401
from .models import (Callback, DistributedIBNet, History, InfoBottleneckAnnealingCallback, InfoPerFeatureCallback,  # noqa: F401
                     PositionalEncoding, SaveCompressionMatricesCallback)
from .set_transformer import SetTransformerDIB  # noqa: F401

__all__ = ["DistributedIBNet", "InfoBottleneckAnnealingCallback", "SaveCompressionMatricesCallback",
           "InfoPerFeatureCallback", "PositionalEncoding", "Callback", "History", "models", "losses", "optimizers", "data", "utils",
           "visualization", "ctw", "chaos_data", "set_transformer", "SetTransformerDIB"]
