"""Keras-shaped loss objects accepted by DistributedIBNet.compile (reference train.py:138-142).

Only the *identity* of the loss lives here; the arithmetic runs in csrc/dib_elementwise.h
(dib_loss_kernel).  Kinds: reference data.py:65 BinaryCrossentropy(from_logits=True),
data.py:343 SparseCategoricalCrossentropy(from_logits=True), 'mse'.
"""
from __future__ import annotations


class Loss:
    kind = None
    name = "loss"

    def __repr__(self):
        return f"{type(self).__name__}(kind={self.kind!r})"


class BinaryCrossentropy(Loss):
    name = "binary_crossentropy"

    def __init__(self, from_logits: bool = False, **_):
        self.from_logits = bool(from_logits)
        self.kind = "bce_logits" if from_logits else "bce"


class SparseCategoricalCrossentropy(Loss):
    name = "sparse_categorical_crossentropy"

    def __init__(self, from_logits: bool = False, **_):
        if not from_logits:
            raise NotImplementedError("SparseCategoricalCrossentropy(from_logits=False) is not on the DIB path; "
                                      "the reference uses from_logits=True (data.py:343)")
        self.from_logits = True
        self.kind = "sparse_cce_logits"


class MeanSquaredError(Loss):
    name = "mean_squared_error"
    kind = "mse"

    def __init__(self, **_):
        pass


_BY_NAME = {
    "mse": MeanSquaredError, "mean_squared_error": MeanSquaredError, "MSE": MeanSquaredError,
    "binary_crossentropy": BinaryCrossentropy, "bce": BinaryCrossentropy,
}


def get(identifier) -> Loss:
    """tf.keras.losses.get equivalent for the kinds on this path."""
    if isinstance(identifier, Loss):
        return identifier
    if isinstance(identifier, str):
        if identifier == "infonce":
            raise NotImplementedError("'infonce' is not a Keras-path loss: it trains through the custom loop "
                                      "(reference train.py:180-289) -> dib_amd.infonce.fit_infonce / "
                                      "`python -m dib_amd.train --infonce_loss True`")
        if identifier in _BY_NAME:
            return _BY_NAME[identifier]()
    kind = getattr(identifier, "kind", None)
    if kind is not None:
        return identifier
    raise ValueError(f"unsupported loss {identifier!r}")
