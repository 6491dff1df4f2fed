"""Keras-shaped optimizer objects (reference train.py:128-129: tf.keras.optimizers.get('adam');
optimizer.learning_rate = 3e-4).  The update itself is the fused flat-buffer HIP kernel
dib_adam_kernel / dib_sgd_kernel."""
from __future__ import annotations


class Optimizer:
    name = "optimizer"

    def __init__(self, learning_rate: float):
        self.learning_rate = float(learning_rate)


class Adam(Optimizer):
    """Keras Adam defaults: lr 1e-3, beta_1 0.9, beta_2 0.999, epsilon 1e-7."""
    name = "adam"

    def __init__(self, learning_rate: float = 1e-3, beta_1: float = 0.9, beta_2: float = 0.999,
                 epsilon: float = 1e-7, **_):
        super().__init__(learning_rate)
        self.beta_1, self.beta_2, self.epsilon = float(beta_1), float(beta_2), float(epsilon)


class SGD(Optimizer):
    name = "sgd"

    def __init__(self, learning_rate: float = 0.01, **_):
        super().__init__(learning_rate)


def get(identifier) -> Optimizer:
    if isinstance(identifier, Optimizer):
        return identifier
    if isinstance(identifier, str):
        key = identifier.lower()
        if key == "adam":
            return Adam()
        if key == "sgd":
            return SGD()
    raise ValueError(f"unsupported optimizer {identifier!r} (supported: 'adam', 'sgd')")
