"""TEST INFRASTRUCTURE ONLY - a NumPy stand-in for the handful of TensorFlow/Keras symbols that the reference's
models.py touches, so that the reference's OWN source (models.py:11-149: PositionalEncoding, DistributedIBNet.__init__ /
call, InfoBottleneckAnnealingCallback) can be imported and executed in this container, where TensorFlow cannot be
installed.  Used only by tests/golden/make_golden_models.py to produce fixtures; nothing in the product imports it.

What this pins and what it does not: every primitive below has one obvious meaning (concat, split, sin, exp, square,
reduce_sum/mean over an axis, Dense = act(x @ kernel[in,out] + bias), N(mean, stddev) = mean + stddev * eps), so running
the reference's code on it pins the GRAPH the reference builds - split order of (mu | logvar), the KL formula and its
reduction axes, positional-encoding block order, feature concat order, the beta ramp formula in float32.  It does not pin
TensorFlow kernel numerics or Keras internals (initialisers, fit loop, Adam, loss classes): those stay 'parity unpinned'.

Arithmetic is float64 (the oracle's dtype) except where the reference asks for float32 explicitly (tf.cast / tf.Variable
dtype / tf.math.log of Python floats, which TF evaluates in float32).
"""
import types

import numpy as np

float32 = np.float32
_EPS_QUEUE = []          # standard-normal draws handed out by random.normal in call order


def push_eps(arrays):
    _EPS_QUEUE.extend(np.asarray(a, dtype=np.float64) for a in arrays)


def concat(values, axis):
    return np.concatenate([np.asarray(v) for v in values], axis=axis)


def split(value, num_or_size_splits, axis=-1):
    value = np.asarray(value)
    if isinstance(num_or_size_splits, (int, np.integer)):
        return np.split(value, int(num_or_size_splits), axis=axis)
    idx = np.cumsum(list(num_or_size_splits))[:-1]
    return np.split(value, idx, axis=axis)


def exp(x):
    return np.exp(x)


def square(x):
    return np.square(x)


def reduce_sum(x, axis=None, keepdims=False):
    return np.sum(np.asarray(x) if not isinstance(x, list) else np.stack(x), axis=axis, keepdims=keepdims)


def reduce_mean(x, axis=None):
    return np.mean(x, axis=axis)


def reduce_max(x, axis=None):
    return np.max(x, axis=axis)


def cast(x, dtype):
    return np.asarray(x).astype(dtype) if isinstance(x, np.ndarray) else dtype(x)


# ---- symbols used by the reference's utils.py:10-175 (MI sandwich bounds, InfoNCE similarities) ----
float64 = np.float64


def function(f):            # @tf.function: graph compilation is irrelevant to the values
    return f


def shape(x):
    return np.array(np.asarray(x).shape)


def reshape(x, new_shape):
    return np.reshape(x, [int(v) for v in new_shape])


def expand_dims(x, axis):
    return np.expand_dims(x, axis)


def maximum(a, b):
    return np.maximum(a, b)


def matmul(a, b, transpose_b=False):
    return np.asarray(a) @ (np.asarray(b).T if transpose_b else np.asarray(b))


def tile(x, multiples):
    return np.tile(x, multiples)


def sqrt(x):
    return np.sqrt(x)


def eye(n, dtype=np.float64):
    return np.eye(int(n), dtype=dtype)


def _abs(x):
    return np.abs(x)


def _normalize(x, ord=2, axis=-1):
    assert ord == 2
    n = np.sqrt(np.sum(np.square(x), axis=axis, keepdims=True))
    return x / n, n


def _diag_part(x):
    return np.diagonal(x).copy()      # TF tensors are immutable: later `x *= ...` in the reference must not alias this


linalg = types.SimpleNamespace(normalize=_normalize, diag_part=_diag_part)


def _log(x):
    # TF turns a Python float into a float32 tensor before tf.math.log
    return np.log(np.float32(x)) if isinstance(x, (float, int)) else np.log(x)


math = types.SimpleNamespace(sin=np.sin, log=_log)


def _normal(shape, mean=0.0, stddev=1.0, dtype=None):
    eps = _EPS_QUEUE.pop(0)
    assert tuple(eps.shape) == tuple(shape), (eps.shape, shape)
    return mean + stddev * eps          # a draw from N(mean, stddev^2)


random = types.SimpleNamespace(normal=_normal)


class Variable:
    def __init__(self, value, dtype=None, trainable=True):
        self._v = (dtype or np.float64)(value)
        self.dtype = dtype

    def assign(self, v):
        self._v = (self.dtype or np.float64)(v)

    def value(self):
        return self._v

    def __mul__(self, other):
        return self._v * other

    __rmul__ = __mul__

    def __float__(self):
        return float(self._v)


_ACT = {None: lambda z: z, "linear": lambda z: z, "relu": lambda z: np.maximum(z, 0.0), "tanh": np.tanh,
        "sigmoid": lambda z: 1.0 / (1.0 + np.exp(-z)), "leaky_relu": lambda z: np.where(z > 0, z, 0.2 * z)}


class Layer:
    def __init__(self, *a, **k):
        pass

    def __call__(self, x):
        return self.call(x)


class Dense(Layer):
    """act(x @ kernel + bias), kernel [in, out] (Keras orientation); weights are injected by the fixture script."""

    def __init__(self, units, activation=None):
        self.units, self.activation = units, activation
        self.kernel = self.bias = None

    def call(self, x):
        return _ACT[self.activation](np.asarray(x) @ self.kernel + self.bias)


class _Input:
    def __init__(self, shape):
        self.shape = shape


class Sequential:
    def __init__(self, layers):
        self.layers = [l for l in layers if not isinstance(l, _Input)]

    def __call__(self, x):
        for l in self.layers:
            x = l(x)
        return x

    def build(self, *a):
        return None


class Model(Layer):
    def __init__(self, *a, **k):
        self.metrics_log, self.losses = {}, []

    def add_metric(self, value, name):
        self.metrics_log[name] = value.value() if isinstance(value, Variable) else value

    def add_loss(self, value):
        self.losses.append(value)

    def __call__(self, x):
        self.metrics_log, self.losses = {}, []
        return self.call(x)


class Callback:
    def __init__(self):
        self.model = None


class FixedBatches:
    """Stand-in for the tf.data pipeline of utils.py:68 (repeat().shuffle().batch().take()): yields the given batches
    in order - the reference's shuffle is random, so the fixture fixes which rows form each batch."""

    def __init__(self, batches):
        self.batches = list(batches)

    def repeat(self):
        return self

    def shuffle(self, n):
        return self

    def batch(self, n):
        return self

    def take(self, n):
        return self.batches[:n]


keras = types.SimpleNamespace(
    layers=types.SimpleNamespace(Layer=Layer, Dense=Dense, Input=_Input),
    Model=Model, Sequential=Sequential, callbacks=types.SimpleNamespace(Callback=Callback))

abs = _abs  # noqa: A001 (tf.abs)
