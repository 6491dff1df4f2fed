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
    if "Sym" in globals() and isinstance(x, Sym):
        return Sym(lambda v: np.mean(v, axis=axis), (x,))
    return np.mean(x, axis=axis)


def reduce_max(x, axis=None):
    return np.max(x, axis=axis)


def cast(x, dtype):
    return np.asarray(x).astype(dtype) if isinstance(x, np.ndarray) else dtype(x)


# ---- symbols used by the reference's utils.py:10-175 (MI sandwich bounds, InfoNCE similarities) ----
float64 = np.float64
int32 = np.int32


def one_hot(indices, depth):
    return np.eye(int(depth), dtype=np.float32)[np.asarray(indices)]


def argsort(values):
    return np.argsort(np.asarray(values), kind="stable")


def squeeze(x):
    return np.squeeze(np.asarray(x))



def function(f=None, **kw):   # @tf.function / @tf.function(): graph compilation is irrelevant to the values
    return f if f is not None else (lambda g: g)


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


ALL_LAYERS = []          # every layer instance in creation order (the fixture scripts inject weights by walking it)


class Sym:
    """Deferred value for the Keras functional API (tf.keras.Input ... tf.keras.Model(inp, out)): a node remembers the
    function and the parent nodes it was produced from and is evaluated when the Model is called on real data."""

    def __init__(self, fn=None, parents=()):
        self.fn, self.parents = fn, tuple(parents)

    def eval(self, feed, memo=None):
        memo = {} if memo is None else memo
        if id(self) in memo:
            return memo[id(self)]
        if self in feed:
            v = feed[self]
        else:
            v = self.fn(*[q.eval(feed, memo) if isinstance(q, Sym) else q for q in self.parents])
        memo[id(self)] = v
        return v

    __hash__ = object.__hash__


def _has_sym(args):
    return any(isinstance(a, Sym) or (isinstance(a, (list, tuple)) and any(isinstance(b, Sym) for b in a)) for a in args)


def _defer(fn, args):
    """Sym node computing fn(*args) where Syms (also inside one level of lists) are replaced by their values."""
    flat, spec = [], []
    for a in args:
        if isinstance(a, (list, tuple)):
            spec.append(len(a))
            flat.extend(a)
        else:
            spec.append(None)
            flat.append(a)

    def run(*vals):
        out, i = [], 0
        for n in spec:
            if n is None:
                out.append(vals[i]); i += 1
            else:
                out.append(list(vals[i:i + n])); i += n
        return fn(*out)
    return Sym(run, flat)


class Layer:
    def __init__(self, *a, **k):
        ALL_LAYERS.append(self)

    def __call__(self, *args):
        if _has_sym(args):
            return _defer(self.call, args)
        return self.call(*args)


class Dense(Layer):
    """act(x @ kernel + bias), kernel [in, out] (Keras orientation); weights are injected by the fixture script."""

    def __init__(self, units, activation=None):
        super().__init__()
        self.units, self.activation = units, activation
        self.kernel = self.bias = None

    def call(self, x):
        z = np.asarray(x) @ self.kernel + self.bias
        return self.activation(z) if callable(self.activation) else _ACT[self.activation](z)


class _Input:
    def __init__(self, shape):
        self.shape = shape


class Sequential:
    def __init__(self, layers):
        self.layers = [l for l in layers if not isinstance(l, (_Input, Sym))]

    def __call__(self, x, training=None):
        for l in self.layers:
            x = l(x)
        return x

    def build(self, *a):
        return None

    @property
    def trainable_variables(self):
        return []


class LeakyReLU(Layer):
    def __init__(self, alpha=0.3):
        super().__init__()
        self.alpha = alpha

    def call(self, z):
        return np.where(z > 0, z, self.alpha * z)


class Add(Layer):
    def call(self, values):
        out = values[0]
        for v in values[1:]:
            out = out + v
        return out


class LayerNormalization(Layer):
    """Keras default: normalise the last axis, epsilon 1e-3, trainable gamma / beta."""

    def __init__(self, epsilon=1e-3):
        super().__init__()
        self.epsilon, self.gamma, self.beta = epsilon, None, None

    def call(self, x):
        mean = np.mean(x, -1, keepdims=True)
        var = np.mean(np.square(x - mean), -1, keepdims=True)
        return (x - mean) / np.sqrt(var + self.epsilon) * self.gamma + self.beta


class MultiHeadAttention(Layer):
    """Keras MultiHeadAttention(num_heads, key_dim) called as (query, value, key): per-head projections with biases,
    softmax(q k^T / sqrt(key_dim)) v, output projection back to the query width.  Written head by head with plain
    matrix products (deliberately not the einsum formulation of oracle/set_transformer_oracle.py)."""

    def __init__(self, num_heads, key_dim):
        super().__init__()
        self.num_heads, self.key_dim = num_heads, key_dim
        self.wq = self.bq = self.wk = self.bk = self.wv = self.bv = self.wo = self.bo = None

    def call(self, query, value, key=None):
        key = value if key is None else key
        out = np.zeros(query.shape[:-1] + (self.wo.shape[-1],)) + self.bo
        for h in range(self.num_heads):
            q = query @ self.wq[:, h, :] + self.bq[h]
            k = key @ self.wk[:, h, :] + self.bk[h]
            v = value @ self.wv[:, h, :] + self.bv[h]
            sc = (q @ np.swapaxes(k, -1, -2)) / np.sqrt(float(self.key_dim))
            sc = sc - sc.max(-1, keepdims=True)
            a = np.exp(sc)
            a = a / a.sum(-1, keepdims=True)
            out = out + (a @ v) @ self.wo[h]
        return out


class _FunctionalModel:
    def __init__(self, inputs, outputs):
        self.inputs, self.outputs = inputs, outputs

    def __call__(self, x, training=None):
        return self.outputs.eval({self.inputs: np.asarray(x)})

    @property
    def trainable_variables(self):
        return []


def _keras_input(shape):
    return Sym()


class _BinaryCrossentropy:
    """Keras BinaryCrossentropy(from_logits=True): mean over the batch of max(z,0) - z*y + log(1 + exp(-|z|))."""

    def __init__(self, from_logits=False):
        assert from_logits

    def __call__(self, y_true, y_pred):
        z, y = np.asarray(y_pred, dtype=np.float64), np.asarray(y_true, dtype=np.float64).reshape(np.shape(y_pred))
        return np.mean(np.maximum(z, 0) - z * y + np.log1p(np.exp(-np.abs(z))))


class _Adam:
    def __init__(self, learning_rate=1e-3):
        self.learning_rate = learning_rate


class GradientTape:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


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


class _ModelFactory(Model):
    """tf.keras.Model is both a base class (models.py subclasses it) and, called as Model(inputs, outputs), the functional
    constructor (set-transformer notebook)."""

    def __new__(cls, *args, **kw):
        if cls is _ModelFactory and len(args) == 2 and isinstance(args[0], Sym):
            return _FunctionalModel(*args)
        return super().__new__(cls)


keras = types.SimpleNamespace(
    layers=types.SimpleNamespace(Layer=Layer, Dense=Dense, Input=_Input, LeakyReLU=LeakyReLU, Add=Add,
                                 LayerNormalization=LayerNormalization, MultiHeadAttention=MultiHeadAttention),
    Model=_ModelFactory, Sequential=Sequential, Input=_keras_input, callbacks=types.SimpleNamespace(Callback=Callback),
    losses=types.SimpleNamespace(BinaryCrossentropy=_BinaryCrossentropy),
    optimizers=types.SimpleNamespace(Adam=_Adam))

abs = _abs  # noqa: A001 (tf.abs)
