"""oracle.weights -- seeded, reproducible values for the reference's TF variables.  TEST INFRASTRUCTURE ONLY.

The reference initialises with tf.contrib.layers.xavier_initializer / constant initialisers and ships no
checkpoint; fixtures therefore do not store weights, they store (seed, variable name, shape) and both sides --
the reference Python running under oracle/tf_shim (tests/golden/make_golden.py cells) and the tests on the GPU
box -- regenerate the values with `make()`.  Only the RAW output of numpy's PCG64 bit generator is used (its
stream is frozen by numpy's compatibility policy; Generator methods are not), turned into doubles the standard
way ((raw >> 11) * 2**-53) and rounded to float32, so an fp32 and an fp64 evaluation see identical numbers.

Distributions by the variable's leaf name (names as TF creates them, utils/tf_util.py:22,40-49,168,177;
tf.contrib.layers.batch_norm): every BN statistic and bias is non-trivial so that the reference's
conv -> bias -> BN -> activation order is actually exercised.
    weights          Xavier uniform, limit sqrt(6 / (fan_in + fan_out)), fans as TF computes them for [kh,kw,cin,cout]
    biases           U(-0.2, 0.2)
    beta             U(-0.2, 0.2)       gamma            U(0.5, 1.5)
    moving_mean      U(-0.2, 0.2)       moving_variance  U(0.5, 1.5)
"""
import hashlib
import math

import numpy as np


def _uniform01(seed, name, count):
    h = hashlib.sha256(f"{int(seed)}:{name}".encode()).digest()
    raw = np.random.PCG64(int.from_bytes(h[:16], "little")).random_raw(count)
    return (raw >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)


def make(seed, name, shape):
    """-> float32 array of `shape` for the TF variable `name` (e.g. 'layer1/layer1/conv_kv_ds/bn/gamma')."""
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if shape else 1
    u = _uniform01(seed, name, n).reshape(shape)
    leaf = name.split("/")[-1]
    if leaf == "weights":
        recept = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
        fan_in, fan_out = shape[-2] * recept, shape[-1] * recept
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        v = (2.0 * u - 1.0) * lim
    elif leaf in ("biases", "beta", "moving_mean"):
        v = (2.0 * u - 1.0) * 0.2
    elif leaf in ("gamma", "moving_variance"):
        v = 0.5 + u
    else:
        raise KeyError(f"no distribution for variable {name!r}")
    return v.astype(np.float32)


def make_all(seed, names_shapes):
    """[(name, shape), ...] -> {name: float32 array}"""
    return {name: make(seed, name, shape) for name, shape in names_shapes}
