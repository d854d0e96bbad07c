"""TEST INFRASTRUCTURE -- closed-form tensor generator.

Golden vectors must be reproducible on the GPU box without shipping pickled
reference modules, so every parameter / buffer / input used for a golden is a
pure function of (key name, shape).  The generator is a splitmix64 hash of the
element index -- identical on every machine, no RNG state.
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix(i, salt):
    with np.errstate(over="ignore"):
        x = i + np.uint64(salt) * np.uint64(0x9E3779B97F4A7C15)
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return x


def uniform(name, shape, lo=-0.5, hi=0.5):
    """float32 array, element k = hash(k, crc32(name)) mapped to [lo, hi)."""
    n = int(np.prod(shape)) if len(shape) else 1
    i = np.arange(n, dtype=np.uint64)
    x = _splitmix(i, zlib.crc32(name.encode()) + 1)
    u = (x >> np.uint64(40)).astype(np.float64) / float(1 << 24)  # [0,1)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def fill_state_dict(shapes, gain=1.0):
    """shapes: {key: tuple} in reference state-dict naming.  Returns {key: np.ndarray}.

    conv / linear weights: uniform with std = gain / sqrt(fan_in);
    biases +-0.1; BatchNorm gamma 1+-0.1, beta +-0.1, running_mean +-0.1,
    running_var 1+-0.2 (so BN folding is exercised); num_batches_tracked 0.
    """
    out = {}
    for key, shape in shapes.items():
        shape = tuple(int(s) for s in shape)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[key] = np.zeros(shape, dtype=np.int64)
        elif leaf == "running_mean":
            out[key] = uniform(key, shape, -0.1, 0.1)
        elif leaf == "running_var":
            out[key] = uniform(key, shape, 0.8, 1.2)
        elif len(shape) >= 2:  # conv / linear weight
            fan_in = int(np.prod(shape[1:]))
            a = gain * np.sqrt(3.0 / fan_in)
            out[key] = uniform(key, shape, -a, a)
        elif leaf == "weight":  # BatchNorm gamma
            out[key] = uniform(key, shape, 0.9, 1.1)
        else:  # any bias (conv / linear / BN beta)
            out[key] = uniform(key, shape, -0.1, 0.1)
    return out
