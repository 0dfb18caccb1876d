"""Deterministic, torch-RNG-independent tensor generator keyed by (seed, name).

Weights for the nets are 11-44 M parameters: they are never committed.  Both the reference modules
(in tests/golden/gen_golden.py) and the build's modules are loaded from this generator, so a golden vector
only needs (seed, net kind, shapes).  Arithmetic: splitmix64 counter hash -> uniform -> Box-Muller,
all in numpy uint64/float64 (version independent)."""
import hashlib

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _key(seed, name):
    h = hashlib.sha256(("%d/%s" % (seed, name)).encode()).digest()
    return np.uint64(int.from_bytes(h[:8], "little"))


def uniform01(seed, name, n):
    """n float64 values in (0, 1)."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        bits = _splitmix64(idx * np.uint64(2) + _key(seed, name))
    return ((bits >> np.uint64(11)).astype(np.float64) + 0.5) / float(1 << 53)


def normal(seed, name, shape, mean=0.0, std=1.0, dtype=torch.float32):
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = uniform01(seed, name + "#a", n)
    u2 = uniform01(seed, name + "#b", n)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return torch.from_numpy(mean + std * z).reshape(shape).to(dtype)


def uniform(seed, name, shape, lo=0.0, hi=1.0, dtype=torch.float32):
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(seed, name, n)
    return torch.from_numpy(lo + (hi - lo) * u).reshape(shape).to(dtype)


def randint(seed, name, shape, high):
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(seed, name, n)
    return torch.from_numpy(np.minimum((u * high).astype(np.int64), high - 1)).reshape(shape)


def blob_labels(seed, name, n, h, w, classes, block=16):
    """Piecewise-constant label map [n,1,h,w] (spatially coherent synthetic labels)."""
    bh, bw = -(-h // block), -(-w // block)
    coarse = randint(seed, name, (n, 1, bh, bw), classes)
    full = coarse.repeat_interleave(block, 2).repeat_interleave(block, 3)
    return full[:, :, :h, :w].contiguous()


def fill_state_dict(spec, seed, dtype=torch.float32, prefix=""):
    """spec: ordered {key: (shape, kind)}; kind in conv|bias|bn_weight|bn_bias|running_mean|running_var|nbt."""
    sd = {}
    for k, (shape, kind) in spec.items():
        name = prefix + k
        if kind == "conv":
            sd[k] = normal(seed, name, shape, 0.0, 0.02, dtype)
        elif kind == "bias":
            sd[k] = normal(seed, name, shape, 0.0, 0.02, dtype)
        elif kind == "bn_weight":
            sd[k] = normal(seed, name, shape, 1.0, 0.02, dtype)
        elif kind == "bn_bias":
            sd[k] = normal(seed, name, shape, 0.0, 0.02, dtype)
        elif kind == "running_mean":
            sd[k] = normal(seed, name, shape, 0.0, 0.05, dtype)
        elif kind == "running_var":
            sd[k] = uniform(seed, name, shape, 0.8, 1.2, dtype)
        elif kind == "nbt":
            sd[k] = torch.zeros((), dtype=torch.int64)
        else:
            raise ValueError(kind)
    return sd
