"""Seeded synthetic weights for the two networks whose real weights are absent from the
reference checkout (`/root/reference/.MISSING_LARGE_BLOBS`: det and rec `.pdiparams`).

Distribution per SURVEY.md §8(d) "Weights": counter-based splitmix64 keyed on
FNV-1a(name) ^ 0x0C125EED; every draw is exact integer/float64 arithmetic (normal = sum of
12 uniforms - 6), so the file is bit-identical on every machine.
Writes `models/<net>/synthetic.pdiparams` in the real `.pdiparams` record format, so the
runtime loads it through the same reader as real weights.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pdmodel import Program, write_params  # noqa: E402

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def fnv1a(name):
    h = 0xCBF29CE484222325
    for b in name.encode():
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix(seed, n, stream):
    """n uniforms in [0,1) from counter-based splitmix64; `stream` decorrelates draws."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64) + np.uint64(stream) * np.uint64(1 << 40)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform(seed, n, lo, hi, stream=0):
    return lo + (hi - lo) * _splitmix(seed, n, stream)


def normal(seed, n, std):
    acc = np.zeros(n, dtype=np.float64)
    for k in range(12):
        acc += _splitmix(seed, n, k + 1)
    return (acc - 6.0) * std


def synth_tensor(name, dims):
    seed = fnv1a(name) ^ 0x0C125EED
    n = int(np.prod(dims)) if dims else 1
    base = name.rsplit(".", 1)[0]
    suf = name.rsplit(".", 1)[1]
    if base.startswith("batch_norm"):
        if suf == "w_0":
            v = uniform(seed, n, 0.9, 1.1)
        elif suf in ("b_0", "w_1"):
            v = uniform(seed, n, -0.1, 0.1)
        else:  # w_2 variance
            v = uniform(seed, n, 0.8, 1.2)
    elif base.startswith("mobile_one_block") or base.startswith("whswish_b"):
        v = uniform(seed, n, 0.95, 1.05) if suf == "w_0" else uniform(seed, n, -0.05, 0.05)
    elif base.startswith("layer_norm"):
        v = uniform(seed, n, 0.9, 1.1) if suf == "w_0" else uniform(seed, n, -0.05, 0.05)
    elif suf == "b_0":
        v = uniform(seed, n, -0.05, 0.05)
    else:
        if base.startswith("linear") or base.startswith("fc"):
            fan_in = dims[0]
        elif base.startswith("conv2d_transpose"):
            fan_in = dims[0]
        else:
            fan_in = dims[1] * dims[2] * dims[3]
        v = normal(seed, n, np.sqrt(2.0 / fan_in))
    return v.astype(np.float32).reshape(dims)


def synth_params(pdmodel_path):
    prog = Program(pdmodel_path)
    return {n: synth_tensor(n, prog.vars[n]["dims"]) for n in prog.persistable_names()}


def ensure(root=None, force=False):
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = []
    for net in ("det", "rec"):
        dst = os.path.join(root, "models", net, "synthetic.pdiparams")
        if force or not os.path.exists(dst):
            write_params(dst, synth_params(os.path.join(root, "models", net, "inference.pdmodel")))
        out.append(dst)
    return out


if __name__ == "__main__":
    for p in ensure(force=True):
        print(p, os.path.getsize(p))
