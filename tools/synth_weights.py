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


# ---- BASELINE configs[4]: the hand-written server plans (tools/make_server_plans.py; NOT reference artifacts) ----------------
def server_param_table(plan_path):
    """[(name, dims)] from the plan's `# param <name> <d0,d1,...>` lines (there is no .pdmodel to take shapes from)"""
    out = []
    for line in open(plan_path):
        if line.startswith("# param "):
            _, _, name, dims = line.split()
            out.append((name, [int(d) for d in dims.split(",")]))
    return out


def synth_server_tensor(name, dims):
    """conv / linear / deconv weights ~ N(0, sqrt(2 / fan_in)) (a sum of 4 uniforms: 67 M values in seconds), biases
    U[-0.05, 0.05], batch norm scale 1 +- 0.1 (the residual branches' last norm 0.25 +- 0.05, so that sixteen residual adds stay
    inside f16's range), offset / mean +- 0.1, variance U[0.8, 1.2], layer norm 1 +- 0.1 / +- 0.05, position embedding N(0, 0.02)"""
    seed = fnv1a(name) ^ 0x5E12F00D
    n = int(np.prod(dims))
    base, suf = name.rsplit(".", 1) if "." in name else (name, "")
    if suf == "scale":
        v = uniform(seed, n, 0.2, 0.3) if base.endswith("branch2c") else uniform(seed, n, 0.9, 1.1)
    elif suf in ("offset", "mean"):
        v = uniform(seed, n, -0.1, 0.1)
    elif suf == "variance":
        v = uniform(seed, n, 0.8, 1.2)
    elif "norm" in base and suf == "w":
        v = uniform(seed, n, 0.9, 1.1)
    elif suf == "b":
        v = uniform(seed, n, -0.05, 0.05)
    elif name == "pos_embed":
        v = (sum(_splitmix(seed, n, k + 1) for k in range(4)) - 2.0) * np.sqrt(3.0) * 0.02
    else:
        if len(dims) == 2:
            fan_in = dims[0]                      # linear [in, out]
        elif "deconv" in base:
            fan_in = dims[0]                      # [Cin, Cout, 2, 2]: one tap per output pixel
        else:
            fan_in = dims[1] * dims[2] * dims[3]  # conv [Cout, Cin, kh, kw]
        v = (sum(_splitmix(seed, n, k + 1) for k in range(4)) - 2.0) * np.sqrt(3.0) * np.sqrt(2.0 / fan_in)
    return v.astype(np.float32).reshape(dims)


def ensure_server(root=None, force=False):
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = []
    for net in ("det", "rec"):
        d = os.path.join(root, "models_server", net)
        os.makedirs(d, exist_ok=True)
        dst = os.path.join(d, "synthetic.pdiparams")
        plan = os.path.join(root, "cpp-paddle-ocr_amd", "plans", "srv_%s.plan" % net)
        if force or not os.path.exists(dst) or os.path.getmtime(dst) < os.path.getmtime(plan):
            write_params(dst, {n: synth_server_tensor(n, dims) for n, dims in server_param_table(plan)})
        out.append(dst)
    return out


if __name__ == "__main__":
    for p in ensure(force=True) + ensure_server(force=True):
        print(p, os.path.getsize(p))
