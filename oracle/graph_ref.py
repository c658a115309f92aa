"""TEST INFRASTRUCTURE — not part of the product path.

Unfused, op-by-op interpretation of a Paddle inference graph with torch (CPU, fp32,
NCHW), used as a *second opinion* on the fused plan and on the C oracle
(SURVEY.md §8c G5).  It follows the op semantics Paddle documents for each operator
(SURVEY.md Appendix A); it does not share code with the plan executor.

Only tests/ and bench.py's cpu_baseline leg (tools/cpu_baseline_worker.py: torch-CPU/oneDNN as the closest available
stand-in for Paddle+MKLDNN kernel quality, SURVEY.md 8d "B2") may import this module.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
from pdmodel import Program  # noqa: E402


def _bcast(y, x, axis):
    """Paddle elementwise broadcasting of y onto x starting at `axis`."""
    if y.dim() == x.dim() or y.numel() == 1:
        return y
    if axis == -1:
        axis = x.dim() - y.dim()
    shape = [1] * x.dim()
    for i, d in enumerate(y.shape):
        shape[axis + i] = d
    return y.reshape(shape)


_CACHE = {}


def _prepared(pdmodel_path, params, dtype):
    """Parsed program + parameter tensors, kept per (graph, parameter dict, dtype): repeated runs (the CPU baseline,
    the float64 arbiter) do not re-parse the protobuf or re-wrap 200 arrays."""
    key = (pdmodel_path, id(params), dtype)
    if key not in _CACHE:
        if len(_CACHE) > 8:
            _CACHE.clear()
        _CACHE[key] = (Program(pdmodel_path), {k: torch.from_numpy(np.ascontiguousarray(v)).to(dtype) for k, v in params.items()})
    return _CACHE[key]


def run_graph(pdmodel_path, params, x, taps=None, dtype=torch.float32):
    """params: name -> ndarray.  x: ndarray [N,3,H,W].  Returns output ndarray (of `dtype`: torch.float64 makes this
    interpreter the arbiter between the f32 oracle and f32 torch).
    taps: optional dict var_name -> None, filled with ndarrays for requested vars."""
    prog, env0 = _prepared(pdmodel_path, params, dtype)
    env = dict(env0)
    shapes = {}
    with torch.no_grad():
        for op in prog.ops:
            t = op.type
            a = op.attrs
            if t == "feed":
                env[op.out("Out")] = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dtype)
            elif t == "fetch":
                out = env[op.inp("X")]
            elif t in ("conv2d", "depthwise_conv2d"):
                env[op.out("Output")] = F.conv2d(env[op.inp("Input")], env[op.inp("Filter")], None,
                                                 stride=a["strides"], padding=a["paddings"],
                                                 dilation=a.get("dilations", [1, 1]), groups=a["groups"])
            elif t == "conv2d_transpose":
                env[op.out("Output")] = F.conv_transpose2d(env[op.inp("Input")], env[op.inp("Filter")], None,
                                                           stride=a["strides"], padding=a["paddings"])
            elif t == "batch_norm":
                env[op.out("Y")] = F.batch_norm(env[op.inp("X")], env[op.inp("Mean")], env[op.inp("Variance")],
                                                env[op.inp("Scale")], env[op.inp("Bias")], False, 0.0,
                                                a["epsilon"])
            elif t == "elementwise_add":
                xx, yy = env[op.inp("X")], env[op.inp("Y")]
                env[op.out("Out")] = xx + _bcast(yy, xx, a.get("axis", -1)) if yy.dim() <= xx.dim() \
                    else _bcast(xx, yy, a.get("axis", -1)) + yy
            elif t == "elementwise_mul":
                xx, yy = env[op.inp("X")], env[op.inp("Y")]
                env[op.out("Out")] = xx * _bcast(yy, xx, a.get("axis", -1)) if yy.dim() <= xx.dim() \
                    else _bcast(xx, yy, a.get("axis", -1)) * yy
            elif t == "relu":
                env[op.out("Out")] = F.relu(env[op.inp("X")])
            elif t == "hard_swish":
                v = env[op.inp("X")]
                env[op.out("Out")] = v * torch.clamp(v + a["offset"], 0.0, a["threshold"]) / a["scale"]
            elif t == "hard_sigmoid":
                v = env[op.inp("X")]
                env[op.out("Out")] = torch.clamp(v * a["slope"] + a["offset"], 0.0, 1.0)
            elif t == "swish":
                v = env[op.inp("X")]
                env[op.out("Out")] = v * torch.sigmoid(v)
            elif t == "sigmoid":
                env[op.out("Out")] = torch.sigmoid(env[op.inp("X")])
            elif t == "pool2d":
                v = env[op.inp("X")]
                if a.get("adaptive"):
                    assert a["ksize"] == [1, 1]
                    env[op.out("Out")] = v.mean(dim=(2, 3), keepdim=True)
                elif a["pooling_type"] == "max":
                    env[op.out("Out")] = F.max_pool2d(v, a["ksize"], a["strides"], a["paddings"])
                else:
                    kh, kw = a["ksize"]
                    if v.shape[2] < kh:  # rec H=28 quirk (SURVEY §A.2 note): truncating output size, exclusive
                        assert (v.shape[2] - kh) // 1 < 0 and a["strides"][0] == kh
                        vv = v.mean(dim=2, keepdim=True)
                        env[op.out("Out")] = F.avg_pool2d(vv, (1, kw), (1, a["strides"][1]))
                    else:
                        env[op.out("Out")] = F.avg_pool2d(v, a["ksize"], a["strides"], a["paddings"],
                                                          count_include_pad=not a.get("exclusive", True))
            elif t == "nearest_interp_v2":
                s = a["scale"]
                env[op.out("Out")] = F.interpolate(env[op.inp("X")], scale_factor=(s[0], s[1]), mode="nearest")
            elif t == "concat":
                env[op.out("Out")] = torch.cat([env[n] for n in op.inputs["X"]], dim=a["axis"])
            elif t == "assign" or t == "dropout":
                env[op.out("Out")] = env[op.inp("X")]
            elif t == "shape":
                shapes[op.out("Out")] = list(env[op.inp("Input")].shape)
            elif t == "slice":
                src = op.inp("Input")
                if src in shapes:
                    shapes[op.out("Out")] = shapes[src][a["starts"][0]]
                else:
                    v = env[src]
                    assert a["axes"] == [0] and a["decrease_axis"] == [0]
                    env[op.out("Out")] = v[a["starts"][0]]
            elif t == "fill_constant":
                shapes[op.out("Out")] = int(float(a["str_value"])) if a.get("str_value") else int(a.get("value", 0))
            elif t == "flatten_contiguous_range":
                env[op.out("Out")] = torch.flatten(env[op.inp("X")], a["start_axis"], a["stop_axis"])
            elif t == "transpose2":
                env[op.out("Out")] = env[op.inp("X")].permute(*a["axis"]).contiguous()
            elif t == "reshape2":
                v = env[op.inp("X")]
                if op.inputs.get("ShapeTensor"):
                    shp = []
                    for i, n in enumerate(op.inputs["ShapeTensor"]):
                        s = shapes[n]
                        shp.append(v.shape[i] if s == 0 and "fill_constant" in n else s)
                else:
                    shp = [v.shape[i] if s == 0 else (-1 if s >= 2 ** 31 else s) for i, s in enumerate(a["shape"])]
                env[op.out("Out")] = v.reshape(shp)
            elif t == "squeeze2":
                env[op.out("Out")] = env[op.inp("X")].squeeze(a["axes"][0])
            elif t == "layer_norm":
                v = env[op.inp("X")]
                nd = v.shape[a["begin_norm_axis"]:]
                env[op.out("Y")] = F.layer_norm(v, nd, env[op.inp("Scale")], env[op.inp("Bias")], a["epsilon"])
            elif t == "matmul_v2":
                xx, yy = env[op.inp("X")], env[op.inp("Y")]
                if a.get("trans_x"):
                    xx = xx.transpose(-1, -2)
                if a.get("trans_y"):
                    yy = yy.transpose(-1, -2)
                env[op.out("Out")] = torch.matmul(xx, yy)
            elif t == "scale":
                assert a.get("bias", 0.0) in (0.0, None)
                env[op.out("Out")] = env[op.inp("X")] * a["scale"]
            elif t == "softmax":
                env[op.out("Out")] = torch.softmax(env[op.inp("X")], dim=a["axis"])
            else:
                raise NotImplementedError(t)
            if taps is not None:
                for k, args in op.outputs.items():
                    for n in args:
                        if n in taps and n in env:
                            taps[n] = env[n].numpy().copy()
    return out.numpy()
