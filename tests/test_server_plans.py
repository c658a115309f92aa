"""BASELINE configs[4] on the CPU: the hand-written server plans (NOT reference artifacts: tools/make_server_plans.py), the oracle
ops they add, and an independent float64 torch interpretation of the same plans as a second opinion on the oracle."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLANS = os.path.join(ROOT, "cpp-paddle-ocr_amd", "plans")


def test_committed_plans_are_what_the_generator_writes(tmp_path):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_server_plans", os.path.join(ROOT, "tools", "make_server_plans.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    for P in (m.det_plan(), m.rec_plan()):
        assert P.text() == open(os.path.join(PLANS, P.name + ".plan")).read(), P.name
    # ResNet50-vd: 16 bottlenecks; SVTR-L: 3 + 9 + 9 mixing blocks, ten of them local
    det = open(os.path.join(PLANS, "srv_det.plan")).read()
    rec = open(os.path.join(PLANS, "srv_rec.plan")).read()
    assert det.count("w=res") == 52 and det.count("_branch2c.w ep=") == 16 and det.count("type=avg") == 3 and det.count("addup:") == 3
    assert rec.count("\nattn ") == 21 and rec.count("lh=7 lw=11") == 10 and "cout=6625" in rec
    assert "NOT a reference artifact" in det and "NOT a reference artifact" in rec


def test_erf_of_the_gelu(built):
    """ocr_erff (Abramowitz-Stegun 7.1.26 with the contract's exp): within 5e-7 of the real erf (the formula's own 1.5e-7 plus f32 rounding) - the plans' act:gelu is the
    exact GELU to f32 accuracy"""
    import ctypes as C
    import oracle as O
    L = O.lib()
    if not hasattr(L, "oracle_erff"):
        pytest.skip("oracle built without the probe")
    L.oracle_erff.restype = C.c_float
    L.oracle_erff.argtypes = [C.c_float]
    xs = np.concatenate([np.linspace(-6, 6, 4001), [0.0, -0.0, 1e-8, 30.0, -30.0]]).astype(np.float32)
    err = max(abs(L.oracle_erff(float(x)) - math.erf(float(x))) for x in xs)
    assert err <= 5e-7, err


def _torch_run(plan_text, params, x):
    """float64 torch interpretation of a server plan, op by op, written from the plan grammar alone"""
    import torch
    import torch.nn.functional as F
    t = {0: torch.from_numpy(x.astype(np.float64)).permute(0, 3, 1, 2)}
    P = {k: torch.from_numpy(np.asarray(v, np.float64)) for k, v in params.items()}

    def ep(y, stages):
        for st in stages:
            k, _, a = st.partition(":")
            a = a.split(",")
            if k == "bias":
                y = y + P[a[0]].view(1, -1, 1, 1)
            elif k == "bn":
                g, b, m, v = (P[n] for n in a[:4])
                y = (y - m.view(1, -1, 1, 1)) / torch.sqrt(v.view(1, -1, 1, 1) + float(a[4])) * g.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)
            elif k == "addt":
                y = y + t[int(a[0])]
            elif k == "addup":
                y = y + F.interpolate(t[int(a[0])], scale_factor=int(a[1]), mode="nearest")
            elif k == "addpos":
                n, c, h, w = y.shape
                y = y + P[a[0]].view(1, h, w, c).permute(0, 3, 1, 2)
            elif k == "act":
                y = {"relu": F.relu, "gelu": lambda z: F.gelu(z), "hswish": F.hardswish, "sigmoid": torch.sigmoid}[a[0]](y)
            else:
                raise ValueError(k)
        return y

    out = None
    for line in plan_text.splitlines():
        if not line or line[0] == "#" or line.startswith("plan "):
            continue
        toks = line.split()
        kind, kv = toks[0], dict(tk.split("=", 1) for tk in toks[1:])
        stages = kv.get("ep", "").split("|") if kv.get("ep") else []
        gi = lambda k, d=0: int(kv.get(k, d))
        if kind == "output":
            out = t[gi("i")]
            continue
        o = gi("o")
        if kind == "conv":
            y = F.conv2d(t[gi("i")], P[kv["w"]], stride=(gi("sh"), gi("sw")), padding=(gi("ph"), gi("pw")))
        elif kind == "linear":
            y = torch.einsum("nchw,co->nohw", t[gi("i")], P[kv["w"]])
        elif kind == "deconv":
            y = F.conv_transpose2d(t[gi("i")], P[kv["w"]], stride=2)
        elif kind == "pool":
            a = dict(kernel_size=(gi("kh"), gi("kw")), stride=(gi("sh"), gi("sw")), padding=(gi("ph"), gi("pw")))
            y = F.max_pool2d(t[gi("i")], **a) if kv["type"] == "max" else F.avg_pool2d(t[gi("i")], count_include_pad=False, **a)
        elif kind == "concat":
            ids, ups = [int(v) for v in kv["i"].split(",")], [int(v) for v in kv["up"].split(",")]
            y = torch.cat([F.interpolate(t[i], scale_factor=u, mode="nearest") if u > 1 else t[i] for i, u in zip(ids, ups)], 1)
        elif kind == "ew":
            y = t[gi("i")]
        elif kind == "ln":
            z = t[gi("i")].permute(0, 2, 3, 1)
            y = F.layer_norm(z, z.shape[-1:], P[kv["g"]], P[kv["b"]], float(kv["eps"])).permute(0, 3, 1, 2)
        elif kind == "attn":
            z = t[gi("i")]
            n, c3, h, w = z.shape
            heads, hd = gi("heads"), gi("hd")
            T = h * w
            q, k, v = z.permute(0, 2, 3, 1).reshape(n, T, 3, heads, hd).permute(2, 0, 3, 1, 4)
            s = (q * float(kv["scale"])) @ k.transpose(-1, -2)
            lh, lw = gi("lh"), gi("lw")
            if lh > 0:  # SVTR's Local mixer: -inf outside the window (rec_svtrnet.py builds this mask by slicing a padded grid)
                yy, xx = np.divmod(np.arange(T), w)
                ok = (np.abs(yy[:, None] - yy[None]) <= lh // 2) & (np.abs(xx[:, None] - xx[None]) <= lw // 2)
                s = s.masked_fill(~torch.from_numpy(ok), float("-inf"))
            y = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(n, h, w, heads * hd).permute(0, 3, 1, 2)
        else:
            raise ValueError(kind)
        t[o] = ep(y, stages)
    return out.permute(0, 2, 3, 1).numpy(), {k: v.permute(0, 2, 3, 1).numpy() for k, v in t.items() if k}


@pytest.mark.parametrize("net,shape", [("srv_det", (1, 64, 96)), ("srv_rec", (1, 48, 320))])
def test_oracle_matches_an_independent_torch_interpretation(built, net, shape):
    """the oracle's f32 run of a server plan (the arithmetic the device's f32 twin reproduces bit for bit) against torch float64 on
    the same seeded parameters: every tensor within 2e-4 of its own scale, the output (probability map / CTC logits) within 2e-4"""
    import oracle as O
    x = np.random.RandomState(2).randn(shape[0], shape[1], shape[2], 3).astype(np.float32)
    o = O.OracleNet(net)
    got = o.run(x)
    want, tensors = _torch_run(O.plan_text(net), o.weights, x)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 2e-4, float(np.abs(got - want).max())
    worst = 0.0
    for tid, w in tensors.items():
        g = o.tensor(tid)
        assert g.shape == w.shape, (tid, g.shape, w.shape)
        worst = max(worst, float(np.abs(g - w).max() / (np.abs(w).max() + 1e-9)))
    assert worst <= 2e-4, worst


def test_server_weights_are_seeded_and_in_the_pdiparams_format(built):
    import synth_weights
    import oracle as O
    paths = synth_weights.ensure_server(ROOT)
    for kind, p in zip(("det", "rec"), paths):
        table = synth_weights.server_param_table(os.path.join(PLANS, "srv_%s.plan" % kind))
        w = O.load_weights("srv_" + kind)
        assert set(w) == {n for n, _ in table}
        n0, d0 = table[0]
        assert np.array_equal(w[n0], synth_weights.synth_server_tensor(n0, d0))
        assert os.path.getsize(p) > sum(int(np.prod(d)) for _, d in table) * 4
