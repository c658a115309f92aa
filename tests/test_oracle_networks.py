"""The fused plan + the C oracle's network executor against an independent, unfused torch-CPU
interpretation of the same Paddle graphs.  cls runs on the reference's REAL weights."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plans_are_current(built, tmp_path):
    """cpp-paddle-ocr_amd/plans/*.plan are what tools/make_plan.py generates from the model graphs."""
    import make_plan
    from pdmodel import Program
    for net in ("det", "rec", "cls"):
        plan, chans = make_plan.lower(Program(os.path.join(ROOT, "models", net, "inference.pdmodel")))
        text = open(os.path.join(ROOT, "cpp-paddle-ocr_amd", "plans", net + ".plan")).read().splitlines()
        body = [l for l in text if not l.startswith("#") and not l.startswith("plan ")]
        assert body == [p.line() for p in plan]


def test_cls_params_consumed_exactly(built):
    from pdmodel import Program, read_params
    p = Program(os.path.join(ROOT, "models", "cls", "inference.pdmodel"))
    w = read_params(os.path.join(ROOT, "models", "cls", "inference.pdiparams"), p.persistable_names())
    assert len(w) == 213 and sum(v.size for v in w.values()) == 133628


@pytest.mark.parametrize("net,shape,tol", [
    ("cls", (3, 3, 48, 192), 2e-6),
    ("det", (1, 3, 96, 160), 2e-4),
    ("rec", (2, 3, 48, 160), 1e-6),
    ("rec", (1, 3, 28, 192), 1e-6),   # the worker's H=28 on a graph exported for 48 (SURVEY A.2 note)
])
def test_oracle_net_matches_torch_graph(built, net, shape, tol):
    from oracle import OracleNet
    from graph_ref import run_graph
    o = OracleNet(net)
    x = np.random.RandomState(1).randn(*shape).astype(np.float32)
    y = o.run(x.transpose(0, 2, 3, 1))
    ref = run_graph(os.path.join(ROOT, "models", net, "inference.pdmodel"), o.weights, x)
    yy = y.transpose(0, 3, 1, 2) if net == "det" else y.reshape(ref.shape)
    assert np.abs(yy - ref).max() <= tol


def test_oracle_expf_accuracy(built):
    import oracle as O
    xs = np.linspace(-80, 80, 4001).astype(np.float32)
    got = np.array([O.lib().oracle_expf(float(v)) for v in xs], np.float64)
    ref = np.exp(xs.astype(np.float64))
    assert np.max(np.abs(got - ref) / ref) < 3e-7
