"""Round-5 GPU tests (through the C-ABI): the bind-time probe of the folded concat, the LAB fold's stage lists as the device
executes them, the OpenCV-rule blast radius on the benchmark batch, the worker pool on two logical devices."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_concat_fold_refused_at_bind_keeps_the_materialised_concat(pkg, built):
    """the DB neck's concat is folded into the 3x3 conv's tile fill only when that kernel's launcher
    accepts the shape AT BIND TIME (shape checks + the device's dynamic-LDS attribute).  Refused there - injected here as a
    failed attribute would refuse it - the concat stays a launch of its own and the conv keeps its fallback chain: the run
    succeeds and the output is the oracle's."""
    from oracle import OracleNet
    L = pkg.lib()
    x = np.random.RandomState(5).randn(2, 96, 160, 3).astype(np.float32)
    want = OracleNet("det").run(x)
    names = {}
    for refuse in (None, b"conv3x3_c24@bind"):
        assert L.ocr_selftest_refuse_launch(refuse) == 0
        try:
            g = pkg.Net("det")
            g.timing(True)
            got = g.forward(x, keep_all=False)
            names[refuse] = sorted(g.timing_report())
            g.close()
        finally:
            L.ocr_selftest_refuse_launch(None)
        assert np.array_equal(want, got), refuse
    assert any("_cat4" in n for n in names[None]) and not any(".concat_" in n for n in names[None])
    assert any(".concat_" in n for n in names[b"conv3x3_c24@bind"]) and not any("_cat4" in n for n in names[b"conv3x3_c24@bind"])


def test_lab_fold_stage_lists_on_the_device(pkg, built):
    """The round-5 arithmetic contract: the loader folds `bias | *s0 | +a0 [| hswish | *s1 | +a1]` into the
    weights, one bias and one fma; a depthwise conv whose only reader is a 1x1 conv hands its scale / shift to that conv.
    Every materialised tensor of the unfused launch list (keep_all = 1: the absorbed depthwise tensors hold the bare
    hard-swish product there) equals the oracle's, which folds with the same operations."""
    from oracle import OracleNet
    for kind, shape in (("rec", (2, 48, 96)), ("det", (1, 64, 96))):
        x = np.random.RandomState(11).randn(shape[0], shape[1], shape[2], 3).astype(np.float32)
        o, g = OracleNet(kind), pkg.Net(kind)
        want = o.run(x)
        got = g.forward(x, keep_all=True)
        assert np.array_equal(want, got)
        checked = 0
        for tid in range(1, g.num_tensors()):
            if not g.exists(tid):
                continue
            a, b = g.fetch(tid), o.tensor(tid)
            assert a.size == b.size and np.array_equal(a.reshape(-1), b.reshape(-1)), (kind, tid)
            checked += 1
        assert checked >= 50
        g.close()
