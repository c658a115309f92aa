"""debug: per-tensor mismatch table of a server network's f32 twin (or fp16) against the oracle.  usage: kind N H W [precision]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from __graft_entry__ import load_package
import synth_weights
from oracle import OracleNet
kind = sys.argv[1]; n, h, w = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]); prec = sys.argv[5] if len(sys.argv) > 5 else "fp32"
synth_weights.ensure_server(ROOT)
pkg = load_package()
x = np.random.RandomState(11).randn(n, h, w, 3).astype(np.float32)
ora = OracleNet("srv_" + kind); ora.run(x)
net = pkg.SrvNet(kind, prec); net.forward(x, keep_all=True)
net.timing(True); net.rerun(1); names = sorted(net.timing_report(), key=lambda k: int(k.split(".")[0]) if k[0].isdigit() else -1)
for tid in range(1, net.num_tensors()):
    want = ora.tensor(tid); got = net.fetch(tid, cap=want.size + 16)
    bad = (got.view(np.uint32) != want.view(np.uint32))
    if bad.any():
        idx = np.argwhere(bad)
        print("tid", tid, want.shape, "mismatch", int(bad.sum()), "of", bad.size, "max|d|", float(np.abs(got - want).max()), "first", idx[0].tolist(), "chan hist", np.bincount(idx[:, 3] // 8, minlength=want.shape[3] // 8 + 1)[:16].tolist(), names[tid] if tid < len(names) else "")
print("done")
if len(sys.argv) > 6:
    tid = int(sys.argv[6])
    want = ora.tensor(tid); got = net.fetch(tid, cap=want.size + 16)
    bad = (got.view(np.uint32) != want.view(np.uint32)).reshape(-1, want.shape[3])
    for p in np.argwhere(bad.any(1)).ravel()[:40]:
        print("pixel", int(p), "bad chans", np.argwhere(bad[p]).ravel().tolist()[:70])
