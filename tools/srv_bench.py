"""Per-launch table of a server network (BASELINE configs[4]) at its benchmark shape: ms, TFLOP/s, algorithmic GB/s.
    python tools/srv_bench.py det|rec [N] [precision] [iters]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from __graft_entry__ import load_package  # noqa: E402


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "det"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else (32 if kind == "det" else 1024)
    prec = sys.argv[3] if len(sys.argv) > 3 else "fp16"
    iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    import synth_weights
    synth_weights.ensure_server(ROOT)
    pkg = load_package()
    h, w = (960, 960) if kind == "det" else (48, 320)
    x = np.random.RandomState(0).randn(n, h, w, 3).astype(np.float32)
    net = pkg.SrvNet(kind, prec)
    t0 = time.time()
    net.forward(x)
    print("first forward (bind + tune): %.2f s" % (time.time() - t0), flush=True)
    net.rerun(2)
    t0 = time.time()
    net.rerun(iters)
    wall = (time.time() - t0) / iters
    net.timing(True)
    net.rerun(iters)
    rep = net.timing_report()
    tot_ms = sum(v["ms"] for v in rep.values()) / iters
    tot_fl = sum(v["flops"] for v in rep.values()) / iters
    tot_by = sum(v["bytes"] for v in rep.values()) / iters
    rows = sorted(rep.items(), key=lambda kv: int(kv[0].split(".")[0]) if kv[0][0].isdigit() else -1)
    print("%-64s %9s %9s %9s" % ("launch", "ms", "TFLOP/s", "GB/s"))
    for name, v in rows:
        ms = v["ms"] / v["count"]
        print("%-64s %9.4f %9.1f %9.0f" % (name, ms, v["flops"] / v["count"] / ms / 1e9, v["bytes"] / v["count"] / ms / 1e6))
    print("TOTAL %s N=%d %s: wall %.3f ms/run, kernels %.3f ms, %.2f TFLOP -> %.1f TFLOP/s (%.3f of 2.5 PF), %.2f GB -> %.0f GB/s" % (
        kind, n, prec, wall * 1e3, tot_ms, tot_fl / 1e12, tot_fl / tot_ms / 1e9, tot_fl / tot_ms / 1e9 / 2500, tot_by / 1e9, tot_by / tot_ms / 1e6))


if __name__ == "__main__":
    main()
