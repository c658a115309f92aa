"""Per-kernel timing of one network alone (no concurrent lanes): python tools/net_bench.py rec 1024 48 320 [reps]
Writes the table to stdout: name, ms, TFLOP/s, algorithmic GB/s.  Development aid, not the bench contract."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from __graft_entry__ import load_package  # noqa: E402


def main():
    kind, N, H, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
    import synth_weights
    synth_weights.ensure(ROOT)
    pkg = load_package()
    net = pkg.Net(kind)
    x = np.random.RandomState(0).uniform(-1, 1, (N, H, W, 3)).astype(np.float32)
    from cpp_paddle_ocr_amd import binding as B
    fwd = lambda: B.check(B.lib().ocr_net_forward(net.h, x.ctypes.data, N, H, W, 0))
    fwd()
    net.timing(True)
    for _ in range(reps):
        fwd()
    rep = net.timing_report()
    rows = [dict(name=k, **v) for k, v in rep.items()]
    tot = sum(r["ms"] for r in rows) / reps
    fl = sum(r["flops"] for r in rows) / reps
    by = sum(r["bytes"] for r in rows) / reps
    print("TOTAL %.3f ms/forward  %.2f TFLOP/s  %.0f GB/s(alg)  %d launches" % (tot, fl / tot / 1e9, by / tot / 1e6, len(rows)))
    for r in sorted(rows, key=lambda r: -r["ms"]):
        ms = r["ms"] / r["count"]
        print("%-44s %8.3f ms %6.2f%%  %7.2f TFLOP/s %8.1f GB/s" % (r["name"], ms, 100 * r["ms"] / reps / tot,
                                                                  r["flops"] / r["count"] / ms / 1e9, r["bytes"] / r["count"] / ms / 1e6))


if __name__ == "__main__":
    main()
