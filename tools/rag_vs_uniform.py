"""Per-kernel time of the recognizer on the SAME 1024 lines of width 320 as a uniform batch (RAG = false kernels) and as
a ragged batch (RAG = true): what the ragged instantiations cost.  python tools/rag_vs_uniform.py [lines]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
x = np.random.RandomState(1).randn(N, 48, 320, 3).astype(np.float32)
def parse(rep):
    return {k.split("@")[0]: v["ms"] / max(v["count"], 1) for k, v in rep.items()}
res = {}
for mode in ("uniform", "ragged"):
    g = pkg.Net("rec")
    run = (lambda: g.forward(x, keep_all=False)) if mode == "uniform" else (lambda: g.forward_ragged(list(x), keep_all=False))
    run(); run()
    g.timing(True)
    for _ in range(3): run()
    rep = g.timing_report()
    res[mode] = parse(rep)
    g.close()
tot = {m: 0.0 for m in res}
for k in res["uniform"]:
    u, r = res["uniform"][k], res["ragged"].get(k, float("nan"))
    tot["uniform"] += u; tot["ragged"] += r if r == r else 0
    if u > 0.05: print("%-40s uniform %.3f  ragged %.3f  %+.1f%%" % (k, u, r, 100 * (r / u - 1)))
print("total", tot)
