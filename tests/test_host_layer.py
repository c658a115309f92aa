"""The C++ host layer (shim classes + OCRWorker + GPUWorkerPool over the C-ABI) through its test binary,
which mirrors the reference's tests/test_ocr_worker.cpp; the JSON of the first request is compared with
the oracle pipeline."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "cpp-paddle-ocr_amd", "host")


def test_host_layer_compiles(built):
    subprocess.check_call(["make", "-s", "-C", HOST])
    assert os.path.exists(os.path.join(HOST, "test_worker"))


@pytest.mark.gpu
def test_worker_and_pool_binary(built, card, tmp_path):
    from pipeline import Pipeline
    subprocess.check_call(["make", "-s", "-C", HOST])
    raw = tmp_path / "card.bgr"
    card.tofile(raw)
    out = subprocess.run([os.path.join(HOST, "test_worker"), os.path.join(ROOT, "models"), str(raw),
                          "%dx%d" % card.shape[:2]], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "ALL OK" in out.stdout and "FAIL" not in out.stdout
    js = json.loads([l for l in out.stdout.splitlines() if l.startswith("JSON ")][0][5:])
    assert js["request_id"] == 100 and js["worker_id"] == 7 and js["success"] is True
    assert js["width"] == card.shape[1] and js["height"] == card.shape[0] and js["processing_time_ms"] > 0
    import oracle as O
    want = Pipeline().process(card)["words"]
    labels = ["#"] + open(os.path.join(ROOT, "models", "rec", "ppocr_keys_v1.txt"), encoding="utf-8").read().split("\n")
    if labels[-1] == "":
        labels.pop()
    labels.append(" ")
    assert len(js["words"]) == len(want)
    for g, w in zip(js["words"], want):
        assert g["box"] == np.asarray(w["box"]).tolist()
        assert g["text"] == "".join(labels[i] for i in w["ids"])
        assert np.float32(g["confidence"]) == np.float32(w["confidence"])


@pytest.mark.gpu
def test_pool_on_two_devices(built, card, tmp_path, pkg):
    """GPUWorkerPool with worker i on GPU i mod n, from ONE process (gpu_worker_pool.cpp:12-16 shape).
    On a node with two visible devices these are two GPUs; on a one-GPU lease the library's logical-device table
    (OCR_DEVICE_MAP=0,0, csrc/hip_guard.h) gives the pool two devices - own LDS-attribute memo entries, occupancy memos,
    priority anchors, arenas and streams per LOGICAL id - that share the physical GPU: the code path is the pool's either
    way, and every reply equals the oracle's words (= the one-worker result)."""
    from pipeline import Pipeline
    subprocess.check_call(["make", "-s", "-C", HOST])
    raw = tmp_path / "card.bgr"
    card.tofile(raw)
    env = dict(os.environ)
    if pkg.lib().ocr_rt_device_count() < 2:
        env["OCR_DEVICE_MAP"] = "0,0"
    out = subprocess.run([os.path.join(HOST, "test_worker"), os.path.join(ROOT, "models"), str(raw),
                          "%dx%d" % card.shape[:2], "multi"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "ALL OK" in out.stdout and "SKIP" not in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    want = Pipeline().process(card)["words"]
    replies = [json.loads(l[9:]) for l in out.stdout.splitlines() if l.startswith("POOLJSON ")]
    assert len(replies) == 8 and {r["worker_id"] for r in replies} == {0, 1}
    for r in replies:
        assert r["success"] is True and len(r["words"]) == len(want)
        for g, w in zip(r["words"], want):
            assert g["box"] == np.asarray(w["box"]).tolist() and np.float32(g["confidence"]) == np.float32(w["confidence"])


def test_device_map_is_parsed_without_a_gpu(built):
    """OCR_DEVICE_MAP is read once per process; with no device visible the count stays 0 whatever the map says (a CPU
    container), and a malformed map is ignored rather than trusted."""
    import subprocess as sp
    code = ("import sys; sys.path.insert(0, %r); import __graft_entry__ as g; L = g.load_package().lib(); print('COUNT', L.ocr_rt_device_count())" % ROOT)
    for m in ("0,0", "0,x", "", "7"):
        out = sp.run([sys.executable, "-c", code], env=dict(os.environ, OCR_DEVICE_MAP=m), capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-1000:]
        n = int(out.stdout.split("COUNT")[1])
        import torch
        if not torch.cuda.is_available():
            assert n == 0
        else:
            assert n == (2 if m == "0,0" else torch.cuda.device_count())


def test_argsort_tie_order_modes(built):
    """Utility::argsort is std::sort (utility.cpp:192-203): the order of EQUAL ratios is the host library's.  "std" is
    this toolchain's std::sort; "stable" keeps ties in input order, which is what MSVC's std::sort (the reference's
    toolchain) does for up to 32 elements (insertion sort).  Both sort correctly; they may differ only on ties."""
    import numpy as np
    import oracle as O
    rs = np.random.RandomState(4)
    for n in (1, 5, 16, 17, 33, 40, 200):
        v = rs.randint(0, 6, n).astype(np.float32) * 0.5       # many ties
        a, b = O.argsort(v), O.argsort(v, stable=True)
        assert sorted(a.tolist()) == sorted(b.tolist()) == list(range(n))
        assert (np.diff(v[a]) >= 0).all() and (np.diff(v[b]) >= 0).all()
        for x in np.unique(v):                                  # stable: ties in input order
            idx = b[v[b] == x]
            assert (np.diff(idx) > 0).all()
    v = rs.rand(64).astype(np.float32)                          # no ties: one answer
    assert np.array_equal(O.argsort(v), O.argsort(v, stable=True))
