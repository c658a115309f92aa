"""Round-6 GPU tests (through the C-ABI): the device's ClipperOffset / UnClip in front of the reference's own vectors."""
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_clipper_offset_on_the_reference_vectors(pkg, built):
    """the HIP `clipper_offset_round` (csrc/kernels_post.hip; ocml's f64 sin / cos / acos / atan2, not the
    host library's) against tests/golden/unclip_ref_20k.npz and unclip_ref.json - outputs of the REFERENCE's compiled
    src/clipper.cpp (/root/reference/src/clipper.cpp:3779-4021 through the call of postprocess_op.cpp:46-55) - point for
    point.  The sliver class (repeated corner, 0 < delta < 0.5: needs the full Vatti union) is excluded by its exact predicate
    and counted.  Also counted: deltas on which the device's acos / sin / cos differ from the host's BEFORE Round() - the
    distance between the two math libraries that the integer outputs then do or do not show."""
    import json
    from test_oracle_unclip import _golden20k, is_sliver
    assert pkg.lib().ocr_rt_init(0) == 0
    quads, deltas, paths, counts, points, off = _golden20k()
    small = json.load(open(os.path.join(ROOT, "tests", "golden", "unclip_ref.json")))
    q2 = np.array([np.array(c["quad"], np.int32).ravel() for c in small], np.int32)
    d2 = np.array([c["delta"] for c in small], np.float64)
    allq = np.concatenate([quads, q2])
    alld = np.concatenate([deltas, d2])
    want = [points[off[i]:off[i + 1]].astype(np.int64) for i in range(len(deltas))] + \
           [np.array(c["out"], np.int64).reshape(-1, 2) for c in small]
    out, cnt, trig = pkg.selftest_unclip(allq, alld, cap=64)
    assert (cnt >= 0).all()
    bad, slivers = [], 0
    for i in range(len(alld)):
        got = out[i, :cnt[i]]
        if got.shape != want[i].shape or not np.array_equal(got, want[i]):
            if is_sliver(allq[i], alld[i]):
                slivers += 1
            else:
                bad.append(i)
    assert not bad, (len(bad), allq[bad[0]].tolist(), alld[bad[0]])
    assert slivers <= 10, slivers
    # the math libraries before rounding (host: what this Python's libm gives, i.e. glibc here)
    ndiff = 0
    nz = 0
    for i in range(len(alld)):
        d = alld[i]
        if -1e-20 < d < 1e-20:
            continue
        nz += 1
        y = min(0.25, abs(d) * 0.25)
        steps = math.pi / math.acos(1 - y / abs(d))
        steps = min(steps, abs(d) * math.pi)
        host = (steps, math.sin(2 * math.pi / steps), math.cos(2 * math.pi / steps))
        ndiff += tuple(trig[i]) != host
    print("unclip: %d cases equal the reference's vectors, %d slivers excluded; device f64 trig differs from the host's on %d of %d deltas"
          % (len(alld) - slivers, slivers, ndiff, nz))
    rel = np.abs(trig[:, 1:] - np.array([[math.sin(2 * math.pi / t[0]) if t[0] else 0, math.cos(2 * math.pi / t[0]) if t[0] else 0] for t in trig]))
    assert rel.max() < 1e-15  # an ulp, never more: Round() of coordinate + normal * delta hides it unless a tie is hit


def test_device_unclip_box_equals_the_oracle(pkg, built):
    """the whole UnClip -> cv::minAreaRect -> GetMiniBoxes of a candidate (border_box_kernel's `unclip_min_rect`, the code the
    detector runs) on 20 000 boxes at both ratios: RotatedRect, ssid and corners bit-identical to the oracle, whose
    ClipperOffset is the one pinned above"""
    import ctypes as C
    import oracle as O
    assert pkg.lib().ocr_rt_init(0) == 0
    rs = np.random.RandomState(606)
    n = 10000
    boxes = np.zeros((n, 8), np.float32)
    for i in range(n):
        cx, cy = rs.rand(2) * 900 + 30
        w, h = rs.rand() * 300 + 1, rs.rand() * 60 + 0.5
        ang = (rs.rand() - 0.5) * (math.pi if i % 3 == 0 else 0.2)
        ca, sa = math.cos(ang), math.sin(ang)
        pts = [(cx + sx * w / 2 * ca - sy * h / 2 * sa, cy + sx * w / 2 * sa + sy * h / 2 * ca) for sx, sy in ((-1, -1), (1, -1), (1, 1), (-1, 1))]
        if i % 47 == 0:
            pts[2] = pts[1]
        boxes[i] = np.array(pts, np.float32).ravel()
    lib = O.lib()
    lib.oracle_unclip_box.argtypes = [C.c_void_p, C.c_float, C.c_void_p]
    for ratio in (1.8, 2.0):
        got, st = pkg.selftest_unclip_box(boxes, ratio)
        assert (st[:, 0] == 0).all()
        want = np.zeros((n, 14), np.float32)
        npoly = np.zeros(n, np.int32)
        for i in range(n):
            npoly[i] = lib.oracle_unclip_box(boxes[i].ctypes.data, ratio, want[i].ctypes.data)
        assert np.array_equal(st[:, 1], npoly)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), int((got.view(np.uint32) != want.view(np.uint32)).any(axis=1).sum())


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs[4]: the server networks (hand-written plans, NOT reference artifacts; tools/make_server_plans.py)
# ---------------------------------------------------------------------------------------------------------------------
def _srv_ready():
    import synth_weights
    synth_weights.ensure_server(ROOT)


def _compare_all(net, ora, ntensors, exact, tol_fn=None):
    worst = {}
    for tid in range(1, ntensors):
        want = ora.tensor(tid)
        got = net.fetch(tid, cap=want.size + 16)
        assert got.shape == want.shape, (tid, got.shape, want.shape)
        if exact:
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (tid, float(np.abs(got - want).max()))
        else:
            worst[tid] = tol_fn(tid, got, want)
    return worst


@pytest.mark.parametrize("kind,shape", [("det", (2, 64, 96)), ("rec", (2, 48, 320))])
def test_server_net_f32_twin_equals_the_oracle(pkg, built, kind, shape):
    """precision "fp32" of the server networks = the parity twin: the SAME launch list (implicit-GEMM family, pools, concat,
    layer norm, attention, map head) on float, every contraction the oracle's ascending-k fma chain on
    v_mfma_f32_32x32x2_f32 - EVERY plan tensor equals the oracle's bit for bit.  This pins indexing, padding, strides, the
    swizzled LDS images, the DMA plans and the epilogue order exactly; the fp16 build differs by rounding only."""
    from oracle import OracleNet
    _srv_ready()
    n, h, w = shape
    x = np.random.RandomState(11).randn(n, h, w, 3).astype(np.float32)
    ora = OracleNet("srv_" + kind)
    ora.run(x)
    net = pkg.SrvNet(kind, "fp32")
    net.forward(x, keep_all=True)
    _compare_all(net, ora, net.num_tensors(), exact=True)
    net.close()


@pytest.mark.parametrize("kind,shape", [("det", (2, 64, 96)), ("rec", (2, 48, 320))])
def test_server_net_fp16_within_tolerance_of_the_oracle(pkg, built, kind, shape):
    """precision "fp16" (the mode BASELINE configs[4] names; /root/reference/src/ocr_det.cpp:50-57): every plan tensor
    against the f32 oracle under the section-9 style tolerances - relative to the tensor's own scale (f16 storage rounds
    every activation to 11 bits, errors accumulate over 50 / 21 layers): max |d| <= 6 % of the tensor's max |value|, mean
    |d| <= 1 % of its mean |value|; det probability map |d| <= 2e-2 on 99 % of the pixels; rec arg max equal on >= 90 % of
    the steps (6625 nearly equal logits under seeded weights)."""
    from oracle import OracleNet
    _srv_ready()
    n, h, w = shape
    x = np.random.RandomState(12).randn(n, h, w, 3).astype(np.float32)
    ora = OracleNet("srv_" + kind)
    ora.run(x)
    net = pkg.SrvNet(kind, "fp16")
    net.forward(x, keep_all=True)

    def tol(tid, got, want):
        d = np.abs(got - want)
        scale = float(np.abs(want).max()) + 1e-6
        mean = float(np.abs(want).mean()) + 1e-6
        assert np.isfinite(got).all(), tid
        assert d.max() <= 0.06 * scale, (tid, float(d.max()), scale)
        assert d.mean() <= 0.01 * mean + 1e-4, (tid, float(d.mean()), mean)
        return float(d.max() / scale)

    worst = _compare_all(net, ora, net.num_tensors(), exact=False, tol_fn=tol)
    out_g, out_w = net.fetch(-1), ora.tensor(-1)
    if kind == "det":
        assert (np.abs(out_g - out_w) <= 2e-2).mean() >= 0.99
    else:
        assert (out_g.argmax(-1) == out_w.argmax(-1)).mean() >= 0.90
    print("fp16 %s: worst tensor max|d|/max|x| = %.4f" % (kind, max(worst.values())))
    net.close()


def test_server_pipeline_twin_equals_the_oracle_and_fp16_stays_close(pkg, built):
    """OCRWorker::processRequest (/root/reference/src/ocr_worker.cpp:213-311) with the server networks behind the stage
    objects (model directories whose arch.txt names a server plan): the f32 TWIN pipeline's words equal the oracle pipeline's -
    boxes and CTC ids exactly, confidences to 1e-5 (the device sums the softmax in its own order) - and the fp16 pipeline (the
    mode BASELINE configs[4] names) finds the same boxes (section-8d protocol maps) with >= 90 % of the lines' id sequences
    equal."""
    import pipeline as P
    from synth_data import cfg2_sample
    _srv_ready()
    n, hw, k = 2, 320, 6
    samples = [cfg2_sample(100 + i, hw, hw, k) for i in range(n)]
    imgs = [s[0] for s in samples]
    probs = [s[1] for s in samples]
    ora = P.Pipeline(P.DetCfg(limit_side_len=hw), rec_batch_num=16, rec_img_h=48, rec_img_w=320, enable_cls=True, det_net="srv_det", rec_net="srv_rec")
    want = [ora.process(im, prob_override=pr) for im, pr in zip(imgs, probs)]
    srv = os.path.join(ROOT, "models_server")
    got = {}
    for prec in ("fp32", "fp16"):
        pipe = pkg.Pipe(device=0, enable_cls=True, limit_side_len=hw, rec_batch_num=16, rec_img_h=48, rec_img_w=320, precision=prec,
                        det_dir=os.path.join(srv, "det"), rec_dir=os.path.join(srv, "rec"))
        d_imgs, d_probs = pkg.DevArray(np.stack(imgs)), pkg.DevArray(np.stack(probs))
        got[prec] = pipe.run_device(d_imgs, hw, hw, n, d_probs, collect=True)
        pipe.close()
    lines = same16 = 0
    for i in range(n):
        w = want[i]["words"]
        assert len(w) > 0
        for prec in ("fp32", "fp16"):
            g = got[prec][i]
            assert len(g) == len(w), (prec, i, len(g), len(w))
            for a, b in zip(g, w):
                assert np.array_equal(a["box"], np.array(b["box"]).reshape(4, 2)), (prec, i)
        for a, b in zip(got["fp32"][i], w):
            assert np.array_equal(a["ids"], b["ids"]), i
            assert abs(a["confidence"] - b["confidence"]) <= 1e-5, (a["confidence"], b["confidence"])
        for a, b in zip(got["fp16"][i], w):
            lines += 1
            same16 += np.array_equal(a["ids"], b["ids"])
    assert same16 >= 0.9 * lines, (same16, lines)


def test_pool_under_a_mixed_size_stream_on_eight_logical_devices(pkg, built):
    """the C++ GPUWorkerPool (worker i -> device i mod n, /root/reference/src/gpu_worker_pool.cpp:12-16,46-59)
    under a configs[3]-shaped closed-loop stream of mixed-size requests on EIGHT logical devices of one process
    (OCR_DEVICE_MAP=0,0,0,0,0,0,0,0 on a one-GPU lease: own attribute memos, arenas, streams and chain threads per logical
    id), in both dispatch modes: every reply's words equal the one-worker pool's reply for the same image, every worker
    served requests, and the same image gets the same words whoever serves it (host/pool_load.cpp)."""
    import json
    import subprocess
    host = os.path.join(ROOT, "cpp-paddle-ocr_amd", "host")
    subprocess.check_call(["make", "-s", "-C", host])
    env = dict(os.environ, OCR_DEVICE_MAP="0,0,0,0,0,0,0,0", OCR_WORKER_DET_LIMIT="960")
    for extra in ([], ["least"]):
        out = subprocess.run([os.path.join(host, "pool_load"), os.path.join(ROOT, "models"), "8", "120", "24", "16"] + extra,
                             capture_output=True, text=True, timeout=600, env=env)
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
        js = json.loads(out.stdout.strip().splitlines()[-1])
        assert js["replies_differing_from_one_worker"] == 0
        assert sum(js["requests_per_worker"]) == 120 and min(js["requests_per_worker"]) > 0, js["requests_per_worker"]


_MLP_CHILD = r"""
import sys, numpy as np
sys.path[:0] = [sys.argv[1], sys.argv[1] + "/tools"]
import __graft_entry__ as ge
pkg = ge.load_package()
x = np.random.RandomState(21).randn(3, 48, 320, 3).astype(np.float32)
net = pkg.SrvNet("rec", "fp16")
plain = net.forward(x, keep_all=True)
net.timing(True)
fused = net.forward(x, keep_all=False)
names = list(net.timing_report())
net.close()
d = np.abs(plain - fused)
print("MLP", sum(".mlp_" in n for n in names), sum(".mlp_ln_" in n for n in names), sum(".ln_" in n for n in names), sum("_768_192_" in n or "_1024_256_" in n for n in names),
      int(np.array_equal(plain.view(np.uint32), fused.view(np.uint32))), float(d.max()), float(np.abs(plain).max()), float((plain.argmax(-1) == fused.argmax(-1)).mean()))
"""


def _mlp_child(env):
    import subprocess
    import sys
    pr = subprocess.run([sys.executable, "-c", _MLP_CHILD, ROOT], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert pr.returncode == 0 and "MLP " in pr.stdout, (pr.stdout[-1500:], pr.stderr[-1500:])
    f = pr.stdout[pr.stdout.index("MLP ") + 4:].split()
    return dict(mlp=int(f[0]), mlp_ln=int(f[1]), ln=int(f[2]), unfused=int(f[3]), same_bits=int(f[4]), dmax=float(f[5]), scale=float(f[6]), argmax=float(f[7]))


def test_fused_mlp_equals_the_two_launches(pkg, built):
    """SVTR's MLP as one kernel (csrc/srv_mlp.h: the 4 C wide hidden tensor stays in LDS) against the two GEMM launches it
    replaces: production mode fuses, keep_all mode materialises every tensor - the recognizer's logits are bit-identical
    (hidden values rounded to f16 once either way, fc2 accumulating them in ascending k), and the timing report shows the
    fused launches.  (The default: the LayerNorm in front of the MLP is a launch of its own; absorbed - OCR_SRV_MLPLN=1 - the arithmetic
    is another one: the next test.)"""
    _srv_ready()
    r = _mlp_child({})
    assert r["mlp"] == 12 and r["mlp_ln"] == 0 and r["unfused"] == 0, r  # the 3 + 9 blocks of width 192 and 256 (512: two launches)
    assert r["same_bits"] == 1, r


def test_layernorm_absorbed_into_the_fused_mlp(pkg, built):
    """OCR_SRV_MLPLN=1 (opt-in): the LayerNorm in front of a fused MLP absorbed into it: LN(u) W1 + b1 = r (u W1') - r m s + c with the folded image W1' = diag(gamma) W1,
    the token's mean and rstd found in the kernel from the tiles it streams anyway, the residual normalised on the fly - the
    normalised tensor is never written, twelve LayerNorm launches are gone.  Another (equally valid) f16 arithmetic than LayerNorm-then-
    MLP: against the keep_all run (every op a launch) the logits agree to 3 % of their scale, the arg max on >= 95 % of the steps."""
    _srv_ready()
    base, absorbed = _mlp_child({}), _mlp_child({"OCR_SRV_MLPLN": "1"})
    assert absorbed["mlp_ln"] == 12 and absorbed["mlp"] == 12 and absorbed["ln"] == base["ln"] - 12, (base, absorbed)  # (".mlp_" also counts ".mlp_ln_")
    print("absorbed LayerNorm: max |d| %.3g of |logit| <= %.3g, arg max equal on %.4f" % (absorbed["dmax"], absorbed["scale"], absorbed["argmax"]))
    assert absorbed["dmax"] <= 0.03 * absorbed["scale"] and absorbed["argmax"] >= 0.95, absorbed


_CFG_CHILD = r"""
import sys, hashlib, numpy as np
sys.path[:0] = [sys.argv[1], sys.argv[1] + "/tools"]
import __graft_entry__ as ge
pkg = ge.load_package()
out = []
for kind, shape, seed in (("rec", (3, 48, 320), 31), ("det", (1, 96, 160), 32)):
    x = np.random.RandomState(seed).randn(shape[0], shape[1], shape[2], 3).astype(np.float32)
    net = pkg.SrvNet(kind, "fp16")
    y = net.forward(x, keep_all=False)
    net.timing(True)
    net.forward(x, keep_all=False)
    names = " ".join(net.timing_report())
    out.append(hashlib.sha1(np.ascontiguousarray(y).view(np.uint8)).hexdigest() + ":" + str(names.count("[" + sys.argv[2] + "/") + names.count("[" + sys.argv[2] + "]")))
    net.close()
print("CFG", *out)
"""


def test_every_tile_configuration_gives_the_same_bits(pkg, built):
    """The f16 GEMM family's tile configurations (csrc/srv_kernels.hip: SRV_CFGS, the 256 x 256 tiles, the two-column-block small
    tiles) accumulate every output in the same order and share one epilogue (residual in, output out as whole lines through a
    wave-private LDS block): forced one at a time (OCR_SRV_CFG, read once per
    process - child processes, four at a time), the recognizer's logits and the detector's map are bit-identical to the tuned
    run's, and the timing report shows that the forced configuration really ran."""
    import subprocess
    import sys
    _srv_ready()
    cfgs = [("", "", {}), ("0", "256x128/2x4", {}), ("4", "128x256/1x8", {}), ("6", "256x128/4x2", {}), ("9", "64x64/2x2", {}),
            ("12", "256x256/2x4", {}), ("13", "256x256/4x2", {}), ("14", "128x64/4x1", {}), ("15", "128x128/4x2", {}), ("16", "256x64/4x1", {}),
            ("3", "128x64/2x2", {}), ("1", "128x128/2x2", {}), ("20", "halo16x16x64", {}), ("21", "128x192/2x2", {}), ("22", "256x192/4x2", {})]
    pending, running, res = list(enumerate(cfgs)), [], {}
    while pending or running:
        while pending and len(running) < 4:
            i, (cfg, name, extra) = pending.pop(0)
            env = dict(os.environ, **extra)
            if cfg:
                env.update(OCR_SRV_CFG=cfg)
            running.append((i, subprocess.Popen([sys.executable, "-c", _CFG_CHILD, ROOT, name or "-"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
        i, pr = running.pop(0)
        so, se = pr.communicate(timeout=600)
        assert pr.returncode == 0 and "CFG " in so, (cfgs[i], so[-1500:], se[-1500:])
        res[i] = so[so.index("CFG ") + 4:].split()
    want = [r.split(":")[0] for r in res[0]]
    for i, (cfg, name, extra) in enumerate(cfgs):
        got = [r.split(":")[0] for r in res[i]]
        assert got == want, (cfg, name, extra)
        if cfg:
            assert any(int(r.split(":")[1]) > 0 for r in res[i]), (cfg, name, res[i])  # launches with the forced tile in their names (the halo form: detector only)


def test_fused_head_tail_matches_the_two_launches(pkg, built):
    """The DB head's two transposed convs as one kernel (csrc/srv_kernels.hip head_tail_kernel: the 64-channel map between them
    stays in registers as the second product's operand) against the two launches: production mode fuses, keep_all materialises.
    The mid values are rounded to f16 either way; the second product runs on the matrix pipe with f16 weights instead of f32 fmas,
    so the probability maps agree to 3e-3 (measured 1.1e-3), not bit for bit.  An image whose size is not a multiple of the tile."""
    _srv_ready()
    x = np.random.RandomState(41).randn(2, 96, 160, 3).astype(np.float32)
    net = pkg.SrvNet("det", "fp16")
    plain = net.forward(x, keep_all=True)
    net.timing(True)
    fused = net.forward(x, keep_all=False)
    names = list(net.timing_report())
    assert sum(".head_tail_" in n for n in names) == 1 and not any("deconv" in n for n in names), names[-4:]
    assert plain.shape == fused.shape == (2, 96, 160, 1)
    d = np.abs(plain - fused)
    print("fused head tail: max |d| %.2e" % d.max())
    assert d.max() <= 3e-3, float(d.max())
    net.close()


def test_server_nets_at_production_shapes_against_the_oracle(pkg, built):
    """What bench.py --config cfg5 launches - the tuned tile configurations of 960 x 960 images and of a full batch of 48 x 320 lines
    (256 x 256 tiles, the halo form, the fused MLP / head tail: choices the small parity shapes never see) - against the oracle's f32
    execution of the plans: one image through the detector in production mode (fp16: probability map within 2e-2 on >= 99 % of the
    pixels, the binarised map's area within 1 %; f32 build: every pixel equal), 32 lines through the recognizer (fp16: arg max
    equal on >= 90 % of the steps, logits within 6 % of their scale; f32 build: every logit equal)."""
    from oracle import OracleNet
    _srv_ready()
    rs = np.random.RandomState(61)
    x = rs.randn(1, 960, 960, 3).astype(np.float32)
    want = OracleNet("srv_det").run(x)
    twin = pkg.SrvNet("det", "fp32")
    assert np.array_equal(twin.forward(x, keep_all=False), want)
    twin.close()
    net = pkg.SrvNet("det", "fp16")
    got = net.forward(x, keep_all=False)
    net.close()
    assert (np.abs(got - want) <= 2e-2).mean() >= 0.99, float((np.abs(got - want) <= 2e-2).mean())
    assert abs(float((got > 0.3).sum()) - float((want > 0.3).sum())) <= 0.01 * max(1.0, float((want > 0.3).sum()))
    x = rs.randn(32, 48, 320, 3).astype(np.float32)
    ora = OracleNet("srv_rec")
    want = np.concatenate([ora.run(x[i:i + 8]) for i in range(0, 32, 8)])
    twin = pkg.SrvNet("rec", "fp32")
    assert np.array_equal(twin.forward(x, keep_all=False), want)
    twin.close()
    net = pkg.SrvNet("rec", "fp16")
    got = net.forward(x, keep_all=False)
    net.close()
    assert np.abs(got - want).max() <= 0.06 * np.abs(want).max(), float(np.abs(got - want).max())
    assert (got.argmax(-1) == want.argmax(-1)).mean() >= 0.90, float((got.argmax(-1) == want.argmax(-1)).mean())


def test_server_fp16_results_do_not_depend_on_the_batch(pkg, built):
    """A size-independent property of the f16 build at production shapes: a text line's logits and an image's probability map are the same
    BITS whether the line / image runs alone, in a small batch or in a full one - every matrix product accumulates a row in one fixed
    order whatever tile configuration the shape's tuning picked (test_every_tile_configuration_gives_the_same_bits), attention, LayerNorm
    and the fused kernels work per line / per pixel.  (The f32 build has this by equalling the oracle.)"""
    _srv_ready()
    rs = np.random.RandomState(71)
    x = rs.randn(96, 48, 320, 3).astype(np.float32)
    net = pkg.SrvNet("rec", "fp16")
    full = net.forward(x, keep_all=False)
    some = net.forward(x[40:48], keep_all=False)
    one = net.forward(x[95:96], keep_all=False)
    net.close()
    assert np.array_equal(full[40:48].view(np.uint32), some.view(np.uint32))
    assert np.array_equal(full[95:96].view(np.uint32), one.view(np.uint32))
    y = rs.randn(3, 480, 640, 3).astype(np.float32)
    net = pkg.SrvNet("det", "fp16")
    full = net.forward(y, keep_all=False)
    one = net.forward(y[2:3], keep_all=False)
    net.close()
    assert np.array_equal(full[2:3].view(np.uint32), one.view(np.uint32))


_CTC_CHILD = r"""
import sys, os, json, numpy as np
sys.path[:0] = [sys.argv[1], sys.argv[1] + "/tools"]
import __graft_entry__ as ge
pkg = ge.load_package()
from synth_data import cfg2_sample
n, hw, k = 3, 320, 8
samples = [cfg2_sample(200 + i, hw, hw, k) for i in range(n)]
srv = os.path.join(sys.argv[1], "models_server")
pipe = pkg.Pipe(device=0, enable_cls=True, limit_side_len=hw, rec_batch_num=16, rec_img_h=48, rec_img_w=320, precision="fp16",
                det_dir=os.path.join(srv, "det"), rec_dir=os.path.join(srv, "rec"))
pipe.timing(True)
got = pipe.run_device(pkg.DevArray(np.stack([s[0] for s in samples])), hw, hw, n, pkg.DevArray(np.stack([s[1] for s in samples])), collect=True)
names = " ".join(pipe.timing_report())
pipe.close()
print("CTC", json.dumps({"ctc_launches": names.count("_ctc["), "words": [[(w["ids"].tolist(), float(w["confidence"])) for w in im] for im in got]}))
"""


def test_ctc_head_in_partial_mode_equals_logits_then_arg_max(pkg, built):
    """The server recognizer's CTC head inside the pipeline (f16 build): every workgroup of the last linear leaves a per-row partial
    (max, sum of exp, first arg max) of its column tile and a small kernel folds them - the 2.2 GB of logits per 1024 lines are never
    written - against OCR_SRV_CTC=0 (logits, then one pass over them): the same id sequences, confidences to 2e-6 (sums in another order)."""
    import json
    import subprocess
    import sys
    _srv_ready()
    res = {}
    procs = {k: subprocess.Popen([sys.executable, "-c", _CTC_CHILD, ROOT], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for k, env in (("partials", {}), ("logits", {"OCR_SRV_CTC": "0"}))}
    for k, pr in procs.items():
        so, se = pr.communicate(timeout=600)
        assert pr.returncode == 0 and "CTC " in so, (k, so[-1500:], se[-1500:])
        res[k] = json.loads(so[so.index("CTC ") + 4:])
    assert res["partials"]["ctc_launches"] >= 1 and res["logits"]["ctc_launches"] == 0
    a, b = res["partials"]["words"], res["logits"]["words"]
    assert len(a) == len(b) and sum(len(im) for im in a) >= 12
    for ia, ib in zip(a, b):
        assert len(ia) == len(ib)
        for (ids_a, c_a), (ids_b, c_b) in zip(ia, ib):
            assert ids_a == ids_b
            assert abs(c_a - c_b) <= 2e-6, (c_a, c_b)


def test_concat_folded_into_the_head_conv_gives_the_same_bits(pkg, built):
    """The DB neck's `concat up=8,4,2,1` folded into the 3 x 3 conv that reads it (halo form: channel tile j of the conv's K order is
    source j, its patch comes from source j at (y >> sh, x >> sh)) against the materialised concatenation: production mode folds,
    OCR_SRV_CAT=0 (a child process) launches the concat.  Same values in the same accumulation order: the probability maps are
    bit-identical; the timing report shows one `_cat` launch and no concat.  An image that is not a multiple of the 16 x 16 tile."""
    import hashlib
    import subprocess
    import sys
    _srv_ready()
    child = r"""
import sys, hashlib, numpy as np
sys.path[:0] = [sys.argv[1], sys.argv[1] + "/tools"]
import __graft_entry__ as ge
pkg = ge.load_package()
x = np.random.RandomState(81).randn(2, 160, 224, 3).astype(np.float32)
net = pkg.SrvNet("det", "fp16")
y = net.forward(x, keep_all=False)
net.timing(True)
net.forward(x, keep_all=False)
names = " ".join(net.timing_report())
net.close()
print("CAT", hashlib.sha1(np.ascontiguousarray(y).view(np.uint8)).hexdigest(), names.count("_cat["), names.count(".concat_"))
"""
    out = {}
    for k, env in (("fold", {}), ("concat", {"OCR_SRV_CAT": "0", "OCR_SRV_CFG": "20"})):
        pr = subprocess.run([sys.executable, "-c", child, ROOT], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert pr.returncode == 0 and "CAT " in pr.stdout, (k, pr.stdout[-1500:], pr.stderr[-1500:])
        out[k] = pr.stdout[pr.stdout.index("CAT ") + 4:].split()
    assert out["fold"][1:] == ["1", "0"] and out["concat"][1:] == ["0", "1"], out
    assert out["fold"][0] == out["concat"][0], out
