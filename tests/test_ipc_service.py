"""The Linux IPC service (host/ocr_ipc_service.h, ocr_service): the reference's JSON protocol
(/root/reference/src/ocr_ipc_service.cpp:310-448) over a Unix-domain socket with 4-byte length framing.
CPU part: everything that does not need a worker (status, errors, limits, shutdown).  GPU part: recognize
with a file path and with base64 PNG, compared with the pipeline."""
import base64
import io
import json
import os
import socket
import struct
import subprocess
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "cpp-paddle-ocr_amd", "host")


class Client:
    def __init__(self, path):
        self.s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        for _ in range(200):
            try:
                self.s.connect(path)
                return
            except OSError:
                time.sleep(0.05)
        raise RuntimeError("service did not come up")

    def call_raw(self, payload: bytes):
        self.s.sendall(struct.pack("<I", len(payload)) + payload)
        n = struct.unpack("<I", self._read(4))[0]
        return json.loads(self._read(n).decode("utf-8"))

    def call(self, obj):
        return self.call_raw(json.dumps(obj).encode("utf-8"))

    def _read(self, n):
        b = b""
        while len(b) < n:
            c = self.s.recv(n - len(b))
            assert c, "service closed the connection"
            b += c
        return b


def _start(tmp_path, gpu_workers):
    subprocess.check_call(["make", "-s", "-C", HOST])
    sock = str(tmp_path / "ocr.sock")
    proc = subprocess.Popen([os.path.join(HOST, "ocr_service"), "--model-dir", os.path.join(ROOT, "models"), "--pipe-name", sock,
                             "--gpu-workers", str(gpu_workers)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    return proc, sock


def _png_bytes(bgr):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(bgr[:, :, ::-1]).save(buf, format="PNG")
    return buf.getvalue()


def _jpeg_bytes(rgb, **kw):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(rgb).save(buf, format="JPEG", **kw)
    return buf.getvalue()


def _jpeg_cases(card):
    from PIL import Image
    rs = np.random.RandomState(0)
    yy, xx = np.mgrid[0:200, 0:300]
    images = {"card": card[:, :, ::-1].copy(), "noise": rs.randint(0, 256, (53, 37, 3)).astype(np.uint8),
              "ramp": np.stack([xx * 255 // 300, yy * 255 // 200, (xx + yy) * 255 // 500], -1).astype(np.uint8),
              "one": rs.randint(0, 256, (1, 1, 3)).astype(np.uint8), "thin": rs.randint(0, 256, (3, 130, 3)).astype(np.uint8)}
    cases = []
    for name, arr in images.items():
        for ss in (0, 1, 2):
            cases.append((name, arr, dict(quality=90, subsampling=ss)))
            cases.append((name, arr, dict(quality=80, subsampling=ss, progressive=True)))
        cases.append((name, arr, dict(quality=25)))
    cases.append(("gray", np.array(Image.fromarray(images["card"]).convert("L")), dict(quality=85)))
    cases.append(("restart", images["card"], dict(quality=85, restart_marker_blocks=3)))
    cases.append(("optimize", images["card"], dict(quality=85, optimize=True)))
    cases.append(("gray progressive", np.array(Image.fromarray(images["card"]).convert("L")), dict(quality=85, progressive=True)))
    cases.append(("restart progressive", images["card"], dict(quality=85, restart_marker_blocks=3, progressive=True)))
    cases.append(("progressive q98", images["card"], dict(quality=98, progressive=True)))
    return images, cases


def test_jpeg_decoder_matches_libjpeg(built, card, tmp_path):
    """host/jpeg_decode.h restates libjpeg's default pipeline (Huffman sequential + progressive, islow IDCT,
    fancy upsampling, fixed-point YCC->RGB): bit-exact with libjpeg-turbo (through PIL) on every variant."""
    from PIL import Image
    subprocess.check_call(["make", "-s", "-C", HOST])
    tool = os.path.join(HOST, "decode_tool")
    images, cases = _jpeg_cases(card)
    for name, arr, kw in cases:
        src, dst = tmp_path / "t.jpg", tmp_path / "t.ppm"
        src.write_bytes(_jpeg_bytes(arr, **kw))
        want = np.array(Image.open(src).convert("RGB"))
        assert subprocess.run([tool, str(src), str(dst)]).returncode == 0, (name, kw)
        got = np.array(Image.open(dst))
        assert np.array_equal(got, want), (name, kw)
    # garbage and truncated headers are refused, not crashed on
    for blob in (b"\xff\xd8\xff\xd9", b"\xff\xd8" + bytes(100), _jpeg_bytes(images["card"], quality=85)[:150]):
        src = tmp_path / "bad.jpg"
        src.write_bytes(blob)
        assert subprocess.run([tool, str(src), str(tmp_path / "bad.ppm")], capture_output=True).returncode == 1
    # the other containers of the service decode to the same pixels
    for fmt in ("PNG", "BMP", "PPM"):
        src = tmp_path / ("c." + fmt.lower())
        Image.fromarray(images["card"]).save(src, format=fmt)
        assert subprocess.run([tool, str(src), str(tmp_path / "c.out.ppm")]).returncode == 0
        assert np.array_equal(np.array(Image.open(tmp_path / "c.out.ppm")), images["card"])


@pytest.mark.gpu
def test_device_jpeg_decode_equals_host(built, card, tmp_path):
    """the pixel half of JPEG decoding on the device (csrc/kernels_jpeg.hip through ocr_jpeg_decode: the
    host only entropy-decodes).  Same variants as the host test - sequential / progressive, 4:4:4 / 4:2:2 / 4:2:0, grey,
    restart intervals, 1x1 and 3-row images, quality 25..98 - each bit for bit equal to libjpeg-turbo (PIL), hence
    to host/jpeg_decode.h."""
    from PIL import Image
    subprocess.check_call(["make", "-s", "-C", HOST])
    tool = os.path.join(HOST, "decode_tool")
    images, cases = _jpeg_cases(card)
    big = np.random.RandomState(5).randint(0, 256, (333, 517, 3)).astype(np.uint8)
    cases += [("big", big, dict(quality=75, subsampling=ss)) for ss in (0, 1, 2)]
    for name, arr, kw in cases:
        src, dst = tmp_path / "t.jpg", tmp_path / "t.ppm"
        src.write_bytes(_jpeg_bytes(arr, **kw))
        want = np.array(Image.open(src).convert("RGB"))
        r = subprocess.run([tool, "--device", str(src), str(dst)], capture_output=True, text=True)
        assert r.returncode == 0, (name, kw, r.stderr)
        assert np.array_equal(np.array(Image.open(dst)), want), (name, kw)


def test_ipc_protocol_without_workers(built, card, tmp_path):
    proc, sock = _start(tmp_path, 0)
    try:
        c = Client(sock)
        st = c.call({"command": "status"})
        assert st["success"] is True
        info = json.loads(st["status"])                      # a JSON document inside a string, like the reference
        assert info == {"running": True, "total_requests": 0, "successful_requests": 0, "average_processing_time_ms": 0.0}
        assert c.call({"command": "bogus"}) == {"success": False, "error": "Unknown command: bogus"}
        assert c.call({"nothing": 1}) == {"success": False, "error": "Unknown command: "}
        bad = c.call_raw(b'{"command": "status"')
        assert bad["success"] is False and bad["error"].startswith("Invalid JSON: ")
        assert c.call({"command": "recognize"}) == {"success": False, "error": "Missing image_path or image_data"}
        assert c.call({"command": "recognize", "image_path": "/nonexistent.png"}) == \
            {"success": False, "error": "Failed to load image from path: /nonexistent.png"}
        assert c.call({"command": "recognize", "image_data": "@@@@"}) == \
            {"success": False, "error": "Base64 decode error: invalid character"}
        assert c.call({"command": "recognize", "image_data": base64.b64encode(b"not an image").decode()}) == \
            {"success": False, "error": "Failed to decode base64 image data"}
        # a decodable image (PNG through libpng, baseline JPEG, PPM, BMP) reaches the worker stage: none configured here
        ppm = b"P6\n# c\n%d %d\n255\n" % (card.shape[1], card.shape[0]) + card[:, :, ::-1].tobytes()
        from PIL import Image
        bmp = io.BytesIO()
        Image.fromarray(card[:, :, ::-1]).save(bmp, format="BMP")
        for blob in (_png_bytes(card), _jpeg_bytes(card[:, :, ::-1].copy(), quality=90), ppm, bmp.getvalue()):
            r = c.call({"command": "recognize", "image_data": base64.b64encode(blob).decode()})
            assert r == {"success": False, "error": "No GPU workers configured (this build has no CPU path)"}
        # a message that fills the 1 MiB read buffer is refused, the connection stays usable
        r = c.call_raw(b" " * (1048576 - 1))
        assert r == {"success": False, "error": "Data too large for buffer (max 1MB). Consider using file path transmission."}
        assert c.call({"command": "status"})["success"] is True
        # two clients at once
        c2 = Client(sock)
        assert c2.call({"command": "status"})["success"] is True
        sd = c.call({"command": "shutdown"})
        assert sd == {"success": True, "message": "Shutdown command received, stopping service..."}
        assert proc.wait(timeout=20) == 0
    finally:
        if proc.poll() is None:
            proc.kill()


@pytest.mark.gpu
def test_ipc_recognize_matches_pipeline(pkg, built, card, tmp_path):
    proc, sock = _start(tmp_path, 1)
    try:
        c = Client(sock)
        path = tmp_path / "card.png"
        path.write_bytes(_png_bytes(card))
        pipe = pkg.Pipe()
        want = pipe.run([card])[0]
        labels = [pipe.label(i) for i in range(6625)]
        for req in ({"command": "recognize", "image_path": str(path)},
                    {"command": "recognize", "image_data": base64.b64encode(_png_bytes(card)).decode()}):
            r = c.call(req)
            assert r["success"] is True and r["width"] == card.shape[1] and r["height"] == card.shape[0]
            assert len(r["words"]) == len(want) > 0
            for g, w in zip(r["words"], want):
                assert np.array_equal(np.array(g["box"]), w["box"])
                assert g["text"] == "".join(labels[i] for i in w["ids"])
                assert np.float32(g["confidence"]) == np.float32(w["confidence"])
        # a JPEG request: the service's decoder, then the same pipeline as on the decoded pixels
        from PIL import Image
        jb = _jpeg_bytes(card[:, :, ::-1].copy(), quality=92)
        dec = np.array(Image.open(io.BytesIO(jb)).convert("RGB"))[:, :, ::-1].copy()
        wj = pipe.run([dec])[0]
        r = c.call({"command": "recognize", "image_data": base64.b64encode(jb).decode()})
        assert r["success"] is True and len(r["words"]) == len(wj) > 0
        for g, w in zip(r["words"], wj):
            assert np.array_equal(np.array(g["box"]), w["box"]) and g["text"] == "".join(labels[i] for i in w["ids"])
        info = json.loads(c.call({"command": "status"})["status"])
        assert info["total_requests"] == 3 and info["successful_requests"] == 3 and info["average_processing_time_ms"] > 0
        c.call({"command": "shutdown"})
        assert proc.wait(timeout=30) == 0
        pipe.close()
    finally:
        if proc.poll() is None:
            proc.kill()


@pytest.mark.gpu
def test_ipc_concurrent_requests_are_batched_with_identical_results(pkg, built, card, tmp_path):
    """Concurrent clients: the worker takes the queued requests as one pipeline run (OCRWorker::processBatch).
    Every reply must be what the request gets alone - same boxes, text and confidence bits."""
    import threading
    proc, sock = _start(tmp_path, 1)
    try:
        imgs = [card, np.ascontiguousarray(card[:, ::-1]), np.ascontiguousarray(card[::2, ::2]),
                np.ascontiguousarray(card[100:700, 50:900])]
        paths = []
        for i, im in enumerate(imgs):
            p = tmp_path / f"img{i}.png"
            p.write_bytes(_png_bytes(im))
            paths.append(str(p))
        # JPEG requests too: a batch of JPEGs only is decoded on the device (OCRWorker::processBatch -> ocr_pipe_stage_jpeg),
        # a batch that mixes them with PNGs finishes its JPEGs on the host - the replies must not depend on which
        for i, (im, ss) in enumerate(zip(imgs, (0, 1, 2, 2))):
            p = tmp_path / f"img{i}.jpg"
            p.write_bytes(_jpeg_bytes(np.ascontiguousarray(im[:, :, ::-1]), quality=90, subsampling=ss))
            paths.append(str(p))
        c0 = Client(sock)
        alone = [c0.call({"command": "recognize", "image_path": p}) for p in paths]
        assert all(a["success"] for a in alone) and len(alone[0]["words"]) > 0
        nthreads, rounds = 8, 3
        out = [[None] * rounds for _ in range(nthreads)]

        def work(t):
            c = Client(sock)
            for r in range(rounds):
                k = (t + r) % len(paths)
                out[t][r] = (k, c.call({"command": "recognize", "image_path": paths[k]}))

        th = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
        [t.start() for t in th]
        [t.join() for t in th]
        # one more round with JPEGs only (the all-JPEG batch path)
        jp = [k for k, p in enumerate(paths) if p.endswith(".jpg")]
        extra = [[None] for _ in range(nthreads)]

        def work_jpeg(t):
            c = Client(sock)
            k = jp[t % len(jp)]
            extra[t][0] = (k, c.call({"command": "recognize", "image_path": paths[k]}))

        th = [threading.Thread(target=work_jpeg, args=(t,)) for t in range(nthreads)]
        [t.start() for t in th]
        [t.join() for t in th]
        out = [out[t] + extra[t] for t in range(nthreads)]
        rounds += 1
        for t in range(nthreads):
            for r in range(rounds):
                k, got = out[t][r]
                want = alone[k]
                assert got["success"] is True and got["width"] == want["width"] and got["height"] == want["height"]
                assert len(got["words"]) == len(want["words"])
                for g, w in zip(got["words"], want["words"]):
                    assert g["box"] == w["box"] and g["text"] == w["text"] and g["confidence"] == w["confidence"]
        info = json.loads(c0.call({"command": "status"})["status"])
        assert info["total_requests"] == len(paths) + nthreads * rounds == info["successful_requests"]
        c0.call({"command": "shutdown"})
        assert proc.wait(timeout=30) == 0
    finally:
        if proc.poll() is None:
            proc.kill()
