"""Throughput of the IPC service under concurrent clients (GPU box): starts host/ocr_service with one GPU
worker, then C client threads send R `recognize` requests each for a 960x960 card image (by path, PNG).
Prints requests/s for OCR_WORKER_MAX_BATCH=1 (the reference's one-at-a-time worker) and for the default
dynamic batching.  usage: python tools/service_load.py [clients=32] [requests_per_client=8]"""
import json, os, socket, struct, subprocess, sys, threading, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
HOST = os.path.join(ROOT, "cpp-paddle-ocr_amd", "host")


def call(s, obj):
    b = json.dumps(obj).encode()
    s.sendall(struct.pack("<I", len(b)) + b)
    def rd(n):
        d = b""
        while len(d) < n:
            c = s.recv(n - len(d))
            assert c
            d += c
        return d
    n = struct.unpack("<I", rd(4))[0]
    return json.loads(rd(n).decode())


def connect(path):
    for _ in range(400):
        try:
            s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            s.connect(path)
            return s
        except OSError:
            time.sleep(0.05)
    raise RuntimeError("service did not come up")


RESULTS = []


def run(max_batch, clients, per_client, img_path, fmt, workers=1, extra_env=None, tag=None):
    sock = f"/tmp/ocr_load_{os.getpid()}_{max_batch}_{abs(hash(str(extra_env))) % 9999}.sock"
    env = dict(os.environ, OCR_WORKER_MAX_BATCH=str(max_batch))
    env.update(extra_env or {})
    proc = subprocess.Popen([os.path.join(HOST, "ocr_service"), "--model-dir", os.path.join(ROOT, "models"), "--pipe-name", sock,
                             "--gpu-workers", str(workers)], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT)
    import psutil
    ps = psutil.Process(proc.pid)
    try:
        s0 = connect(sock)
        for _ in range(3):
            r = call(s0, {"command": "recognize", "image_path": img_path})
            assert r["success"], r
        lat = []
        lock = threading.Lock()
        def work():
            s = connect(sock)
            for _ in range(per_client):
                t0 = time.perf_counter()
                r = call(s, {"command": "recognize", "image_path": img_path})
                assert r["success"]
                with lock:
                    lat.append((time.perf_counter() - t0) * 1e3)
        th = [threading.Thread(target=work) for _ in range(clients)]
        c0 = ps.cpu_times()
        t0 = time.perf_counter()
        [t.start() for t in th]
        [t.join() for t in th]
        dt = time.perf_counter() - t0
        c1 = ps.cpu_times()
        lat.sort()
        rec = {"max_batch": max_batch, "gpu_workers": workers, "clients": clients, "requests": len(lat), "image": fmt,
               "requests_per_s": round(len(lat) / dt, 1), "p50_ms": round(lat[len(lat) // 2], 1),
               "p99_ms": round(lat[int(len(lat) * 0.99) - 1], 1), "words_per_image": len(r["words"]),
               "service_host_cores_busy": round(((c1.user + c1.system) - (c0.user + c0.system)) / dt, 2)}
        if tag:
            rec["mode"] = tag
        rec["host_cores_per_1000_requests_per_s"] = round(rec["service_host_cores_busy"] / max(1e-9, rec["requests_per_s"]) * 1000, 2)
        RESULTS.append(rec)
        print(json.dumps(rec), flush=True)
        call(s0, {"command": "shutdown"})
        proc.wait(timeout=30)
    finally:
        if proc.poll() is None:
            proc.kill()


def jpeg_mode(clients, per):
    """960x960 JPEG requests (a configs[1] card, quality 90, 4:2:0) against a service with two GPU workers on the one
    device: OCR_DEVICE_JPEG=1 (entropy decoding on the client's service thread, dequantisation / IDCT / upsampling /
    colour on the GPU, straight into the staging slot) against =0 (the whole decode on the host), with the service
    process's busy host cores beside it.  One JSON line per mode."""
    from PIL import Image
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from synth_data import cfg2_sample
    img = cfg2_sample(0, 960, 960, 32)[0]
    path = f"/tmp/ocr_load_{os.getpid()}.jpg"
    Image.fromarray(img[:, :, ::-1]).save(path, quality=90, subsampling=2)
    fmt = "jpeg 960x960 q90 4:2:0 (%d KB)" % (os.path.getsize(path) // 1024)
    op = {"OCR_WORKER_DET_LIMIT": "960", "OCR_WORKER_REC_H": "48", "OCR_WORKER_REC_W": "320", "OCR_WORKER_CLS": "1"}   # configs[1]'s operating point
    for dev in ("1", "0"):
        run(16, clients, per, path, fmt, workers=2, extra_env=dict(op, OCR_DEVICE_JPEG=dev),
            tag="device_jpeg" if dev == "1" else "host_jpeg")
    # the same service with the stages' precision parameter set to "fp16" (OCR_WORKER_PRECISION: the reference's worker
    # hard-codes "fp32") - where the GPU, not the host, was the limit this is what the mode buys a service
    run(16, clients, per, path, fmt, workers=2, extra_env=dict(op, OCR_DEVICE_JPEG="1", OCR_WORKER_PRECISION="fp16"), tag="device_jpeg_fp16")
    # the yardstick: the pipeline itself (one handle, two chains) on the SAME decoded image, 64 per batch from host memory
    # through the double-buffered staging - what bench.py calls host_input, but with this image's own detector output
    # (no probability-map protocol: the service cannot be handed one), i.e. the same words per image as the service saw
    import threading as th_
    from __graft_entry__ import load_package
    pkg = load_package()
    dec = np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])
    pipe = pkg.Pipe(enable_cls=True, limit_side_len=960, rec_batch_num=16, rec_img_h=48, rec_img_w=320)
    batch = [dec] * 64
    pipe.stage(0, batch)
    pipe.stage(1, batch)
    w = pipe.run_staged(0)
    pipe.run_staged(1, collect=False)
    steps = 8
    pipe.stage(0, batch)
    t0 = time.perf_counter()
    for k in range(steps):
        t = th_.Thread(target=pipe.stage, args=(1 - (k & 1), batch))
        t.start()
        pipe.run_staged(k & 1, collect=False)
        t.join()
    dt = time.perf_counter() - t0
    print(json.dumps({"mode": "pipeline_host_input_same_image", "images_per_s": round(64 * steps / dt, 1), "words_per_image": len(w[0]),
                      "what": "ocr_pipe_stage + ocr_pipe_run_staged, 64 copies of the decoded image per batch, double-buffered"}), flush=True)
    pipe.close()


def sweep_mode(per):
    """Latency against offered load: the device-JPEG service of jpeg_mode with 8 .. 96 closed-loop
    clients, for the shipped batching (whatever is queued, up to 16: OCRWorker's default since round 4) and for other caps
    (OCR_WORKER_MAX_BATCH).  A batching window was measured in round 4 (no gain in rate at p99 <= 100 ms) and its knob
    removed in round 5.  The summary line names, per variant, the highest rate whose p99 stayed
    within 100 ms, beside the saturated rate, and the service's busy host cores per 1000 requests/s."""
    from PIL import Image
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from synth_data import cfg2_sample
    img = cfg2_sample(0, 960, 960, 32)[0]
    path = f"/tmp/ocr_load_{os.getpid()}.jpg"
    Image.fromarray(img[:, :, ::-1]).save(path, quality=90, subsampling=2)
    fmt = "jpeg 960x960 q90 4:2:0 (%d KB)" % (os.path.getsize(path) // 1024)
    op = {"OCR_WORKER_DET_LIMIT": "960", "OCR_WORKER_REC_H": "48", "OCR_WORKER_REC_W": "320", "OCR_WORKER_CLS": "1", "OCR_DEVICE_JPEG": "1"}
    variants = [("batch<=16 (shipped default)", 16, {}), ("batch<=32", 32, {}), ("batch<=64", 64, {})]
    summary = []
    for name, mb, env in variants:
        rows = []
        for clients in (8, 16, 24, 32, 48, 64, 96):
            run(mb, clients, max(6, per * 64 // clients), path, fmt, workers=2, extra_env=dict(op, **env), tag=name)   # ~64 * per requests per run
            rows.append(RESULTS[-1])
        ok = [r for r in rows if r["p99_ms"] <= 100.0]
        best = max(ok, key=lambda r: r["requests_per_s"]) if ok else None
        sat = max(rows, key=lambda r: r["requests_per_s"])
        summary.append({"variant": name,
                        "requests_per_s_at_p99_le_100ms": best["requests_per_s"] if best else None,
                        "clients_there": best["clients"] if best else None, "p50_ms_there": best["p50_ms"] if best else None,
                        "p99_ms_there": best["p99_ms"] if best else None,
                        "saturated_requests_per_s": sat["requests_per_s"], "saturated_p99_ms": sat["p99_ms"], "saturated_clients": sat["clients"],
                        "host_cores_per_1000_requests_per_s": sat["host_cores_per_1000_requests_per_s"]})
    print(json.dumps({"mode": "sweep_summary", "image": fmt, "gpu_workers": 2, "variants": summary}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "sweep":
        subprocess.check_call(["make", "-s", "-C", HOST])
        import synth_weights
        synth_weights.ensure(ROOT)
        sweep_mode(int(sys.argv[2]) if len(sys.argv) > 2 else 12)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "jpeg":
        subprocess.check_call(["make", "-s", "-C", HOST])
        import synth_weights
        synth_weights.ensure(ROOT)
        jpeg_mode(int(sys.argv[2]) if len(sys.argv) > 2 else 64, int(sys.argv[3]) if len(sys.argv) > 3 else 16)
        sys.exit(0)
    clients = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    subprocess.check_call(["make", "-s", "-C", HOST])
    card = np.load(os.path.join(ROOT, "tests", "golden", "card_jd_bgr.npy"))  # the tests' card image
    import synth_weights
    synth_weights.ensure(ROOT)
    # uncompressed PPM: the request cost is the pipeline, not PNG inflate on the host cores
    path = f"/tmp/ocr_load_{os.getpid()}.ppm"
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (card.shape[1], card.shape[0]) + card[:, :, ::-1].tobytes())
    for mb in (1, 32):
        run(mb, clients, per, path, "ppm %dx%d" % (card.shape[1], card.shape[0]))
