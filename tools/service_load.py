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


def run(max_batch, clients, per_client, img_path, fmt):
    sock = f"/tmp/ocr_load_{os.getpid()}_{max_batch}.sock"
    env = dict(os.environ, OCR_WORKER_MAX_BATCH=str(max_batch))
    proc = subprocess.Popen([os.path.join(HOST, "ocr_service"), "--model-dir", os.path.join(ROOT, "models"), "--pipe-name", sock,
                             "--gpu-workers", "1"], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT)
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
        t0 = time.perf_counter()
        [t.start() for t in th]
        [t.join() for t in th]
        dt = time.perf_counter() - t0
        lat.sort()
        print(json.dumps({"max_batch": max_batch, "clients": clients, "requests": len(lat), "image": fmt,
                          "requests_per_s": round(len(lat) / dt, 1), "p50_ms": round(lat[len(lat) // 2], 1),
                          "p99_ms": round(lat[int(len(lat) * 0.99) - 1], 1), "words_per_image": len(r["words"])}))
        call(s0, {"command": "shutdown"})
        proc.wait(timeout=30)
    finally:
        if proc.poll() is None:
            proc.kill()


if __name__ == "__main__":
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
