"""Benchmark of the hot path: images/sec end-to-end (det + cls + rec) at 960x960 on N MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` without a launcher starts the N ranks itself (before anything touches the GPU); under a launcher
(WORLD_SIZE set) `--gpus` must equal the world size.  One rank = one process = one GPU.

One step = one pass of the whole pipeline (OCRWorker::processRequest, batched) over one batch of 64
synthetic 960x960 card images per GPU (BASELINE.json configs[1]); inputs are resident in HBM when
the timed region starts.  det/rec weights are seeded synthetic (the reference ships none), so the
SURVEY.md section-8d protocol applies: the det network runs in full and is timed, while
thresholding/box extraction/recognition consume a synthetic probability map rendered from the same
text-line layout so that box counts and rec batch shapes are controlled.  cls uses the real weights.
Whole images are sharded over ranks (image i -> rank i mod N, gpu_worker_pool.cpp:46-59); there is no
data-path collective (weak scaling).  After the timed region every rank's results travel to all ranks as
fixed-size records in ONE all_gather (RCCL over xGMI) and each rank re-computes a few images of its
neighbour's shard to check the gathered records against what a single rank produces for the same seeds.
"""
import argparse
import importlib.util
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

BATCH = 64
H = W = 960
K_LINES = 32
FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
HBM_PEAK_GBS = 8000.0
GATHER_CAP = BATCH * 64        # records per rank in the gather block (a cfg2 image has 32 words)


def load_gather():
    """cpp-paddle-ocr_amd/result_gather.py by path (numpy only: safe before the ranks exist)."""
    spec = importlib.util.spec_from_file_location("ocr_result_gather", os.path.join(ROOT, "cpp-paddle-ocr_amd", "result_gather.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def shard_seeds(rank, world, batch=BATCH):
    """Image i of the stream goes to rank i mod world (gpu_worker_pool sharding): rank r gets seeds
    r, r+world, ...  Returned as cfg2 sample indices."""
    return [rank + world * j for j in range(batch)]


def _cfg2_cached(s):
    from synth_data import cfg2_sample
    cache = os.path.join("/tmp", "ocr_bench_cache")
    os.makedirs(cache, exist_ok=True)
    fn = os.path.join(cache, "cfg2_%d.npz" % s)
    if os.path.exists(fn):
        try:
            z = np.load(fn)
            return z["img"], z["prob"]
        except Exception:
            pass
    img, prob, _ = cfg2_sample(s, H, W, K_LINES)
    try:
        tmp = fn + ".%d.tmp.npz" % os.getpid()
        np.savez(tmp, img=img, prob=prob)
        os.replace(tmp, fn)
    except OSError:
        pass
    return img, prob


def make_inputs(seeds, workers=1):
    """Synthetic images + probability maps of the given cfg2 sample indices (generated on `workers` host
    processes; must run before this process touches the GPU when workers > 1: the pool forks)."""
    if workers > 1 and len(seeds) > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(min(workers, len(seeds))) as pool:
            res = pool.map(_cfg2_cached, seeds)
    else:
        res = [_cfg2_cached(s) for s in seeds]
    return np.stack([r[0] for r in res]), np.stack([r[1] for r in res])


# ------------------------------------------------------------------------------------------------ stub pipeline
class StubPipe:
    """CPU stand-in for the HIP pipeline (`--stub-pipeline`): a deterministic function of the image bytes.  It
    exists so that the launcher, the sharding, the timing reduction and the result gather of this file can be
    rehearsed with gloo ranks on a machine without a GPU (tests/test_dist_sharding.py).  Never a measurement."""
    times = (0.0, 0.0, 0.0)

    def words_of(self, img):
        k = int(img[0, 0, 0]) % 3 + 1
        base = int(img.astype(np.int64).sum() % 100003)
        return [dict(box=np.arange(8, dtype=np.int32).reshape(4, 2) + base + j, ids=np.arange(j + 2, dtype=np.int32) + base % 97,
                     confidence=float((base % 1000) / 1000.0)) for j in range(k)]

    def run_host(self, imgs):
        return [self.words_of(im) for im in imgs]


def stub_inputs(seeds):
    imgs = np.zeros((len(seeds), 8, 8, 3), np.uint8)
    for i, s in enumerate(seeds):
        imgs[i] = np.random.RandomState(1000 + s).randint(0, 255, (8, 8, 3))
    return imgs, None


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(imgs, probs, budget_s=25.0):
    """The CPU oracle (this build's restatement of the reference CPU path: the reference itself needs
    Paddle Inference + OpenCV, absent here) on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from pipeline import Pipeline, DetCfg
    import oracle as O
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or O.usable_cores()
    pipe = Pipeline(det_cfg=DetCfg(limit_side_len=960), rec_batch_num=16, rec_img_h=48, rec_img_w=320, enable_cls=True)
    t0 = time.time()
    done = 0
    for i in range(len(imgs)):
        img = imgs[i].copy()
        x, _ = O.det_preprocess(img, H, W)
        pipe.det.run(x[None])  # full det network (timed, result replaced by the synthetic map as on the GPU)
        boxes = O.det_post(probs[i], 0.2, 0.4, 1.8, H, W)
        views = []
        for b in boxes:
            r = O.crop_rect(b, H, W)
            if r:
                xx, yy, ww, hh = r
                views.append(img[yy:yy + hh, xx:xx + ww])
        if views:
            labels, _ = pipe.cls_run(views)
            for k, v in enumerate(views):
                if labels[k] == 1:
                    O.rotate180_inplace(v)
            pipe.rec_run(views)
        done += 1
        if time.time() - t0 > budget_s:
            break
    dt = time.time() - t0
    return {"value": done / dt, "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": "%d of the %d images of rank 0's batch, same pipeline and parameters, %.1f s" % (done, len(imgs), dt)}


# ------------------------------------------------------------------------------------------------ main
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-request latency calls (profiling runs)")
    ap.add_argument("--verify-images", type=int, default=4,
                    help="images of the neighbour rank's shard each rank re-computes to check the gathered records")
    ap.add_argument("--stub-pipeline", action="store_true",
                    help="CPU rehearsal of launcher + sharding + gather with a stub pipeline and gloo (no measurement)")
    return ap.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    if args.gpus < 1:
        sys.exit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: become one.  Nothing in this process has touched the GPU (only numpy is imported).
        sys.exit(load_gather().launch_ranks(os.path.abspath(__file__), sys.argv[1:] if argv is None else list(argv), args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    G = load_gather()
    stub = args.stub_pipeline
    batch = 8 if stub else BATCH

    # ---- inputs first (host processes), then the GPU
    seeds = shard_seeds(rank, world, batch)
    nb = (rank + 1) % world
    vcount = max(0, min(args.verify_images, batch))
    vseeds = shard_seeds(nb, world, batch)[:vcount]
    if stub:
        imgs, probs = stub_inputs(seeds)
        vimgs, vprobs = stub_inputs(vseeds) if vcount else (None, None)
    else:
        workers = max(1, min(16, (os.cpu_count() or 2) // max(1, world)))
        if any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD")) \
                or any(k.startswith("ROCPROF") for k in os.environ):
            workers = 1   # a profiler's preloaded library has initialised the GPU already: do not fork this process
        imgs, probs = make_inputs(seeds, workers)
        vimgs, vprobs = make_inputs(vseeds, workers) if vcount else (None, None)

    dist = None
    device = None
    backend = None
    if world > 1 or os.environ.get("OCR_BENCH_FORCE_DIST"):  # the env switch rehearses the RCCL path with one rank
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if stub:
            backend = "gloo"
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            backend = "nccl"   # RCCL on ROCm
            torch.cuda.set_device(local)
            device = torch.device("cuda", local)
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=device)

    pkg = pipe = d_imgs = d_probs = None
    if stub:
        pipe = StubPipe()
        run_step = lambda collect=False: pipe.run_host(imgs) if collect else sum(len(w) for w in pipe.run_host(imgs))
        sync = lambda: None
    else:
        from __graft_entry__ import load_package
        pkg = load_package()
        pipe = pkg.Pipe(device=local, enable_cls=True, limit_side_len=960, rec_batch_num=16, rec_img_h=48, rec_img_w=320)
        d_imgs = pkg.DevArray(imgs)
        d_probs = pkg.DevArray(probs)
        run_step = lambda collect=False: pipe.run_device(d_imgs, H, W, BATCH, d_probs, collect=collect)
        sync = lambda: pkg.check(pkg.lib().ocr_dev_sync())

    def barrier():
        sync()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        run_step()
    survey = primary = None
    kernel_timing = not args.no_kernel_timing and not stub
    if kernel_timing:
        # One untimed survey pass with HIP events around EVERY network launch (on the launch stream) finds the
        # dominant kernel and gives the per-kernel table; in the timed region only that kernel carries events
        # (a thousand event pairs per step cost ~4% of the step).
        pipe.timing(True)
        run_step()
        survey = pipe.timing_report()
        # The odd-width rec launches (16-32 lines) share the GPU with the big one on a second stream: their
        # event spans are mostly time spent waiting for free CUs, not kernel time.  The dominant kernel is
        # looked for among the det launches and the rec launch with the most lines.
        def lines_of(name):
            return int(name.split("@")[1].split("x")[0])
        rec_max = max(lines_of(k) for k in survey if k.startswith("rec."))
        primary = {k: v for k, v in survey.items() if not k.startswith("rec.") or lines_of(k) == rec_max}
        dominant = max(primary.items(), key=lambda kv: kv[1]["ms"])[0]
        pipe.timing(True, only=dominant)   # also resets the accumulated timings
    step_ms = []
    nwords = 0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        s0 = time.perf_counter()
        nwords = run_step()
        step_ms.append((time.perf_counter() - s0) * 1e3)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    stage_ms = list(pipe.times)
    rep_timed = None
    if kernel_timing and rank == 0:
        rep_timed = pipe.timing_report()   # the dominant kernel's launches inside the timed region
    if kernel_timing:
        pipe.timing(False)                 # (resets the accumulated timings: read them first)

    # ---- result gather (after the timed region): every rank's words of one step as fixed-size records
    gather = None
    words = run_step(collect=True)
    recs, nrec = G.pack_records(words, seeds, GATHER_CAP)
    if stub and os.environ.get("OCR_BENCH_STUB_CORRUPT") == str(rank):
        recs[0, 2] ^= 1   # rehearsal of a rank that reports a wrong box: its neighbour's check must catch it
    if dist is not None:
        import torch
        sync()
        g0 = time.perf_counter()
        allrec = G.gather_records(dist, recs, device)
        gather_ms = (time.perf_counter() - g0) * 1e3
        ok = True
        if vcount:
            # what a single rank produces for the same seeds: this rank runs the first images of its neighbour's
            # shard itself and compares with the records that came over the wire
            if stub:
                vwords = pipe.run_host(vimgs)
            else:
                dv_i, dv_p = pkg.DevArray(vimgs), pkg.DevArray(vprobs)
                vwords = pipe.run_device(dv_i, H, W, vcount, dv_p, collect=True)
                dv_i.free()
                dv_p.free()
            mine, _ = G.pack_records(vwords, vseeds, GATHER_CAP)
            got = G.records_by_image(allrec[nb])
            want = G.records_by_image(mine)
            ok = all(s in got and len(got[s]) == len(want.get(s, [])) and all(np.array_equal(a, b) for a, b in zip(got[s], want[s]))
                     for s in vseeds) and len(want) > 0
        ok = ok and np.array_equal(allrec[rank], recs)
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        counts = [int((allrec[r][:, 0] >= 0).sum()) for r in range(world)]
        gather = {"backend": backend, "ranks": dist.get_world_size(), "records": int(sum(counts)), "records_per_rank": counts,
                  "bytes_per_rank": int(recs.nbytes), "all_gather_ms": gather_ms,
                  "verified_images_per_rank": vcount, "matches_single_rank": bool(flag.item() == 1)}
        if not gather["matches_single_rank"]:
            if rank == 0:
                print(json.dumps({"error": "gathered records differ from single-rank results", "gather": gather}), flush=True)
            dist.barrier()
            dist.destroy_process_group()
            sys.exit(3)

    # one request at a time (the other half of BASELINE.json's metric, "p50 ms/image"): same protocol, one
    # resident 960x960 image with its 32 lines per call, after the timed region
    single_ms = []
    if rank == 0 and not stub:
        for i in range(0 if args.no_latency else 25):
            s0 = time.perf_counter()
            pipe.run_device(d_imgs, H, W, 1, d_probs, collect=False)
            if i >= 5:
                single_ms.append((time.perf_counter() - s0) * 1e3)

    if rank == 0:
        n_ranks = dist.get_world_size() if dist is not None else 1
        out = {
            "metric": "images/sec end-to-end (det+cls+rec) at 960x960",
            "value": n_ranks * batch * args.steps / elapsed,
            "unit": "images/sec",
            "n_gpus": n_ranks,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "stub (launcher/gather rehearsal, not a measurement)" if stub else "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: batch=64 synthetic 960x960 card images per GPU, "
                                   "PP-OCRv4 mobile det (limit_side_len=960) + cls + rec (48x320, rec_batch_num=16), "
                                   "%d text lines/image, seeded synthetic det/rec weights with the SURVEY 8d "
                                   "probability-map protocol, real cls weights" % K_LINES,
                       "images_per_step_per_gpu": batch, "sharding": "image i -> rank i mod n_gpus, no data-path collective; "
                                                                     "results all-gathered as 64-byte records after the timed region"},
            "p50_step_ms": statistics.median(step_ms),
            "p50_ms_per_image": statistics.median(step_ms) / batch,
            "stage_ms_last_step": {"det": stage_ms[0], "cls": stage_ms[1], "rec": stage_ms[2]},
            "words_per_step": nwords,
        }
        if gather:
            out["gather"] = gather
        if single_ms:
            out["single_image_latency_ms"] = {"p50": statistics.median(single_ms), "p90": sorted(single_ms)[int(len(single_ms) * 0.9)],
                                              "what": "one resident 960x960 image (32 lines) per call, det+cls+rec, 20 calls"}
        if kernel_timing:
            rep = rep_timed
            if rep:
                top = max(rep.items(), key=lambda kv: kv[1]["ms"])
                name, r = top
                avg_ms = r["ms"] / max(1, r["count"])
                tflops = r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] > 0 else 0.0
                traffic = None
                for tf in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("pmc_traffic.json")), reverse=True):
                    # HBM bytes per launch from the rocprofv3 --pmc passes (tools/pmc_traffic.py), latest round first
                    pm = json.load(open(os.path.join(ROOT, "profiles", tf)))
                    if pm.get("kernel") == name:
                        traffic = pm["traffic_bytes_per_launch"]
                        break
                out["roofline"] = {"kernel": name, "bound": "mfma", "achieved": tflops, "peak": FP32_MFMA_PEAK_TFLOPS,
                                   "unit": "TFLOP/s", "frac": tflops / FP32_MFMA_PEAK_TFLOPS, "traffic": traffic,
                                   "avg_launch_ms": avg_ms, "launches": r["count"],
                                   "algorithmic_flops_per_launch": r["flops"] / max(1, r["count"]),
                                   "algorithmic_bytes_per_launch": r["bytes"] / max(1, r["count"]),
                                   "hbm_GBps_algorithmic": r["bytes"] / (r["ms"] * 1e-3) / 1e9 if r["ms"] > 0 else 0.0}
            if survey:   # per-kernel shares from the untimed survey pass (one step, every launch timed)
                tot = sum(v["ms"] for v in primary.values())
                out["kernel_time_share_top5"] = {k: round(v["ms"] / tot, 4) for k, v in
                                                 sorted(primary.items(), key=lambda kv: -kv[1]["ms"])[:5]}
                out["network_kernel_ms_per_step"] = tot   # det + the big rec launch (see `primary` above)
                if os.environ.get("OCR_BENCH_KERNEL_TABLE"):
                    with open(os.environ["OCR_BENCH_KERNEL_TABLE"], "w") as f:
                        for k, v in sorted(survey.items(), key=lambda kv: -kv[1]["ms"]):
                            f.write("%-40s ms/step %8.3f launches/step %5.1f  TFLOP/s %7.2f  GB/s(alg) %8.1f\n" % (
                                k, v["ms"], v["count"],
                                v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] else 0,
                                v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] else 0))
        if n_ranks == 1 and not args.no_cpu_baseline and not stub:
            out["cpu_baseline"] = cpu_baseline(imgs, probs)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if not stub:
        pipe.close()


if __name__ == "__main__":
    import faulthandler
    faulthandler.enable()   # a crash inside the HIP library or RCCL leaves a stack on stderr, not an empty line
    main()
