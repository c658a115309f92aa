"""Benchmark of the hot path: images/sec end-to-end (det + cls + rec) at 960x960 on N MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` without a launcher starts the N ranks itself (before anything touches the GPU); under a launcher
(WORLD_SIZE set) `--gpus` must equal the world size.  One rank = one process = one GPU.

One step = one pass of the whole pipeline (OCRWorker::processRequest, batched) over one batch of 64
synthetic 960x960 card images per GPU (BASELINE.json configs[1]); inputs are resident in HBM when
the timed region starts (`value`; the same batch handed over as HOST buffers every step is `host_input`).
The pipeline runs a batch as two chains on two parts of it (ocr_pipe_cfg.phases = 2, the default); the per-kernel
`roofline` comes from a second timed region on a single-chain pipeline (phases = 1: one kernel at a time owns the
device, so a launch's HIP-event duration is that kernel's), reported with its own images/sec as `single_chain`.  det/rec weights are seeded synthetic (the reference ships none), so the
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

# hardware queues the HIP runtime multiplexes streams onto (its default of 4 serialises the pipeline's chains).  The
# library sets the same default when it is loaded; under a launcher torch / RCCL initialise HIP before that, so every
# rank of every run gets it here.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

BATCH = 64
H = W = 960
K_LINES = 32
FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
FP16_MFMA_PEAK_TFLOPS = 2500.0  # dense f16 matrix peak (same guide); the fp16 mode's kernels all sit below its ridge
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


def _cfg3_item(args):
    """cfg3 sample i with its probability map at the detector's input size (limit 960 'max', multiples of 32)."""
    from synth_data import cfg3_item
    return cfg3_item(*args)


def make_cfg3_inputs(ids, workers=1):
    """-> (list of BGR images of mixed sizes, list of probability maps at each image's det input size)"""
    items = [(i, 960) for i in ids]
    if workers > 1 and len(items) > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(min(workers, len(items))) as pool:
            res = pool.map(_cfg3_item, items, chunksize=4)
    else:
        res = [_cfg3_item(it) for it in items]
    return [r[0] for r in res], [r[1] for r in res]


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
def cpu_baseline(budget_s=8.0):
    """The reference's CPU deployment shape timed beside the GPU number, on the box's own host cores.
    Paddle Inference + MKLDNN and OpenCV are absent (the reference ships headers only), so the legs are stand-ins,
    each one or more `tools/cpu_baseline_worker.py` processes (fresh interpreters: this process has used the GPU):
      B1  the oracle (this build's restatement of the reference CPU path; bit-exact contract arithmetic, a checker)
      B2  torch-CPU / oneDNN interpreting the same .pdmodel graphs (closest proxy for Paddle+MKLDNN kernel quality)
    each as ONE worker with det 2 / cls 1 / rec 2 math threads (ocr_worker.cpp:16-18) and as W = floor(0.8*cores/3)
    such workers (ocr_worker.cpp:345-349), on cfg2 images (bounded sample: every leg runs ~budget_s seconds) and on
    the reference's own card image with the worker's default parameters (cfg1)."""
    import subprocess
    cores = host_cores_allowed()
    W = max(1, int(0.8 * cores / 3))
    script = os.path.join(ROOT, "tools", "cpu_baseline_worker.py")

    def leg(engine, workload, nworkers, budget):
        env = dict(os.environ)
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
            env.pop(k, None)
        procs = [subprocess.Popen([sys.executable, script, "--engine", engine, "--workload", workload, "--threads", "2",
                                   "--budget", str(budget), "--first", str(8 * w)], env=env, stdout=subprocess.PIPE,
                                  stderr=subprocess.DEVNULL, text=True) for w in range(nworkers)]
        res = []
        for p_ in procs:
            try:
                out, _ = p_.communicate(timeout=budget * 6 + 240)
                res.append(json.loads([l for l in out.splitlines() if l.startswith("{")][-1]))
            except Exception:
                p_.kill()
        if not res:
            return None
        ms = sorted(m for r in res for m in r["ms"])
        return {"images_per_sec": sum(r["images"] for r in res) / max(r["seconds"] for r in res), "workers": len(res),
                "threads_per_worker": 2, "images": sum(r["images"] for r in res), "p50_ms_per_image": ms[len(ms) // 2]}

    out = {}
    for name, engine in (("B1_oracle", "oracle"), ("B2_torch_onednn", "torch")):
        out[name + "_cfg2_1worker"] = leg(engine, "cfg2", 1, budget_s)
        out[name + "_cfg2_%dworkers" % W] = leg(engine, "cfg2", W, budget_s)
        out[name + "_cfg1_1worker"] = leg(engine, "cfg1", 1, budget_s * 0.6)
    tput = {k: v for k, v in out.items() if v and "cfg2" in k}
    best = max(tput.items(), key=lambda kv: kv[1]["images_per_sec"]) if tput else (None, None)
    cpu = ""
    try:
        cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    return {"value": best[1]["images_per_sec"] if best[1] else None, "unit": "images/sec",
            "cores": best[1]["workers"] * 2 if best[1] else 0, "host_cores_allowed": cores, "kind": "port",
            "sample": "best of the reference-shaped legs (%s): every leg is ~%.0f s of the same cfg2 pipeline and parameters on "
                      "fresh images per worker; host has %d usable cores (%s)" % (best[0], budget_s, cores, cpu),
            "legs": out}


# ------------------------------------------------------------------------------------------------ kernel groups
def group_key(name, r):
    """Launch rows are named `<net>.<op index>.<descriptor>@<bound shape>`.  Ops with the same descriptor (kernel kind,
    channels, strides, gate) that do the same algorithmic work per launch on the same bound shape run the SAME kernel
    instantiation on the SAME grid (e.g. rec ops 13/15/17/19: four launches of `dwpw_kernel<5,1,1,16,...>`): they are one
    row of a rocprofv3 kernel summary grouped by (symbol, grid), and one group here."""
    net, _, rest = name.split(".", 2)
    desc, _, shape = rest.partition("@")
    per = max(1, r["count"])
    return (net, desc, shape, int(round(r["flops"] / per)), int(round(r["bytes"] / per)))


def kernel_groups(rep):
    """timing report {launch name: {ms, count, flops, bytes}} -> {group key: {ms, count, flops, bytes, ops}}"""
    out = {}
    for name, r in rep.items():
        g = out.setdefault(group_key(name, r), dict(ms=0.0, count=0, flops=0.0, bytes=0.0, ops=[]))
        for k in ("ms", "count", "flops", "bytes"):
            g[k] += r[k]
        g["ops"].append(name.split(".")[1])
    return out


def group_label(key, g):
    return "%s.%s@%s [ops %s]" % (key[0], key[1], key[2], ",".join(sorted(g["ops"])))


RIDGE = FP32_MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)   # f32 FLOP per HBM byte where the two roofs meet


def roof_of(flops, nbytes, ms, mfma_peak=FP32_MFMA_PEAK_TFLOPS):
    """algorithmic work over a duration against the roof that bounds it (matrix peak above the ridge, HBM below)"""
    sec = ms * 1e-3
    tflops = flops / sec / 1e12 if sec > 0 else 0.0
    gbps = nbytes / sec / 1e9 if sec > 0 else 0.0
    hbm = nbytes > 0 and flops / nbytes < mfma_peak * 1e12 / (HBM_PEAK_GBS * 1e9)
    return {"bound": "hbm" if hbm else "mfma", "tflops": tflops, "frac_mfma": tflops / mfma_peak,
            "hbm_GBps_algorithmic": gbps, "frac_hbm": gbps / HBM_PEAK_GBS}


def fp16_traffic(name):
    """HBM bytes per launch of the fp16 mode's dominant kernel from the PMC passes of `bench.py --precision fp16` (tools/run_profile.sh
    fp16 -> profiles/*pmc_traffic_fp16.json), or None"""
    pdir = os.path.join(ROOT, "profiles")
    for tf in sorted((f for f in os.listdir(pdir) if f.endswith("pmc_traffic_fp16.json")), reverse=True):
        pm = json.load(open(os.path.join(pdir, tf)))
        if pm.get("kernel") == name and pm.get("durations_agree_within_10pct", False):
            return pm["traffic_bytes_per_launch"]
    return None


def fp16_leg(mk_pipe, run_of, ref_words, batch, steps, sync):
    """The opt-in precision = "fp16" mode (the reference's TensorRT precision switch, ocr_det.cpp:50-56) on the same
    resident batch: f16 activation tensors, f16 matrix products with f32 accumulation.  NEVER `value` - the
    arithmetic is narrower than the reference's CPU path.  Reports its rate, how far its words are from the fp32 run's,
    and the roofline of ITS dominant kernel (HBM-bound: every kernel of this mode is below the f16 ridge)."""
    pipe = mk_pipe(0, "fp16")
    run = run_of(pipe)
    run(False)
    run(False)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        run(False)
    sync()
    el = time.perf_counter() - t0
    words = run(True)
    stage = list(pipe.times)
    pipe.close()
    tot = same_ids = same_box = 0
    dconf = 0.0
    for wa, wb in zip(ref_words, words):
        if len(wa) != len(wb):
            tot += max(len(wa), len(wb))
            continue
        for a, b in zip(wa, wb):
            tot += 1
            same_box += bool(np.array_equal(a["box"], b["box"]))
            same_ids += bool(np.array_equal(a["ids"], b["ids"]))
            dconf = max(dconf, abs(a["confidence"] - b["confidence"]))
    out = {"value": batch * steps / el, "unit": "images/sec", "ms_per_step": el * 1e3 / steps, "steps": steps,
           "stage_ms_last_step": dict(zip(("det", "cls", "rec"), stage)),
           "vs_fp32_words": {"words": tot, "identical_boxes": same_box / max(1, tot), "identical_id_sequences": same_ids / max(1, tot),
                             "max_abs_confidence_diff": dconf},
           "what": "precision = \"fp16\" on every stage: activation tensors stored as f16, f16 matrix instructions (v_mfma_f32_32x32x16_f16 in the big 1x1 and the 3x3 96-channel convs, 32x32x8 elsewhere) with f32 accumulation, reductions and epilogues in f32; "
                   "the depthwise taps of the fused blocks - the mode's dominant kernel - stay f32 VALU arithmetic on f16-stored operands, so this is a STORAGE and matrix-instruction mode, not f16 arithmetic throughout; "
                   "measured distance from the fp32 path at the logits (tools/fp16_check.py, profiles/r5_fp16_check.txt): max |dlogit| 6.8e-3 classifier, 3.4e-3 recognizer - 3-7x outside the north star's 1e-3, "
                   "which is why the mode is opt-in, tolerance-tested and an extra key, never `value` (narrower arithmetic than the reference's CPU path)"}
    # its dominant kernel, on a single chain (as the fp32 roofline)
    pipe1 = mk_pipe(1, "fp16")
    run1 = run_of(pipe1)
    run1(False)
    run1(False)
    pipe1.timing(True)
    run1(False)
    survey = pipe1.timing_report()
    key, _ = max(kernel_groups(survey).items(), key=lambda kv: kv[1]["ms"])
    pipe1.timing(True, only=key[1] + "@")
    rsteps = max(2, min(steps, 5))
    for _ in range(rsteps):
        run1(False)
    sync()
    rep = kernel_groups(pipe1.timing_report())
    pipe1.timing(False)
    pipe1.close()
    g = rep.get(key)
    if g:
        roof = roof_of(g["flops"], g["bytes"], g["ms"], FP16_MFMA_PEAK_TFLOPS)
        hbm = roof["bound"] == "hbm"
        out["roofline"] = {"kernel": group_label(key, g), "measured_in": "single_chain", "bound": roof["bound"],
                           "achieved": roof["hbm_GBps_algorithmic"] if hbm else roof["tflops"],
                           "peak": HBM_PEAK_GBS if hbm else FP16_MFMA_PEAK_TFLOPS, "unit": "GB/s" if hbm else "TFLOP/s",
                           "frac": roof["frac_hbm"] if hbm else roof["frac_mfma"], "traffic": fp16_traffic(group_label(key, g)),
                           "avg_launch_ms": g["ms"] / max(1, g["count"]), "launches": g["count"],
                           "algorithmic_flops_per_launch": g["flops"] / max(1, g["count"]),
                           "algorithmic_bytes_per_launch": g["bytes"] / max(1, g["count"]), "tflops": roof["tflops"]}
    tot_ms = sum(v["ms"] for v in survey.values())
    out["network_kernel_ms_per_step"] = tot_ms
    # the step as a whole, as roofline.step of the fp32 line: algorithmic work of every network launch over this leg's ms_per_step
    sfl, sby = sum(v["flops"] for v in survey.values()), sum(v["bytes"] for v in survey.values())
    sroof = roof_of(sfl, sby, out["ms_per_step"], FP16_MFMA_PEAK_TFLOPS)
    out.setdefault("roofline", {})["step"] = {"flops": sfl, "algorithmic_bytes": sby, "ms": out["ms_per_step"], "tflops": sroof["tflops"],
                                              "frac_mfma": sroof["frac_mfma"], "hbm_GBps_algorithmic": sroof["hbm_GBps_algorithmic"],
                                              "frac_hbm": sroof["frac_hbm"], "bound": sroof["bound"],
                                              "what": "f16 tensors: 2 bytes per activation element; peaks: dense f16 matrix pipe and 8 TB/s"}
    out["kernel_time_share_top5"] = {group_label(k, g_): round(g_["ms"] / tot_ms, 4) for k, g_ in
                                     sorted(kernel_groups(survey).items(), key=lambda kv: -kv[1]["ms"])[:5]}
    return out


# ------------------------------------------------------------------------------------------------ host placement
def host_cores_allowed():
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:   # the box's CPU share is a cgroup quota, not an affinity mask (256 visible cores, 16 usable)
        q, p_ = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = min(cores, max(1, int(int(q) / int(p_))))
    except (OSError, ValueError):
        pass
    return cores


def _cpulist(text):
    out = []
    for part in text.strip().split(","):
        if part:
            a, _, b = part.partition("-")
            out.extend(range(int(a), int(b or a) + 1))
    return out


def gpu_numa_node(local):
    """NUMA node of this rank's GPU from the KFD topology in sysfs - no HIP call (a rank pins itself BEFORE anything
    initialises the GPU).  GPU nodes (simd_count > 0) come in device order; an io link to a CPU node names the NUMA node.
    ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES lists of indices are followed.  None when the topology cannot be read."""
    try:
        vis = os.environ.get("ROCR_VISIBLE_DEVICES") or os.environ.get("HIP_VISIBLE_DEVICES")
        if vis and all(t.strip().isdigit() for t in vis.split(",")):
            ids = [int(t) for t in vis.split(",")]
            if local < len(ids):
                local = ids[local]
        root = "/sys/class/kfd/kfd/topology/nodes"
        nodes = sorted(int(n) for n in os.listdir(root) if n.isdigit())
        props = lambda path: dict(l.split()[:2] for l in open(path) if len(l.split()) >= 2)
        cpu_nodes = [n for n in nodes if int(props("%s/%d/properties" % (root, n)).get("cpu_cores_count", "0")) > 0]
        gpus = [n for n in nodes if int(props("%s/%d/properties" % (root, n)).get("simd_count", "0")) > 0]
        if local >= len(gpus):
            return None
        links = "%s/%d/io_links" % (root, gpus[local])
        for l in sorted(os.listdir(links)):
            to = int(props("%s/%s/properties" % (links, l)).get("node_to", "-1"))
            if to in cpu_nodes:
                return cpu_nodes.index(to)
    except (OSError, ValueError, KeyError):
        pass
    return None


def pin_rank_to_its_gpus_numa_node(local, world):
    """Host threads of a rank (input generation, staging copies, the pipeline's chain threads) on the cores next to its
    GPU: the CPUs of the GPU's NUMA node that this process may use, divided among the ranks that share the node.  Falls
    back to an even contiguous split of the allowed CPUs.  Returns what it did (reported per rank in the N > 1 line)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return {"pinned": False, "why": "no sched_getaffinity"}
    info = {"pinned": False, "numa_node": None, "cpus": len(allowed)}
    # every per-node computation uses the ranks of THIS node (a multi-node launch has world > GPUs per node)
    try:
        world = max(1, min(world, int(os.environ.get("LOCAL_WORLD_SIZE", world))))
    except ValueError:
        pass
    if world <= 1 or len(allowed) < 2 * world or os.environ.get("OCR_BENCH_NO_PIN"):
        return info
    node = gpu_numa_node(local)
    mine = None
    if node is not None:
        try:
            cpus = [c for c in _cpulist(open("/sys/devices/system/node/node%d/cpulist" % node).read()) if c in set(allowed)]
            sharers = [r for r in range(world) if gpu_numa_node(r) == node]
            if len(cpus) >= 2 * len(sharers):
                k = sharers.index(local)
                per = len(cpus) // len(sharers)
                mine = cpus[k * per:(k + 1) * per]
                info["numa_node"] = node
        except (OSError, ValueError):
            mine = None
    if not mine:
        per = len(allowed) // world
        mine = allowed[local * per:(local + 1) * per]
    try:
        os.sched_setaffinity(0, mine)
        info.update(pinned=True, cpus=len(mine), first_cpu=mine[0], last_cpu=mine[-1])
    except OSError as e:
        info["why"] = str(e)
    return info


# ------------------------------------------------------------------------------------------------ main
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=["cfg2", "cfg3", "cfg4", "cfg5"], default="cfg2",
                    help="cfg2 = BASELINE.json configs[1] (the headline: 64 x 960x960 per GPU, resident inputs); cfg3 = configs[2] "
                         "(512 mixed 640-1280 px images per GPU, resident); cfg4 = configs[3] (10k-image stream of cfg3 images "
                         "sharded i mod N, host inputs through the double-buffered staging)")
    ap.add_argument("--images", type=int, default=0, help="cfg3: images per GPU (default 512); cfg4: stream length (default 10000)")
    ap.add_argument("--precision", choices=["fp32", "fp16"], default="fp32",
                    help="the stages' precision parameter for EVERY leg of this run (exploration; the default line measures fp32 and "
                         "reports the opt-in fp16 mode under its own key)")
    ap.add_argument("--no-fp16", action="store_true", help="skip the fp16 leg of the default line")
    ap.add_argument("--no-cfg5", action="store_true", help="skip the BASELINE configs[4] leg (server det + SVTR-large rec, fp16, batch 32; NOT a "
                                                            "reference artifact) that the default line carries as an extra key")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-request latency calls (profiling runs)")
    ap.add_argument("--no-host-input", action="store_true", help="skip the host-input (PCIe-inclusive) leg")
    ap.add_argument("--no-two-workers", action="store_true", help="skip the two-workers-per-GPU leg")
    ap.add_argument("--verify-images", type=int, default=4,
                    help="images of the neighbour rank's shard each rank re-computes to check the gathered records")
    ap.add_argument("--stub-pipeline", action="store_true",
                    help="CPU rehearsal of launcher + sharding + gather with a stub pipeline and gloo (no measurement)")
    return ap.parse_args(argv)


def under_profiler():
    return any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD")) \
        or any(k.startswith("ROCPROF") for k in os.environ)


def host_input_leg(pipe, imgs, probs, steps):
    """The same batch from HOST memory every step (PCIe-inclusive): ocr_pipe_stage copies it into pinned memory and
    uploads it on the copy stream from a helper thread into the slot that is not running, while the main thread is
    inside ocr_pipe_run_staged on the other slot - SURVEY.md 8e's double-buffered pinned staging.  The probability
    maps of the benchmark protocol are attached to both slots once, outside the timed region."""
    import threading
    pipe.stage(0, imgs, probs)
    pipe.stage(1, imgs, probs)
    pipe.run_staged(0, collect=False)
    pipe.run_staged(1, collect=False)
    pipe.stage(0, imgs)
    t0 = time.perf_counter()
    for k in range(steps):
        s = k & 1
        th = threading.Thread(target=pipe.stage, args=(1 - s, imgs))
        th.start()
        pipe.run_staged(s, collect=False)
        th.join()
    return time.perf_counter() - t0


# ------------------------------------------------------------------------------------------------ cfg5 (BASELINE configs[4])
def srv_kernel_groups(rep):
    """{(network, launch name without its op index): {ms, count, flops, bytes}} - the nine SVTR blocks of one stage launch the
    same kernel on the same shape: one group, as in a rocprofv3 summary"""
    g = {}
    for name, v in rep.items():
        net, rest = name.split(".", 1)
        key = rest.split(".", 1)[1] if rest.split(".", 1)[0].isdigit() else rest
        t = g.setdefault((net, key), {"ms": 0.0, "count": 0, "flops": 0.0, "bytes": 0.0})
        for k in t:
            t[k] += v[k]
    return g


def cfg5_traffic(kernel, ms):
    """HBM bytes per launch of the line's dominant kernel from the PMC passes of this command (tools/run_profile_cfg5.sh ->
    profiles/*cfg5_profile_summary.json, `by_bench_kernel`): only an entry that names THIS (kernel, shape) group and whose traced
    duration agrees with this run's within 15 %, else None"""
    pdir = os.path.join(ROOT, "profiles")
    for tf in sorted((f for f in os.listdir(pdir) if f.endswith("cfg5_profile_summary.json")), reverse=True):
        try:
            e = json.load(open(os.path.join(pdir, tf))).get("by_bench_kernel", {}).get(kernel)
        except (OSError, ValueError):
            continue
        if e and e.get("traffic_bytes_per_launch") and abs(e["avg_ns"] / 1e6 - ms) <= 0.15 * ms:
            return e["traffic_bytes_per_launch"], ("profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, gfx950 correction) of the "
                                                   "same command on %s (%.3f ms there), not this run" % (tf, e["rocprof_kernel"], e["avg_ns"] / 1e6))
    return None, None


def cfg5_cpu_baseline():
    """the oracle's f32 run of the two server plans on the host cores, a bounded sample (about 10-20 s): FOUR 960 x 960 images through
    the detector plan and 32 lines of 48 x 320 (one image's worth) through the recognizer plan, one at a time / four at a time as a
    caller of the reference would; per image = det + 32 lines"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from oracle import OracleNet, usable_cores
    rs = np.random.RandomState(5)
    d = OracleNet("srv_det")
    n_img, n_lines = 8, 64
    t0 = time.perf_counter()
    for _ in range(n_img):
        d.run(rs.randn(1, H, W, 3).astype(np.float32))
    t_det = (time.perf_counter() - t0) / n_img
    r = OracleNet("srv_rec")
    t0 = time.perf_counter()
    for _ in range(n_lines // 4):
        r.run(rs.randn(4, 48, 320, 3).astype(np.float32))
    t_rec = time.perf_counter() - t0
    per_image = t_det + t_rec * (K_LINES / float(n_lines))
    return {"value": 1.0 / per_image, "unit": "images/sec", "cores": usable_cores(), "kind": "port",
            "sample": "oracle (f32 restatement of the hand-written server plans, OpenMP over the usable cores): %d images 960x960 through "
                      "srv_det (%.2f s each) + %d lines 48x320 through srv_rec in batches of 4 (%.1f s), %.0f s of CPU work; per image = det + 32 "
                      "lines; networks only (pre/post-processing is < 1 %% of this)" % (n_img, t_det, n_lines, t_rec, t_det * n_img + t_rec)}


def cfg5_line(args):
    """BASELINE configs[4]: "PP-OCRv4_server_det (ResNet50 backbone) + SVTR-large rec, fp16, batch=32" - NOT a reference
    artifact: hand-written plans from the public PaddleOCR definitions, seeded weights.  One step =
    one pass of the whole pipeline over 32 synthetic 960 x 960 images per GPU (the cfg2 generator, 32 lines each, the
    section-8d probability-map protocol); never the headline `value` of the default line."""
    if args.gpus != 1 or "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != 1:
        sys.exit("--config cfg5 is a single-GPU line (batch 32 per GPU at N = 1; the scaling path is cfg2's)")
    batch = 32
    precision = "fp16" if args.precision == "fp32" and not os.environ.get("OCR_CFG5_FP32") else args.precision
    workers = 1 if under_profiler() else max(1, min(16, os.cpu_count() or 2))
    imgs, probs = make_inputs(list(range(batch)), workers)
    import synth_weights
    synth_weights.ensure_server(ROOT)
    from __graft_entry__ import load_package
    pkg = load_package()
    srv = os.path.join(ROOT, "models_server")
    mk = lambda phases: pkg.Pipe(device=0, enable_cls=True, limit_side_len=960, rec_batch_num=16, rec_img_h=48, rec_img_w=320,
                                 phases=phases, precision=precision, det_dir=os.path.join(srv, "det"), rec_dir=os.path.join(srv, "rec"))
    sync = lambda: pkg.check(pkg.lib().ocr_dev_sync())
    d_imgs, d_probs = pkg.DevArray(imgs), pkg.DevArray(probs)
    pipe = mk(0)
    run_step = lambda: pipe.run_device(d_imgs, H, W, batch, d_probs, collect=False)
    for _ in range(max(1, args.warmup)):
        run_step()   # (the first pass binds the shapes and picks every GEMM's tile configuration on the device)
    sync()
    cpu0 = time.process_time()
    t0 = time.perf_counter()
    step_ms = []
    for _ in range(args.steps):
        s0 = time.perf_counter()
        nwords = run_step()
        step_ms.append((time.perf_counter() - s0) * 1e3)
    sync()
    elapsed = time.perf_counter() - t0
    cpu_s = time.process_time() - cpu0
    stage_ms = list(pipe.times)
    pipe.close()
    # ---- roofline leg: one chain, HIP events around every network launch of one step
    pipe1 = mk(1)
    run1 = lambda: pipe1.run_device(d_imgs, H, W, batch, d_probs, collect=False)
    run1(); run1()
    pipe1.timing(True)
    rsteps = max(2, min(args.steps, 5))
    sync()
    t0r = time.perf_counter()
    for _ in range(rsteps):
        run1()
    sync()
    elr = time.perf_counter() - t0r
    rep = pipe1.timing_report()
    pipe1.timing(False)
    pipe1.close()
    groups = srv_kernel_groups(rep)
    (dnet, dkey), dom = max(groups.items(), key=lambda kv: kv[1]["ms"])
    dom_ms = dom["ms"] / dom["count"]
    ai = dom["flops"] / max(dom["bytes"], 1.0)
    ridge = FP16_MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
    if ai >= ridge:
        ach, peak, unit, bound = dom["flops"] / dom["count"] / dom_ms / 1e9, FP16_MFMA_PEAK_TFLOPS, "TFLOP/s", "mfma"
    else:
        ach, peak, unit, bound = dom["bytes"] / dom["count"] / dom_ms / 1e6, HBM_PEAK_GBS, "GB/s", "hbm"
    tot = {"ms": 0.0, "flops": 0.0, "bytes": 0.0}
    per_net = {}
    for (net, key), v in groups.items():
        pn = per_net.setdefault(net, {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "conv_ms": 0.0, "conv_flops": 0.0})
        for k in ("ms", "flops", "bytes"):
            tot[k] += v[k] / rsteps
            pn[k] += v[k] / rsteps
        if key.startswith(("conv", "deconv", "linear", "head_tail")):
            pn["conv_ms"] += v["ms"] / rsteps
            pn["conv_flops"] += v["flops"] / rsteps
    # every launch against its own binding roof (f16 matrix peak or HBM peak, whichever takes longer)
    roof_ms = sum(max(v["flops"] / (FP16_MFMA_PEAK_TFLOPS * 1e9), v["bytes"] / (HBM_PEAK_GBS * 1e6)) for v in groups.values()) / rsteps
    table = sorted(((v["ms"] / rsteps, net + "." + key, v) for (net, key), v in groups.items()), reverse=True)[:12]
    det = per_net.get("det", {})
    line = {
        "metric": "images/sec end-to-end (det+cls+rec) at 960x960", "value": batch * args.steps / elapsed, "unit": "images/sec",
        "n_gpus": 1, "steps": args.steps, "warmup": max(1, args.warmup), "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f16" if precision == "fp16" else "f32", "data": "synthetic",
        "reference_artifact": False,
        "config": {"workload": "BASELINE configs[4]: batch=32 synthetic 960x960 images, server det (ResNet50-vd + DBFPN + DB head) + SVTR-large "
                               "rec (48x320 lines, 6625 classes) + the reference's cls model, precision %s, 32 text lines per image, 1xMI355X; "
                               "hand-written plans (tools/make_server_plans.py) and seeded weights - the reference ships no such graph "
                               "(SURVEY.md section 8d cfg5); probability-map protocol of section 8d" % precision,
                   "global_batch": batch, "parallelism": "1 GPU"},
        "words_last_step": int(nwords), "stage_ms_last_step": dict(zip(("det", "cls", "rec"), stage_ms)),
        "p50_step_ms": statistics.median(step_ms),
        "host": {"cpu_s_per_step": cpu_s / args.steps, "what": "user + system CPU seconds of this process (all its threads) per timed step"},
        "roofline": {
            "kernel": dnet + "." + dkey, "launches_per_step": dom["count"] // rsteps, "ms_per_launch": dom_ms, "bound": bound,
            "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak, "traffic": cfg5_traffic(dnet + "." + dkey, dom_ms)[0],
            "traffic_source": cfg5_traffic(dnet + "." + dkey, dom_ms)[1],
            "arithmetic_intensity": ai,
            "what": "the (kernel, shape) group with the most time in the step; duration = HIP events on the launch stream, one chain",
            "det_conv_stack": {"tflops": det.get("conv_flops", 0) / max(det.get("conv_ms", 1e-9), 1e-9) / 1e9,
                               "frac_of_f16_mfma_peak": det.get("conv_flops", 0) / max(det.get("conv_ms", 1e-9), 1e-9) / 1e9 / FP16_MFMA_PEAK_TFLOPS,
                               "ms": det.get("conv_ms", 0), "gflop": det.get("conv_flops", 0) / 1e9,
                               "what": "every conv / transposed conv launch of the server detector in one step (32 images)"},
            "step": {"kernel_ms": tot["ms"], "tflop": tot["flops"] / 1e12, "gb": tot["bytes"] / 1e9,
                     "frac_mfma": tot["flops"] / max(tot["ms"], 1e-9) / 1e9 / FP16_MFMA_PEAK_TFLOPS,
                     "frac_hbm": tot["bytes"] / max(tot["ms"], 1e-9) / 1e6 / HBM_PEAK_GBS,
                     "layer_roof_frac": roof_ms / max(tot["ms"], 1e-9),
                     "what": "network launches of one step; layer_roof_frac = sum over launches of max(flops / 2.5 PF, bytes / 8 TB/s) "
                             "over their measured time"},
            "per_network_ms": {k: v["ms"] for k, v in per_net.items()},
            "top_kernels": [{"kernel": n, "ms_per_step": ms, "launches": v["count"] // rsteps,
                             "tflops": v["flops"] / max(v["ms"], 1e-9) / 1e9, "gbs": v["bytes"] / max(v["ms"], 1e-9) / 1e6} for ms, n, v in table],
            "single_chain_images_per_s": batch * rsteps / elr,
        },
        "parity": "tests/test_gpu_round6.py: every plan tensor of both networks - the f32 twin of the same launch list == the oracle bit for "
                  "bit, the f16 build within stated tolerances of the oracle; the pipeline's f16 words against the f32 twin's",
    }
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = cfg5_cpu_baseline()
    return line


def main_cfg5(args):
    print(json.dumps(cfg5_line(args)))


def main(argv=None):
    args = parse_args(argv)
    if args.gpus < 1:
        sys.exit("--gpus must be >= 1")
    if args.config == "cfg5":
        return main_cfg5(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: become one.  Nothing in this process has touched the GPU (only numpy is imported).
        sys.exit(load_gather().launch_ranks(os.path.abspath(__file__), sys.argv[1:] if argv is None else list(argv), args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries ONE JSON line (rank 0).  Libraries write to file descriptor 1 behind Python's back (RCCL prints a version
    # banner there when its communicator comes up): from here on fd 1 IS stderr, and the line goes to the saved descriptor.
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)

    def emit_line(obj):
        os.write(line_fd, (json.dumps(obj) + "\n").encode())

    if args.gpus != world:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    G = load_gather()
    stub = args.stub_pipeline
    share_gpu = bool(os.environ.get("OCR_BENCH_SHARE_GPU")) and not stub
    dev_index = 0 if share_gpu else local
    cfg = args.config
    if stub and cfg != "cfg2":
        sys.exit("--stub-pipeline rehearses cfg2 only")
    # before anything of this rank touches the GPU (and before its input generators fork): host threads next to the GPU
    placement = pin_rank_to_its_gpus_numa_node(local, world) if not stub else {"pinned": False}
    cores_mine = host_cores_allowed()
    workers = max(1, min(16, cores_mine if world > 1 else (os.cpu_count() or 2)))
    if under_profiler():
        workers = 1   # a profiler's preloaded library has initialised the GPU already: do not fork this process
    t_inputs = time.perf_counter()

    # ---- inputs first (host processes), then the GPU.  Image i of the stream belongs to rank i mod world.
    nb = (rank + 1) % world
    if cfg == "cfg2":
        batch = 8 if stub else BATCH
        seeds = shard_seeds(rank, world, batch)
        vcount = max(0, min(args.verify_images, batch))
        vseeds = shard_seeds(nb, world, batch)[:vcount]
        if stub:
            imgs, probs = stub_inputs(seeds)
            vimgs, vprobs = stub_inputs(vseeds) if vcount else (None, None)
        else:
            imgs, probs = make_inputs(seeds, workers)
            vimgs, vprobs = make_inputs(vseeds, workers) if vcount else (None, None)
        img_list, prob_list = (list(imgs), list(probs)) if not stub else (None, None)
    else:
        # cfg3: `batch` distinct mixed-size images per rank; cfg4: a stream of --images requests over all ranks, drawn
        # cyclically from a pool of 256 distinct cfg3 images per rank (generating 10k distinct ones would take an hour)
        batch = (args.images or 512) if cfg == "cfg3" else 64
        pool_n = batch if cfg == "cfg3" else 256
        seeds = shard_seeds(rank, world, pool_n)
        vcount = max(0, min(args.verify_images, pool_n))
        vseeds = shard_seeds(nb, world, pool_n)[:vcount]
        img_list, prob_list = make_cfg3_inputs(seeds, workers)
        vimgs, vprobs = make_cfg3_inputs(vseeds, workers) if vcount else (None, None)
        imgs = probs = None
    input_gen_s = time.perf_counter() - t_inputs

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
        elif share_gpu:
            # rehearsal on a one-GPU lease (OCR_BENCH_SHARE_GPU=1): every rank drives GPU 0 with the REAL pipeline; RCCL refuses
            # two ranks on one device, so the records travel over gloo - launcher, pinning, sharding, gather and the neighbour
            # re-check are the production ones.  The line says so (`rehearsal`); its rate is N pipelines sharing one GPU.
            backend = "gloo"
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            backend = "nccl"   # RCCL on ROCm
            torch.cuda.set_device(local)
            device = torch.device("cuda", local)
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=device)

    pkg = pipe = d_imgs = d_probs = None
    stream_batches = 0
    if stub:
        pipe = StubPipe()
        run_step = lambda collect=False: pipe.run_host(imgs) if collect else sum(len(w) for w in pipe.run_host(imgs))
        sync = lambda: None
    else:
        from __graft_entry__ import load_package
        pkg = load_package()
        mk_pipe = lambda phases=0, precision=None: pkg.Pipe(device=dev_index, enable_cls=True, limit_side_len=960, rec_batch_num=16, rec_img_h=48,
                                                            rec_img_w=320, phases=phases, precision=precision or args.precision)
        pipe = mk_pipe()
        sync = lambda: pkg.check(pkg.lib().ocr_dev_sync())
        if cfg == "cfg2":
            d_imgs = pkg.DevArray(imgs)
            d_probs = pkg.DevArray(probs)
            run_step = lambda collect=False: pipe.run_device(d_imgs, H, W, BATCH, d_probs, collect=collect)
        elif cfg == "cfg3":
            pipe.stage(0, img_list, prob_list)     # resident in HBM from here on: a step re-runs the staged slot
            run_step = lambda collect=False: pipe.run_staged(0, collect=collect)
        else:
            # cfg4: this rank's share of the stream in batches of 64 through the two staging slots.  Batch b is images
            # b*64 .. b*64+63 of the rank's cyclic pool of `pool_n` distinct images: pool_n / 64 distinct compositions, every
            # one staged from HOST memory when its turn comes; the protocol's probability maps are uploaded once per pool
            # image and attached device-to-device (they stand in for the detector's own output, not for request data).
            total_stream = args.images or 10000
            mine = len(range(rank, total_stream, world))
            stream_batches = (mine + batch - 1) // batch
            d_pool_probs = [pkg.DevArray(p_) for p_ in prob_list]
            comp_of = lambda b: [(b * batch + j) % pool_n for j in range(batch)]
            def stage_batch(slot, b):
                ids_ = comp_of(b)
                pipe.stage_dev_probs(slot, [img_list[i] for i in ids_], [d_pool_probs[i] for i in ids_])
            stage_batch(0, 0)
            run_step = lambda collect=False: pipe.run_staged(0, collect=collect)

    def barrier():
        sync()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        run_step()
    kernel_timing = not args.no_kernel_timing and not stub and cfg in ("cfg2", "cfg3")
    if cfg == "cfg4" and not stub:
        for b in range(pool_n // batch):   # every composition once, untimed: the stream's steady state has seen all its shapes
            stage_batch(0, b)
            pipe.run_staged(0, collect=False)
        stats0 = pipe.stats()
    step_ms = []
    nwords = 0
    barrier()
    cpu_t0 = time.process_time()   # user + system seconds of this process, every thread (the library's chain threads included)
    t0 = time.perf_counter()
    if cfg == "cfg4" and not stub:
        import threading
        steps_done = stream_batches
        stage_batch(0, 0)
        for b in range(stream_batches):
            s0 = time.perf_counter()
            sl = b & 1
            th = threading.Thread(target=stage_batch, args=(1 - sl, b + 1))
            th.start()
            nwords = pipe.run_staged(sl, collect=False)
            th.join()
            step_ms.append((time.perf_counter() - s0) * 1e3)
    else:
        steps_done = args.steps
        for _ in range(args.steps):
            s0 = time.perf_counter()
            nwords = run_step()
            step_ms.append((time.perf_counter() - s0) * 1e3)
    barrier()
    elapsed = time.perf_counter() - t0
    host_cpu_s = time.process_time() - cpu_t0
    # ---- the other way of waiting for a stream (round 6): the same timed steps with the library's host threads
    # spinning in hipStreamSynchronize (what rounds 1-5 did) instead of sleeping on a blocking event (the default since) - images/s
    # and host CPU seconds per step of both modes; rank 0 of a single-GPU resident-batch run only
    wait_modes = None
    if not stub and world == 1 and cfg in ("cfg2", "cfg3") and hasattr(pkg.lib(), "ocr_rt_set_wait_mode"):
        L_ = pkg.lib()
        mode0 = L_.ocr_rt_get_wait_mode()
        wait_modes = {("block" if mode0 else "spin"): {"images_per_s": batch * steps_done / elapsed, "cpu_s_per_step": host_cpu_s / max(1, steps_done)}}
        L_.ocr_rt_set_wait_mode(0 if mode0 else 1)
        run_step()
        sync()
        c0, w0 = time.process_time(), time.perf_counter()
        for _ in range(args.steps):
            run_step()
        sync()
        wait_modes["spin" if mode0 else "block"] = {"images_per_s": batch * args.steps / (time.perf_counter() - w0),
                                                    "cpu_s_per_step": (time.process_time() - c0) / max(1, args.steps)}
        L_.ocr_rt_set_wait_mode(mode0)
        wait_modes["default"] = "block" if mode0 else "spin"
    per_rank = None
    if dist is not None:
        import torch
        # every rank's own view, for the N > 1 line: its rate over its own clock, how long its inputs took to generate,
        # the host cores it may use and whether it is pinned next to its GPU (one small all_gather, outside the timed region)
        mine_t = torch.tensor([elapsed, input_gen_s, float(cores_mine), 1.0 if placement.get("pinned") else 0.0,
                               float(placement.get("numa_node") if placement.get("numa_node") is not None else -1)],
                              dtype=torch.float64, device=device)
        allr = [torch.empty_like(mine_t) for _ in range(world)]
        dist.all_gather(allr, mine_t)
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        rows_ = [[float(v) for v in r.cpu()] for r in allr]
        per_rank = {"seconds": [r[0] for r in rows_], "input_generation_s": [r[1] for r in rows_],
                    "host_cores_allowed": [int(r[2]) for r in rows_], "pinned": [bool(r[3]) for r in rows_],
                    "numa_node": [int(r[4]) if r[4] >= 0 else None for r in rows_]}
    stage_ms = list(pipe.times)
    stream_stats = None
    if cfg == "cfg4" and not stub:
        st1 = pipe.stats()
        runs = st1["runs"] - stats0["runs"]
        stream_stats = {"network_runs": runs, "new_bindings": st1["binds"] - stats0["binds"],
                        "binding_cache_hit_rate": 1.0 - (st1["binds"] - stats0["binds"]) / max(1, runs),
                        "graph_replay_rate": (st1["graph_replays"] - stats0["graph_replays"]) / max(1, runs),
                        "distinct_batch_compositions": pool_n // batch}

    # ---- roofline leg, rank 0 of a single-GPU run: a second timed region of the same steps on a SINGLE-CHAIN pipeline
    # (phases = 1).  One untimed survey pass with HIP events around every network launch (on the launch stream) finds the
    # dominant kernel and gives the per-kernel table; in the timed steps only that kernel carries events (a thousand
    # event pairs per step cost ~4 % of the step).  With two chains a kernel shares the chip with the other chain's
    # kernels and an event span stops being a property of the kernel.
    survey = rep_timed = single_chain = dom_key = None
    if kernel_timing and rank == 0 and world == 1:
        pipe1 = mk_pipe(1)
        if cfg == "cfg2":
            run1 = lambda: pipe1.run_device(d_imgs, H, W, BATCH, d_probs, collect=False)
        else:
            pipe1.stage(0, img_list, prob_list)
            run1 = lambda: pipe1.run_staged(0, collect=False)
        run1()
        run1()
        pipe1.timing(True)
        run1()
        survey = pipe1.timing_report()
        # the dominant kernel = the (instantiation, grid) group with the most time in the step - four launches of one
        # symbol on one shape are ONE kernel, as in a rocprofv3 summary (round 3 picked per op name and reported a 7 % kernel)
        dom_key, _ = max(kernel_groups(survey).items(), key=lambda kv: kv[1]["ms"])
        pipe1.timing(True, only=dom_key[1] + "@")   # events on that group's launches only; also resets the accumulated timings
        rsteps = max(2, min(args.steps, 10))
        sync()
        t0r = time.perf_counter()
        for _ in range(rsteps):
            run1()
        sync()
        elr = time.perf_counter() - t0r
        rep_timed = pipe1.timing_report()
        pipe1.timing(False)
        single_chain = {"value": batch * rsteps / elr, "unit": "images/sec", "ms_per_step": elr * 1e3 / rsteps, "steps": rsteps,
                        "stage_ms_last_step": dict(zip(("det", "cls", "rec"), list(pipe1.times))),
                        "what": "the same resident steps on a pipeline with ONE chain (ocr_pipe_cfg.phases = 1): the timed region the "
                                "roofline figures are measured in"}
        pipe1.close()

    # ---- host-input leg (PCIe-inclusive), rank 0 of a single-GPU run only: same batch from host memory every step
    host_in = None
    if not stub and cfg in ("cfg2", "cfg3") and world == 1 and not args.no_host_input:
        hsteps = max(2, args.steps)
        hel = host_input_leg(pipe, img_list, prob_list, hsteps)
        host_in = {"value": batch * hsteps / hel, "unit": "images/sec", "ms_per_step": hel * 1e3 / hsteps, "steps": hsteps,
                   "what": "the same batch handed over as HOST buffers every step: pinned staging + upload of batch k+1 on a copy "
                           "stream (helper thread, ocr_pipe_stage) while batch k runs (ocr_pipe_run_staged)"}
        if cfg == "cfg3":
            pipe.stage(0, img_list, prob_list)   # slot 0 as the resident batch again (the gather step below re-runs it)

    # ---- two batches in flight on the ONE handle (ocr_pipe_run_device_on): chain c of the handle runs whole batches c, c + 2,
    # ... while the other chain runs the batches between - consecutive batches overlap (the reference's shape: a pool of
    # workers on one GPU, each on its own request) instead of the two halves of one batch.  EXACTLY `steps` batches of 64 in the
    # timed region, barrier + sync on both sides; a batch's latency is what it is with one chain.  An extra key.
    in_flight = None
    # (a handle built with ONE chain - OCR_PIPE_PHASES=1 - has no second chain to name: the leg is skipped)
    one_chain = os.environ.get("OCR_PIPE_PHASES", "") == "1"
    if not stub and cfg == "cfg2" and world == 1 and not args.no_two_workers and hasattr(pipe, "run_device_on") and not one_chain:
        import threading
        for c_ in (0, 1):
            pipe.run_device_on(c_, d_imgs, H, W, BATCH, d_probs, collect=False)
        fsteps = max(2, args.steps)
        def flight(c_):
            for _ in range(c_, fsteps, 2):
                pipe.run_device_on(c_, d_imgs, H, W, BATCH, d_probs, collect=False)
        ths = [threading.Thread(target=flight, args=(c_,)) for c_ in (0, 1)]
        sync()
        t0f = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        sync()
        elf = time.perf_counter() - t0f
        in_flight = {"value": batch * fsteps / elf, "unit": "images/sec", "ms_per_step": elf * 1e3 / fsteps, "steps": fsteps,
                     "what": "the same handle with two batches in flight: chain 0 runs batches 0, 2, 4, ... and chain 1 batches 1, 3, 5, ... "
                             "(ocr_pipe_run_device_on, two host threads), %d batches of %d images in the timed region" % (fsteps, batch)}

    # ---- two pipeline workers sharing the GPU (the reference's pool maps worker i -> GPU i mod n, gpu_worker_pool.cpp:
    # 46-59, so a device may serve several workers): each worker has its own ocr_pipe, streams and arenas and runs the
    # same resident batch `steps` times; while one worker is in its latency-bound phases (det post-processing, cls,
    # the tail of rec) the other one's dense kernels fill the GPU.  An extra key, never `value`: the per-kernel
    # roofline figures above are measured with one worker owning the device.
    two_workers = None
    if not stub and cfg == "cfg2" and world == 1 and not args.no_two_workers:
        import threading
        pipe2 = mk_pipe()
        pipe.stage(0, img_list, prob_list)
        pipe2.stage(0, img_list, prob_list)
        for p_ in (pipe, pipe2):
            p_.run_staged(0, collect=False)
        wsteps = max(2, args.steps)
        def work(p_):
            for _ in range(wsteps):
                p_.run_staged(0, collect=False)
        ths = [threading.Thread(target=work, args=(p_,)) for p_ in (pipe, pipe2)]
        sync()
        t0w = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        sync()
        elw = time.perf_counter() - t0w
        pipe2.close()
        two_workers = {"value": batch * 2 * wsteps / elw, "unit": "images/sec", "ms_per_step": elw * 1e3 / (2 * wsteps), "steps": 2 * wsteps,
                       "what": "two pipeline workers (two ocr_pipe handles, own streams and arenas, two chains each) on this one GPU, "
                               "each running the resident batch %d times concurrently" % wsteps}

    # ---- the opt-in fp16 mode on the same resident batch (cfg2, one GPU): an extra key
    fp16 = None
    if not stub and cfg == "cfg2" and world == 1 and not args.no_fp16 and args.precision == "fp32":
        ref_words = run_step(collect=True)
        fp16 = fp16_leg(mk_pipe, lambda p_: (lambda collect=False: p_.run_device(d_imgs, H, W, BATCH, d_probs, collect=collect)),
                        ref_words, batch, max(2, args.steps), sync)

    # ---- BASELINE configs[4] (server det + SVTR-large rec, fp16, batch 32; hand-written plans, NOT a reference artifact): an extra
    # key of the default line, so that the driver's record holds a number measured by the driver for it; `--config cfg5` prints the
    # full line (roofline tables, CPU baseline).  Never `value`.
    cfg5 = None
    if not stub and cfg == "cfg2" and world == 1 and not args.no_cfg5 and args.precision == "fp32" and not under_profiler():
        try:
            import argparse as _ap
            a5 = _ap.Namespace(**vars(args))
            a5.steps, a5.warmup, a5.no_cpu_baseline, a5.gpus = max(2, min(args.steps, 6)), 1, True, 1
            l5 = cfg5_line(a5)
            r5 = l5["roofline"]
            cfg5 = {"value": l5["value"], "unit": l5["unit"], "ms_per_step": l5["ms_per_step"], "steps": l5["steps"], "dtype": l5["dtype"],
                    "reference_artifact": False, "global_batch": l5["config"]["global_batch"], "workload": l5["config"]["workload"],
                    "roofline": {k: r5[k] for k in ("kernel", "ms_per_launch", "bound", "achieved", "peak", "unit", "frac", "traffic", "det_conv_stack", "step", "per_network_ms")},
                    "what": "BASELINE configs[4] on this GPU, after the headline's timed region (its own inputs, handles and timed region); "
                            "`python bench.py --config cfg5` prints the whole line"}
        except Exception as e:  # (the default line must not die with its extra)
            cfg5 = {"error": repr(e)[:400], "reference_artifact": False}

    # ---- result gather (after the timed region): every rank's words of one step as fixed-size records
    gather = None
    gcap = max(GATHER_CAP, batch * 80)
    words = run_step(collect=True)
    recs, nrec = G.pack_records(words, seeds[:len(words)], gcap)
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
            elif cfg == "cfg2":
                dv_i, dv_p = pkg.DevArray(vimgs), pkg.DevArray(vprobs)
                vwords = pipe.run_device(dv_i, H, W, vcount, dv_p, collect=True)
                dv_i.free()
                dv_p.free()
            else:
                pipe.stage(1, vimgs, vprobs)
                vwords = pipe.run_staged(1, collect=True)
            mine_r, _ = G.pack_records(vwords, vseeds, gcap)
            got = G.records_by_image(allrec[nb])
            want = G.records_by_image(mine_r)
            ok = all(s in got and len(got[s]) == len(want.get(s, [])) and all(np.array_equal(a, b) for a, b in zip(got[s], want[s]))
                     for s in vseeds if s in want or s in got) and len(want) > 0
        ok = ok and np.array_equal(allrec[rank], recs)
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        counts = [int((allrec[r][:, 0] >= 0).sum()) for r in range(world)]
        gather = {"backend": backend, "ranks": dist.get_world_size(), "records": int(sum(counts)), "records_per_rank": counts,
                  "bytes_per_rank": int(recs.nbytes), "all_gather_ms": gather_ms,
                  "verified_images_per_rank": vcount, "matches_single_rank": bool(flag.item() == 1)}
        if not gather["matches_single_rank"]:
            if rank == 0:
                emit_line({"error": "gathered records differ from single-rank results", "gather": gather})
            dist.barrier()
            dist.destroy_process_group()
            sys.exit(3)

    # one request at a time (the other half of BASELINE.json's metric, "p50 ms/image"): same protocol, one
    # resident 960x960 image with its 32 lines per call, after the timed region
    single_ms = []
    if rank == 0 and not stub and cfg == "cfg2":
        for i in range(0 if args.no_latency else 25):
            s0 = time.perf_counter()
            pipe.run_device(d_imgs, H, W, 1, d_probs, collect=False)
            if i >= 5:
                single_ms.append((time.perf_counter() - s0) * 1e3)

    if rank == 0:
        n_ranks = dist.get_world_size() if dist is not None else 1
        workload = {
            "cfg2": "BASELINE.json configs[1]: batch=64 synthetic 960x960 card images per GPU, "
                    "PP-OCRv4 mobile det (limit_side_len=960) + cls + rec (48x320, rec_batch_num=16), "
                    "%d text lines/image, seeded synthetic det/rec weights with the SURVEY 8d "
                    "probability-map protocol, real cls weights" % K_LINES,
            "cfg3": "BASELINE.json configs[2]: batch=%d mixed-aspect 640-1280 px images per GPU (seeds 2000+i), 4-64 text lines each, "
                    "det (limit 960 'max', one pass per distinct size) + cls + rec with the lines of all images pooled, resident in HBM; "
                    "same weights and probability-map protocol as cfg2" % batch,
            "cfg4": "BASELINE.json configs[3]: %d-image stream of cfg3 images, image i -> rank i mod n_gpus; every rank cycles through its "
                    "pool of 256 distinct images in batches of 64 (4 distinct batch compositions, each staged from HOST memory when its "
                    "turn comes, through the double-buffered pinned staging: PCIe-inclusive by construction); same weights and "
                    "protocol as cfg2" % (args.images or 10000),
        }[cfg]
        images_total = (sum(len(range(r, args.images or 10000, world)) for r in range(world)) if cfg == "cfg4" and not stub
                        else n_ranks * batch * steps_done)
        out = {
            "metric": "images/sec end-to-end (det+cls+rec) at 960x960" if cfg == "cfg2" else "images/sec end-to-end (det+cls+rec), mixed 640-1280 px",
            "value": images_total / elapsed,
            "unit": "images/sec",
            "n_gpus": n_ranks,
            "steps": steps_done,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / steps_done,
            "higher_is_better": True,
            "scaling": "weak" if cfg != "cfg4" else "strong",
            "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "f16 tensors and matrix products, f32 accumulation (exploration run: --precision)",
            "data": "stub (launcher/gather rehearsal, not a measurement)" if stub else "synthetic",
            "config": {"workload": workload,
                       "images_per_step_per_gpu": batch, "sharding": "image i -> rank i mod n_gpus, no data-path collective; "
                                                                     "results all-gathered as 64-byte records after the timed region"},
            "p50_step_ms": statistics.median(step_ms),
            "p50_ms_per_image": statistics.median(step_ms) / batch,
            "stage_ms_last_step": {"det": stage_ms[0], "cls": stage_ms[1], "rec": stage_ms[2]},
            "words_per_step": nwords,
        }
        out["resident"] = {"value": out["value"], "unit": "images/sec",
                           "what": "`value`: inputs already in HBM when the timed region starts (the contract of this line)"} if cfg != "cfg4" else None
        if stream_stats:
            out["stream"] = stream_stats
        if single_chain:
            out["single_chain"] = single_chain
        if host_in:
            out["host_input"] = host_in
        if in_flight:
            out["two_batches_in_flight"] = in_flight
        if two_workers:
            out["two_workers_per_gpu"] = two_workers
        if fp16:
            out["fp16"] = fp16
        if cfg5:
            out["cfg5"] = cfg5
        if args.precision != "fp32":
            out["precision_override"] = args.precision
        if gather:
            out["gather"] = gather
        if single_ms:
            out["single_image_latency_ms"] = {"p50": statistics.median(single_ms), "p90": sorted(single_ms)[int(len(single_ms) * 0.9)],
                                              "what": "one resident 960x960 image (32 lines) per call, det+cls+rec, 20 calls"}
        if kernel_timing:
            rep = rep_timed
            if rep and dom_key is not None:
                groups = kernel_groups(rep)
                r = groups.get(dom_key) or max(groups.values(), key=lambda g: g["ms"])
                name = group_label(dom_key, r)
                avg_ms = r["ms"] / max(1, r["count"])
                mpeak = FP32_MFMA_PEAK_TFLOPS if args.precision == "fp32" else FP16_MFMA_PEAK_TFLOPS  # (an exploration run in fp16: that mode's roofs)
                roof = roof_of(r["flops"], r["bytes"], r["ms"], mpeak)
                tflops, gbps, hbm_bound = roof["tflops"], roof["hbm_GBps_algorithmic"], roof["bound"] == "hbm"
                traffic, traffic_src = None, None
                for tf in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("pmc_traffic.json")), reverse=True):
                    # HBM bytes per launch from the rocprofv3 --pmc passes (tools/pmc_traffic.py), latest round first
                    pm = json.load(open(os.path.join(ROOT, "profiles", tf)))
                    # (only an entry that names THIS kernel group and whose traced duration agrees with the bench's: tools/pmc_traffic.py
                    # labels the trace's top (symbol, grid) group and records whether the two durations agree)
                    if pm.get("kernel") == name and pm.get("durations_agree_within_10pct", False):
                        traffic = pm["traffic_bytes_per_launch"]
                        traffic_src = "profiles/%s: rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, gfx950 corrections) of the same bench command on this kernel (%s), not this run" % (tf, pm.get("launch", ""))
                        break
                out["roofline"] = {"kernel": name, "measured_in": "single_chain", "bound": "hbm" if hbm_bound else "mfma",
                                   "achieved": gbps if hbm_bound else tflops, "peak": HBM_PEAK_GBS if hbm_bound else mpeak,
                                   "unit": "GB/s" if hbm_bound else "TFLOP/s",
                                   "frac": gbps / HBM_PEAK_GBS if hbm_bound else tflops / mpeak, "traffic": traffic, "traffic_source": traffic_src,
                                   "avg_launch_ms": avg_ms, "launches": r["count"], "launches_per_step": r["count"] / max(1, single_chain["steps"]),
                                   "dominant_by": "time per step summed over the launches of one kernel instantiation on one shape (the "
                                                  "grouping of a rocprofv3 kernel summary by symbol and grid)",
                                   "algorithmic_flops_per_launch": r["flops"] / max(1, r["count"]),
                                   "algorithmic_bytes_per_launch": r["bytes"] / max(1, r["count"]),
                                   "tflops": tflops, "hbm_GBps_algorithmic": gbps}
            if survey:   # per-kernel shares from the untimed survey pass (one step, every launch timed)
                tot = sum(v["ms"] for v in survey.values())
                sgroups = kernel_groups(survey)
                out["kernel_time_share_top5"] = {group_label(k, g): round(g["ms"] / tot, 4) for k, g in
                                                 sorted(sgroups.items(), key=lambda kv: -kv[1]["ms"])[:5]}
                out["network_kernel_ms_per_step"] = tot
                # ---- the step as a whole against the roofs (the line had no step-level figure)
                fl = sum(v["flops"] for v in survey.values())
                by = sum(v["bytes"] for v in survey.values())
                step = {"flops": fl, "algorithmic_bytes": by, "ms": out["ms_per_step"],
                        "what": "algorithmic FLOPs of every network launch of one step (survey pass) over the headline ms_per_step "
                                "(two chains); pre/post-processing kernels add time, no FLOPs"}
                step.update({k: v for k, v in roof_of(fl, by, out["ms_per_step"]).items() if k != "bound"})
                step["frac"] = step["frac_mfma"]
                if single_chain:
                    step["single_chain_ms"] = single_chain["ms_per_step"]
                    step["single_chain_frac_mfma"] = roof_of(fl, by, single_chain["ms_per_step"])["frac_mfma"]
                step["kernel_ms_sum"] = tot
                step["kernel_ms_sum_frac_mfma"] = roof_of(fl, by, tot)["frac_mfma"]
                nets = {}
                for nm in ("det", "cls", "rec"):
                    rows = [v for k, v in survey.items() if k.startswith(nm + ".")]
                    if rows:
                        f_, b_, m_ = sum(v["flops"] for v in rows), sum(v["bytes"] for v in rows), sum(v["ms"] for v in rows)
                        nets[nm] = dict(flops=f_, algorithmic_bytes=b_, kernel_ms=m_, launches=sum(v["count"] for v in rows), **roof_of(f_, b_, m_))
                step["networks"] = nets
                # the north star's "det conv stack": the detector's dense convolutions (matrix-core kernels)
                dense = [v for k, v in survey.items() if k.startswith("det.") and any(t in k for t in (".conv1x1_", ".conv3x3_", ".dwpw"))]
                if dense:
                    f_, b_, m_ = sum(v["flops"] for v in dense), sum(v["bytes"] for v in dense), sum(v["ms"] for v in dense)
                    dn = nets.get("det", {})
                    step["det_conv_stack"] = dict(flops=f_, algorithmic_bytes=b_, kernel_ms=m_, launches=len(dense), **roof_of(f_, b_, m_),
                                                  north_star_target_frac_mfma=0.60,
                                                  note="the detector as launched moves %.1f algorithmic FLOP per HBM byte in f32, on the f32 ridge "
                                                       "(%.1f FLOP/B): its thin high-resolution layers are bandwidth work, so the stack as a whole "
                                                       "cannot reach 0.60 of the f32 matrix peak in the reference's precision (SURVEY.md section "
                                                       "7/8d, DESIGN.md section 6); per-layer figures in the kernel table"
                                                       % (dn.get("flops", 0.0) / max(1.0, dn.get("algorithmic_bytes", 1.0)), RIDGE))
                small = [v for v in survey.values() if v["count"] and v["ms"] / v["count"] < 0.25]
                step["launches_below_250us"] = {"launches": sum(v["count"] for v in small), "ms": sum(v["ms"] for v in small)}
                # every launch against ITS binding roof: sum over launches of max(flops / matrix peak, bytes / HBM peak) over the
                # kernel time - an HBM-bound layer is not punished with a matrix-peak denominator
                roof_ms = sum(max(v["flops"] / (FP32_MFMA_PEAK_TFLOPS * 1e12), v["bytes"] / (HBM_PEAK_GBS * 1e9)) * 1e3 for v in survey.values())
                step["layer_roof_ms"] = roof_ms
                step["layer_roof_frac"] = roof_ms / tot if tot else 0.0
                out["roofline"] = dict(out.get("roofline") or {}, step=step)
                # the same figures as top-level scalars of `roofline` (the driver's summary keeps scalars)
                out["roofline"].update({"step_frac": step["frac"], "step_ms": step["ms"], "step_layer_roof_frac": step["layer_roof_frac"],
                                        "det_conv_stack_frac": step.get("det_conv_stack", {}).get("frac_mfma"),
                                        "small_launch_ms": step["launches_below_250us"]["ms"], "kernel_ms_sum": tot})
                if os.environ.get("OCR_BENCH_KERNEL_TABLE"):
                    with open(os.environ["OCR_BENCH_KERNEL_TABLE"], "w") as f:
                        f.write("# groups (kernel instantiation x shape), one step, single chain\n")
                        for k, g in sorted(sgroups.items(), key=lambda kv: -kv[1]["ms"]):
                            f.write("G %-58s ms/step %8.3f launches %3d  ms/launch %7.3f  TFLOP/s %7.2f  GB/s(alg) %8.1f\n" % (
                                group_label(k, g), g["ms"], g["count"], g["ms"] / max(1, g["count"]),
                                g["flops"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] else 0, g["bytes"] / (g["ms"] * 1e-3) / 1e9 if g["ms"] else 0))
                        f.write("# launches\n")
                        for k, v in sorted(survey.items(), key=lambda kv: -kv[1]["ms"]):
                            f.write("%-40s ms/step %8.3f launches/step %5.1f  TFLOP/s %7.2f  GB/s(alg) %8.1f\n" % (
                                k, v["ms"], v["count"],
                                v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] else 0,
                                v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] else 0))
        if n_ranks > 1:
            out["roofline"] = None
            out["roofline_note"] = "per-kernel and step rooflines are measured by the N = 1 run (one process owning one GPU); an N > 1 line reports rates only"
        if per_rank:
            per_rank["images_per_sec"] = [batch * steps_done / s_ if s_ > 0 else 0.0 for s_ in per_rank["seconds"]] if cfg != "cfg4" else None
            out["per_rank"] = per_rank
        if share_gpu:
            out["rehearsal"] = "OCR_BENCH_SHARE_GPU: %d ranks drive GPU 0 with the real pipeline, records over gloo - a rehearsal of the N > 1 path on a one-GPU lease, not a scaling point" % n_ranks
        out["host"] = {"host_cores_allowed": cores_mine, "input_generation_s": input_gen_s, "input_generation_workers": workers,
                       "placement": placement,
                       "cpu_s_per_step": host_cpu_s / max(1, steps_done), "cpu_cores_busy": host_cpu_s / max(elapsed, 1e-9),
                       "threads": len(os.listdir("/proc/self/task")) if os.path.isdir("/proc/self/task") else None,
                       "wait_modes": wait_modes,
                       "what": "cpu_s_per_step: user + system CPU seconds of rank 0's process (the library's chain threads included) per "
                               "timed step; cpu_cores_busy = that over the wall time; threads: tasks of the process at the end of the run; "
                               "wait_modes: the same steps with the host threads blocking on an event / spinning in hipStreamSynchronize"}
        if n_ranks == 1 and not args.no_cpu_baseline and not stub and cfg == "cfg2":
            out["cpu_baseline"] = cpu_baseline()
        emit_line(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if not stub:
        pipe.close()


if __name__ == "__main__":
    import faulthandler
    faulthandler.enable()   # a crash inside the HIP library or RCCL leaves a stack on stderr, not an empty line
    main()
