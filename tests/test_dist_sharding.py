"""N>1 path on CPU: two gloo ranks shard the image stream like bench.py (image i -> rank i mod N, no
data-path collective) and agree on the max-over-ranks time."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seeds = bench.shard_seeds(rank, world, batch=8)
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    q.put((rank, seeds, float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    s0, s1 = got[0][1], got[1][1]
    assert set(s0).isdisjoint(s1) and sorted(s0 + s1) == list(range(16))   # a partition of the stream
    assert all(i % 2 == 0 for i in s0) and all(i % 2 == 1 for i in s1)
    assert got[0][2] == got[1][2] == 2.0                                     # max over ranks


def test_single_rank_is_the_whole_stream():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.shard_seeds(0, 1, batch=64) == list(range(64))


def _run_bench(args, env_extra=None, timeout=300):
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr


def test_bench_gpus_flag_starts_that_many_ranks_and_gathers_their_results():
    """`bench.py --gpus 2` with no launcher: the same spawn, sharding, max-over-ranks timing and record gather the GPU
    run uses, on two gloo ranks with the stub pipeline.  The line must report the ranks the
    process group saw, and every rank's neighbour check of the gathered records must pass."""
    rc, out, err = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--stub-pipeline"])
    assert rc == 0, err[-2000:]
    assert out["n_gpus"] == 2 and out["gather"]["ranks"] == 2 and out["gather"]["backend"] == "gloo"
    assert out["gather"]["matches_single_rank"] is True and out["gather"]["verified_images_per_rank"] == 4
    assert len(out["gather"]["records_per_rank"]) == 2 and all(8 <= c <= 24 for c in out["gather"]["records_per_rank"])
    assert out["value"] > 0 and out["scaling"] == "weak" and out["data"].startswith("stub")
    # round 4: what an N > 1 line says about each rank and about itself
    pr = out["per_rank"]
    assert all(len(pr[k]) == 2 for k in ("seconds", "input_generation_s", "host_cores_allowed", "pinned", "numa_node", "images_per_sec"))
    assert all(c >= 1 for c in pr["host_cores_allowed"]) and all(v > 0 for v in pr["images_per_sec"])
    assert min(pr["images_per_sec"]) * 2 >= out["value"] * 0.999        # whole-job rate = all images over the SLOWEST rank's time
    assert out["host"]["host_cores_allowed"] >= 1 and "placement" in out["host"]
    assert out["roofline"] is None and "N = 1" in out["roofline_note"]


def test_rank_placement_helpers_never_touch_the_gpu_and_split_the_cpus():
    """bench.py pins a rank's host threads next to its GPU before anything initialises HIP: the helpers read sysfs only,
    survive a machine without KFD, and two ranks that fall back to the even split get disjoint CPU sets."""
    import subprocess
    code = r"""
import os, sys, json
sys.path.insert(0, %r)
import bench
a0 = sorted(os.sched_getaffinity(0))
node = bench.gpu_numa_node(0)
info = bench.pin_rank_to_its_gpus_numa_node(int(sys.argv[1]), 2)
print(json.dumps({"node": node, "info": info, "now": sorted(os.sched_getaffinity(0)), "before": a0, "torch": "torch" in sys.modules}))
""" % ROOT
    res = []
    for r in (0, 1):
        p = subprocess.run([sys.executable, "-c", code, str(r)], capture_output=True, text=True, timeout=120)
        assert p.returncode == 0, p.stderr[-1000:]
        res.append(__import__("json").loads(p.stdout.strip().splitlines()[-1]))
    assert not res[0]["torch"]
    if len(res[0]["before"]) >= 4:
        assert res[0]["info"]["pinned"] and res[1]["info"]["pinned"]
        assert set(res[0]["now"]).isdisjoint(res[1]["now"]) and set(res[0]["now"]) <= set(res[0]["before"])
    assert bench_cpulist_ok()


def bench_cpulist_ok():
    sys.path.insert(0, ROOT)
    import bench
    return bench._cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and bench._cpulist("") == []


def test_bench_gather_check_catches_a_rank_that_reports_a_wrong_box():
    rc, out, err = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--stub-pipeline"], {"OCR_BENCH_STUB_CORRUPT": "1"})
    assert rc != 0 and out is not None and "error" in out and out["gather"]["matches_single_rank"] is False


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    rc, out, err = _run_bench(["--gpus", "4", "--stub-pipeline"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc != 0 and out is None and "WORLD_SIZE=1" in err


def test_result_records_roundtrip():
    import importlib.util
    import numpy as np
    spec = importlib.util.spec_from_file_location("rg", os.path.join(ROOT, "cpp-paddle-ocr_amd", "result_gather.py"))
    G = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(G)
    words = [[dict(box=np.arange(8).reshape(4, 2), ids=np.array([5, 9, 9, 2]), confidence=0.75)], [],
             [dict(box=-np.arange(8).reshape(4, 2), ids=np.array([], np.int32), confidence=0.0),
              dict(box=np.full((4, 2), 959), ids=np.array([6624]), confidence=1.0)]]
    recs, n = G.pack_records(words, [10, 12, 14], cap=8)
    assert n == 3 and recs.shape == (8, 16) and (recs[3:, 0] == -1).all()
    by = G.records_by_image(recs)
    assert sorted(by) == [10, 14] and len(by[14]) == 2 and by[14][1][1] == 1
    assert by[10][0][11] == 4 and by[10][0][10] == np.float32(0.75).view(np.int32)
    assert G.fnv1a32([5, 9, 9, 2]) != G.fnv1a32([5, 9, 2, 9])          # order-sensitive
    assert by[10][0][12] == G.fnv1a32(np.array([5, 9, 9, 2]))
    import pytest
    with pytest.raises(ValueError):
        G.pack_records(words, [0, 1, 2], cap=2)


def test_bench_kernel_groups_follow_the_rocprof_rule():
    """bench.py's dominant kernel = the (instantiation, shape) group with the most time: launches with the same descriptor,
    bound shape and algorithmic work per launch are one group (four ops of one symbol on one grid), the same descriptor at
    another resolution (other work per launch) is another; roof_of picks the roof by the ridge of the given peak."""
    sys.path.insert(0, ROOT)
    import bench
    rep = {
        "rec.13.dwpw5x5_240_240_s11@2048x48x~321": dict(ms=3.0, count=1, flops=251e9, bytes=3.79e9),
        "rec.15.dwpw5x5_240_240_s11@2048x48x~321": dict(ms=3.1, count=1, flops=251e9, bytes=3.79e9),
        "rec.30.conv1x1_480_480_gated@2048x48x~321": dict(ms=4.0, count=1, flops=454e9, bytes=3.79e9),
        "det.67.conv3x3_96_24@64x960x960": dict(ms=1.7, count=1, flops=152e9, bytes=1.77e9),
        "det.63.conv3x3_96_24@64x960x960": dict(ms=0.47, count=1, flops=38e9, bytes=0.44e9),   # same name, 120x120 level
        "det.72.conv3x3_96_24@64x960x960": dict(ms=1.75, count=1, flops=152e9, bytes=1.77e9),
    }
    g = bench.kernel_groups(rep)
    assert len(g) == 4
    key, top = max(g.items(), key=lambda kv: kv[1]["ms"])
    assert key[:2] == ("rec", "dwpw5x5_240_240_s11") and top["count"] == 2 and abs(top["ms"] - 6.1) < 1e-9 and sorted(top["ops"]) == ["13", "15"]
    assert "[ops 13,15]" in bench.group_label(key, top)
    det = {k: v for k, v in g.items() if k[0] == "det"}
    assert sorted(v["count"] for v in det.values()) == [1, 2]
    r = bench.roof_of(top["flops"], top["bytes"], top["ms"])
    assert r["bound"] == "mfma" and abs(r["tflops"] - 502e9 / 6.1e-3 / 1e12) < 1e-6 and abs(r["frac_mfma"] - r["tflops"] / 157.3) < 1e-12
    assert bench.roof_of(1e9, 1e9, 1.0)["bound"] == "hbm"                                    # 1 FLOP/B: far below the f32 ridge
    assert bench.roof_of(top["flops"], top["bytes"], top["ms"], bench.FP16_MFMA_PEAK_TFLOPS)["bound"] == "hbm"   # 66 FLOP/B < 312
