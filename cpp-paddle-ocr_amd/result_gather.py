"""Multi-process side of the image-sharded pool: rank launcher, fixed-size result records, cross-rank gather.

The reference's GPUWorkerPool (/root/reference/src/gpu_worker_pool.cpp:12-16,37-59) is N workers in one
process handing results back through std::future.  One process per GPU replaces the workers here; the only
thing that crosses ranks is the RESULT of each image - a few hundred bytes - gathered after the work is done
(north star: "RCCL only for result gather").  Nothing on the data path is a collective.

Record layout (int32[16] per recognised word; 64 bytes, fixed size so that one all_gather moves them):
  [0] global image index   [1] word index inside the image   [2..9] box (4 points x,y)
  [10] confidence (f32 bits)   [11] number of CTC ids   [12] FNV-1a hash of the ids   [13..15] 0
"""
import os
import socket
import subprocess
import sys

import numpy as np

REC_INTS = 16


def fnv1a32(ids):
    h = 0x811C9DC5
    for v in np.asarray(ids, np.int64).ravel():
        for sh in (0, 8, 16, 24):
            h ^= (int(v) >> sh) & 0xFF
            h = (h * 0x01000193) & 0xFFFFFFFF
    return h if h < 0x80000000 else h - 0x100000000


def pack_records(words_per_image, image_ids, cap):
    """words_per_image: list (one per image) of lists of dict(box[4,2], ids, confidence) -> int32 [cap, 16] + count.
    Rows beyond the count are -1 in slot 0."""
    out = np.zeros((cap, REC_INTS), np.int32)
    out[:, 0] = -1
    k = 0
    for gi, words in zip(image_ids, words_per_image):
        for wi, w in enumerate(words):
            if k >= cap:
                raise ValueError("more words than the gather capacity (%d)" % cap)
            r = out[k]
            r[0], r[1] = gi, wi
            r[2:10] = np.asarray(w["box"], np.int32).reshape(8)
            r[10] = np.float32(w["confidence"]).view(np.int32)
            r[11] = len(w["ids"])
            r[12] = fnv1a32(w["ids"])
            k += 1
    return out, k


def gather_records(dist, records, device=None):
    """all_gather of every rank's [cap, 16] record block -> int32 [world, cap, 16] on every rank (numpy).
    `dist` is torch.distributed with an initialised group (nccl = RCCL over xGMI on the GPU box, gloo on CPU)."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(records))
    if device is not None:
        t = t.to(device)
    world = dist.get_world_size()
    bufs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(bufs, t)
    return np.stack([b.cpu().numpy() for b in bufs])


def records_by_image(block):
    """[cap, 16] -> {global image index: [record rows in word order]}"""
    out = {}
    for r in block:
        if r[0] < 0:
            continue
        out.setdefault(int(r[0]), []).append(r.copy())
    for v in out.values():
        v.sort(key=lambda r: int(r[1]))
    return out


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(script, argv, n, extra_env=None, timeout=None):
    """Starts `n` copies of `script argv` as ranks 0..n-1 of one job (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in
    the environment, as torch.distributed.run would) and waits for them.  Must be called from a process that has
    not touched the GPU: the children are fresh interpreters, nothing is exec'ed over an initialised runtime.
    Rank 0's stdout is this process's stdout.  Returns the largest exit code."""
    env0 = dict(os.environ)
    env0.update(extra_env or {})
    env0.setdefault("MASTER_ADDR", "127.0.0.1")
    env0.setdefault("MASTER_PORT", str(free_port()))
    env0["WORLD_SIZE"] = str(n)
    env0["LOCAL_WORLD_SIZE"] = str(n)
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # the library's default (set when it is loaded) comes too late in a rank: torch / RCCL initialise HIP first
    env0.setdefault("GPU_MAX_HW_QUEUES", "16")
    procs = []
    for r in range(n):
        env = dict(env0)
        env["RANK"] = env["LOCAL_RANK"] = str(r)
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    import time
    rc = 0
    t0 = time.time()
    try:
        live = list(procs)
        while live:
            for p in list(live):
                code = p.poll()
                if code is not None:
                    live.remove(p)
                    rc = max(rc, abs(code))
            if rc or (timeout and time.time() - t0 > timeout):
                rc = rc or 124
                break  # a rank failed: the others would wait for it in the next collective
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()  # exactly the children started here
    return rc
