"""Timeline view of a rocprofv3 kernel trace: per HIP queue, when it was busy inside the last bench
step, and how much of the step each queue's kernels cover (union of intervals).  Answers "is the
small-launch lane the critical path of the rec stage?".

    python tools/lane_timeline.py <dir with *_kernel_trace.csv> [n_last_steps=1]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def union_ms(iv):
    iv = sorted(iv)
    tot, cur_a, cur_b = 0, None, None
    for a, b in iv:
        if cur_b is None or a > cur_b:
            if cur_b is not None:
                tot += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    if cur_b is not None:
        tot += cur_b - cur_a
    return tot / 1e6


def main():
    root = sys.argv[1]
    rows = []
    for fn in glob.glob(os.path.join(root, "**", "*_kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?"),
                         r["Kernel_Name"], int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])))
    rows.sort()
    # a step starts at each det_pre_kernel
    starts = [i for i, r in enumerate(rows) if "det_pre_kernel" in r[4]]
    if not starts:
        print("no det_pre_kernel in trace")
        return
    s = starts[-1]
    step = rows[s:]
    t0 = step[0][0]
    t1 = max(r[1] for r in step)
    print("last step: %d dispatches, %.3f ms from first start to last end" % (len(step), (t1 - t0) / 1e6))
    print("all queues busy (union) %.3f ms" % union_ms([(r[0], r[1]) for r in step]))
    per = defaultdict(list)
    for r in step:
        per[(r[2], r[3])].append(r)
    for key, rs in sorted(per.items(), key=lambda kv: kv[1][0][0]):
        a = min(r[0] for r in rs)
        b = max(r[1] for r in rs)
        print("queue %s stream %s: %4d dispatches, first start +%.3f ms, last end +%.3f ms, busy %.3f ms, sum %.3f ms" % (
            key[0], key[1], len(rs), (a - t0) / 1e6, (b - t0) / 1e6, union_ms([(r[0], r[1]) for r in rs]),
            sum(r[1] - r[0] for r in rs) / 1e6))
    # phase markers: first dispatch of selected kernels
    marks = ["det_pre_kernel", "det_tail_kernel", "ccl_rows_kernel", "border_box_kernel", "line_pre_kernel", "rotate180", "ctc_kernel"]
    for m in marks:
        hits = [r for r in step if m in r[4]]
        if hits:
            print("  %-20s first +%.3f ms  last end +%.3f ms  (%d)" % (m, (hits[0][0] - t0) / 1e6, (max(h[1] for h in hits) - t0) / 1e6, len(hits)))
    # top kernels per queue
    for key, rs in sorted(per.items(), key=lambda kv: kv[1][0][0]):
        agg = defaultdict(lambda: [0, 0])
        for r in rs:
            nm = r[4].replace("void ocr::", "").replace("ocr::", "")
            nm = nm[:nm.index("(")] if "(" in nm else nm
            agg[nm][0] += r[1] - r[0]
            agg[nm][1] += 1
        print("queue %s stream %s top kernels:" % key)
        for nm, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:10]:
            print("    %-44s %8.3f ms %5d" % (nm, t / 1e6, c))


if __name__ == "__main__":
    main()
