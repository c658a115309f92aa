#!/bin/bash
# Runs on the GPU box: kernel trace of a short bench run, then the per-step time of the det post-processing kernels.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/post_trace
rm -rf $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-latency --no-host-input --no-two-workers > $O.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$O/**/*_kernel_stats.csv", recursive=True)[0]
tot = 0
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if any(k in n for k in ("border_box", "trace_", "rotate180", "ccl_", "starts_", "bitmap", "dilate", "boxes_compact", "det_tail", "line_pre", "ctc_")):
        print("%-70s calls %4s avg %9.1f us total %8.3f ms" % (n[:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
