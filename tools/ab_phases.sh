#!/bin/bash
# Runs on the GPU box: throughput against the number of chains per pipeline, configs[1] and configs[2].
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; mkdir -p $O
for cfg in cfg2 cfg3; do
for p in 1 2 3 4; do
  OCR_PIPE_PHASES=$p python3 $R/bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-two-workers --no-latency --no-host-input > $O/${cfg}_p$p.json 2> $O/${cfg}_p$p.err
  python3 - <<PY
import json
try:
    d=json.loads(open("$O/${cfg}_p$p.json").read().strip().splitlines()[-1]); print("$cfg chains $p:", round(d["value"],1), d["stage_ms_last_step"])
except Exception as e: print("$cfg $p ERR", e)
PY
done
done
