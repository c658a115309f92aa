#!/bin/bash
# Runs on the GPU box: configs[1] throughput against the number of chains per pipeline and parts per chain.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; mkdir -p $O
for cfg in "1 1" "2 1" "2 2" "2 4" "3 1" "3 2" "4 1" "4 2"; do
  set -- $cfg
  OCR_PIPE_PHASES=$1 OCR_PIPE_PARTS=$2 python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-two-workers --no-latency > $O/cfg2_p$1_m$2.json 2> $O/cfg2_p$1_m$2.err
  python3 - <<PY
import json
try:
    d=json.loads(open("$O/cfg2_p$1_m$2.json").read().strip().splitlines()[-1]); print("chains $1 parts/chain $2:", round(d["value"],1), "host", round(d.get("host_input",{}).get("value",0),1))
except Exception as e: print("$1 $2 ERR", e)
PY
done
