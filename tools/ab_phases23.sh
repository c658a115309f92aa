#!/bin/bash
# chains per pipeline 2 vs 3, alternating in one box: configs[1], configs[2], configs[3]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; mkdir -p $O
for rep in 1 2; do
for p in 2 3; do
  for cfg in "cfg2" "cfg3" "cfg4 --images 2560"; do
    set -- $cfg
    OCR_PIPE_PHASES=$p python3 $R/bench.py --config $cfg --no-cpu-baseline --no-kernel-timing --no-two-workers --no-latency --no-host-input > $O/$1_p${p}_$rep.json 2> $O/$1_p${p}_$rep.err
    python3 - $O/$1_p${p}_$rep.json $1 $p <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], "chains", sys.argv[3], round(d["value"],1))
PY
  done
done
done
