#!/bin/bash
# Runs on the GPU box: configs[2] against the number of detector lanes (post-processing concurrency) with ragged det.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; mkdir -p $O
for l in 4 8 16 32; do
  OCR_DET_LANES=$l OCR_DET_CHUNK_MP=32 python3 $R/bench.py --config cfg3 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-host-input > $O/cfg3_l$l.json 2> $O/cfg3_l$l.err
  python3 - <<PY
import json
try:
    d=json.loads(open("$O/cfg3_l$l.json").read().strip().splitlines()[-1]); print("lanes $l:", round(d["value"],1), {k:round(v,1) for k,v in d["stage_ms_last_step"].items()})
except Exception as e: print("$l ERR", e, open("$O/cfg3_l$l.err").read()[-300:])
PY
done
