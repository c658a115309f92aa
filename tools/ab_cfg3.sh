#!/bin/bash
# Runs on the GPU box: cfg3 bench under two builds of the library and a few runtime switches (A/B of a regression).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; mkdir -p $O
run() {  # name, lib(old|new), extra env...
  name=$1; lib=$2; shift 2
  ( if [ $lib = old ]; then export OCR_LIB_PATH=$R/tools/ab/libocr_hip_old.so; fi
    env "$@" python3 $R/bench.py --config cfg3 --images 256 --steps 2 --warmup 1 --no-cpu-baseline --no-host-input --no-kernel-timing > $O/$name.json 2> $O/$name.err )
  python3 - <<PY
import json
try:
    d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1]); print("$name", round(d["value"],1), {k:round(v,1) for k,v in d["stage_ms_last_step"].items()})
except Exception as e: print("$name ERR", e)
PY
}
run new_x01 new OCR_DEBUG_EXTRA_STREAMS=0,1
run new_dq0 new DEBUG_HIP_DYNAMIC_QUEUES=0
run new_dq1 new DEBUG_HIP_DYNAMIC_QUEUES=1
run new_dp new OCR_DEBUG_DET_PRIO=1
run new_dp_l16 new OCR_DEBUG_DET_PRIO=1 OCR_DET_LANES=16
run new_x01_l16 new OCR_DEBUG_EXTRA_STREAMS=0,1 OCR_DET_LANES=16
run new_x01_l4 new OCR_DEBUG_EXTRA_STREAMS=0,1 OCR_DET_LANES=4
