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
run a3 new OCR_PRIO_ANCHOR=3
run a4 new OCR_PRIO_ANCHOR=4
run a5 new OCR_PRIO_ANCHOR=5
run a3l16 new OCR_PRIO_ANCHOR=3 OCR_DET_LANES=16
run a4l16 new OCR_PRIO_ANCHOR=4 OCR_DET_LANES=16
run a0l16 new OCR_PRIO_ANCHOR=0 OCR_DET_LANES=16
run a0l32 new OCR_PRIO_ANCHOR=0 OCR_DET_LANES=32
