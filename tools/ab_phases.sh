#!/bin/bash
# Runs on the GPU box: configs[1] / configs[2] throughput against the number of chains per pipeline.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; mkdir -p $O
for p in 1 2 3 4; do
  OCR_PIPE_PHASES=$p python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-two-workers --no-latency > $O/cfg2_p$p.json 2> $O/cfg2_p$p.err
  OCR_PIPE_PHASES=$p python3 $R/bench.py --config cfg3 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-host-input > $O/cfg3_p$p.json 2> $O/cfg3_p$p.err
done
python3 - <<PY
import json
for c in ("cfg2","cfg3"):
    for p in (1,2,3,4):
        try:
            d=json.loads(open("$O/%s_p%d.json"%(c,p)).read().strip().splitlines()[-1]); print(c, "phases", p, round(d["value"],1), {k:round(v,1) for k,v in d["stage_ms_last_step"].items()}, "host", round(d.get("host_input",{}).get("value",0),1))
        except Exception as e: print(c, p, "ERR", e)
PY
