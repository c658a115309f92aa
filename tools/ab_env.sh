#!/bin/bash
# kernel tables of bench.py (one chain) under several settings of ONE environment switch: tools/ab_env.sh <outdir> <VAR> <value>...
out=$1; var=$2; shift 2
mkdir -p "$out"
for v in "$@"; do
  env $var=$v OCR_BENCH_KERNEL_TABLE=$out/kt_$v.txt python bench.py --no-cpu-baseline --no-latency --no-host-input --no-two-workers --no-fp16 > $out/bench_$v.json 2> $out/bench_$v.err || exit 1
  python - "$out/bench_$v.json" "$var=$v" <<'P'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d['value'],1), round(d['single_chain']['value'],1), d['stage_ms_last_step'])
P
done
