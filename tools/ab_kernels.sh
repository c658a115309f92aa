#!/bin/bash
# kernel tables of bench.py (one chain) for several builds: tools/ab_kernels.sh <outdir> <tag>...   ("base" = the in-tree library)
out=$1; shift
mkdir -p "$out"
for tag in "$@"; do
  if [ "$tag" = base ]; then unset OCR_LIB_PATH; else export OCR_LIB_PATH=$PWD/tools/ab/libocr_hip_$tag.so; fi
  OCR_BENCH_KERNEL_TABLE=$out/kt_$tag.txt python bench.py --no-cpu-baseline --no-latency --no-host-input --no-two-workers > $out/bench_$tag.json 2> $out/bench_$tag.err || exit 1
  python - "$out/bench_$tag.json" "$tag" <<'P'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d['value'],1), round(d['single_chain']['value'],1), d['stage_ms_last_step'])
P
done
