#!/bin/bash
# Runs on the GPU box: kernel-trace stats of the cfg3 bench under two builds of the library (A/B of a regression).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in old new; do
  O=$R/gpurun_out/r3f/$v
  rm -rf $O; mkdir -p $O
  if [ $v = old ]; then export OCR_LIB_PATH=$R/tools/ab/libocr_hip_old.so; else unset OCR_LIB_PATH; fi
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --config cfg3 --images 256 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-latency --no-host-input --no-two-workers > $O.log 2>&1 || exit 1
  f=$(find $O -name '*_kernel_stats.csv' | head -1)
  cp $f $R/gpurun_out/r3f/${v}_kernel_stats.csv
  rm -rf $O
done
