#!/bin/bash
# A/B builds: tools/ab_build.sh <tag> <source in csrc> [-Dmacro ...]  ->  tools/ab/libocr_hip_<tag>.so
# (the named source compiled with the extra macros, every other object taken from cpp-paddle-ocr_amd/build/: run
# cpp-paddle-ocr_amd/build.py first).  Use with OCR_LIB_PATH=tools/ab/libocr_hip_<tag>.so.
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
tag=$1; src=$2; shift 2
mkdir -p "$R/tools/ab"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-pass-failed "$@" -x hip -c "$R/cpp-paddle-ocr_amd/csrc/$src" -o "$R/tools/ab/$src.$tag.o"
objs=""
for o in "$R"/cpp-paddle-ocr_amd/build/*.o; do
  if [ "$(basename "$o")" = "$src.o" ]; then objs="$objs $R/tools/ab/$src.$tag.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/tools/ab/libocr_hip_$tag.so" $objs
echo "$R/tools/ab/libocr_hip_$tag.so"
