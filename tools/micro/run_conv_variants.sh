#!/bin/bash
# tools/micro/conv_time on the recognizer's / detector's big 1x1 conv shapes: M cin cout gate nt mt
cd "$(dirname "$0")"
for cfg in "983040 480 480 1 3 1" "983040 480 480 1 3 2" "983040 480 480 0 3 1" "983040 480 480 0 3 2" "983040 240 480 1 3 1" "983040 240 480 1 3 2" \
           "1966080 240 240 0 4 1" "1966080 240 240 0 4 2" "230400 384 384 1 4 1" "230400 384 384 1 4 2" "983040 480 120 0 4 1" "983040 120 480 0 3 1"; do
  ./conv_time $cfg || exit 1
done
