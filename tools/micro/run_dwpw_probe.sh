#!/bin/bash
# knock-out table of the fused depthwise->1x1 kernel (build the variants first, see below); shapes: N H W K SH SW Cin Cout
#   for v in "full:" "nomma:-DOCR_DWPW_NO_MMA" "notaps:-DOCR_DWPW_NO_TAPS" "nommataps:-DOCR_DWPW_NO_MMA -DOCR_DWPW_NO_TAPS" \
#            "allmem:-DOCR_DWPW_NO_G -DOCR_PROBE_NOSTORE -DOCR_DWPW_NO_B" "none:-DOCR_DWPW_NO_G -DOCR_PROBE_NOSTORE -DOCR_DWPW_NO_B -DOCR_DWPW_NO_MMA -DOCR_DWPW_NO_TAPS"; do
#     hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../cpp-paddle-ocr_amd/csrc ${v#*:} -o dwpw_probe_${v%%:*} dwpw_probe.hip; done
cd "$(dirname "$0")"
for shape in "2048 12 160 3 1 1 128 128" "2048 24 160 3 1 1 64 64" "2048 12 80 5 1 1 240 240"; do
  echo "== $shape"
  for v in full nomma notaps nommataps allmem none; do printf "%-10s " $v; ./dwpw_probe_$v $shape | head -1; done
done
