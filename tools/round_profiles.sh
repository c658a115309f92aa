#!/bin/bash
# On the GPU box: the round's evidence, second call (after tools/round_evidence.sh <tag>, which leaves bench_cfg2.json) - the
# rocprofv3 kernel trace + PMC passes of the fp32 bench command and of `bench.py --precision fp16`, their summaries, and the
# micro-benchmarks the round's DESIGN notes cite.  tools/round_profiles.sh <tag>  ->  gpurun_out/ev_<tag>/
T=${1:-r5}
O=gpurun_out/ev_$T
mkdir -p $O
# (gpurun_out/ does not travel to the box: this call makes its own bench line for the duration check of pmc_traffic.py)
python bench.py --no-cpu-baseline --no-fp16 --no-cfg5 --no-two-workers --no-host-input --no-latency > $O/bench_cfg2_prof.json 2> $O/bench_cfg2_prof.err
tools/run_profile.sh $T > $O/profile.log 2>&1
python tools/prof_summary.py gpurun_out/prof_$T > $O/prof_summary.txt 2>&1
python tools/pmc_traffic.py gpurun_out/prof_$T $O/pmc_traffic.json $O/bench_cfg2_prof.json dw_lds_kernel,dw_conv_kernel > $O/pmc_traffic.log 2>&1
python tools/hbm_table.py gpurun_out/prof_$T > $O/kernel_hbm_table.txt 2>&1
cp gpurun_out/prof_$T/trace/*/*kernel_stats.csv $O/ 2>/dev/null
# precision "fp16": its own bench line (every leg in fp16; the line's roofline is the mode's dominant kernel), then the same passes
python bench.py --precision fp16 --no-cpu-baseline --no-two-workers --no-host-input --no-latency > $O/bench_fp16.json 2> $O/bench_fp16.err
tools/run_profile.sh $T fp16 > $O/profile_fp16.log 2>&1
python tools/pmc_traffic.py gpurun_out/prof_${T}_fp16 $O/pmc_traffic_fp16.json $O/bench_fp16.json > $O/pmc_traffic_fp16.log 2>&1
python tools/hbm_table.py gpurun_out/prof_${T}_fp16 > $O/kernel_hbm_table_fp16.txt 2>&1
mkdir -p $O/fp16; cp gpurun_out/prof_${T}_fp16/trace/*/*kernel_stats.csv $O/fp16/ 2>/dev/null
python tools/fp16_check.py 8 > $O/fp16_check.txt 2>&1
# micro-benchmarks (binaries built in the container, tools/micro/): what the matrix pipe sustains on data, the vendor GEMM on the
# big 1x1 convs' shapes, the fused 5x5 block in both forms with its shader clock
# (the binaries are not tracked - build them here, and stop with a message rather than record "not found" lines as evidence)
make -s -C tools/micro all > $O/micro_build.log 2>&1 || { echo "tools/micro does not build: see $O/micro_build.log" | tee $O/micro.txt; exit 1; }
( cd tools/micro
  ./mfma_power 100
  python mm_ref.py
  for f in 1 0; do if [ $f = 1 ]; then export PROBE_FORM1=1; else unset PROBE_FORM1; fi
    echo "fused 5x5 240->240 block, form $((2 - f)) (PROBE_FORM1=$f): zeros, then seeded data, 200 launches each"
    PROBE_ITERS=200 ./dwpw_probe_clkrate 2048 12 80 5 1 1 240 240
    PROBE_ITERS=200 PROBE_RANDOM=1 ./dwpw_probe_clkrate 2048 12 80 5 1 1 240 240
  done
  unset PROBE_FORM1
  for a in "983040 480 480 1 3 2" "983040 480 480 0 3 2" "983040 240 480 1 3 2"; do CONV_TIME_ITERS=40 ./conv_time_clk $a; done ) > $O/micro.txt 2>&1
tail -30 $O/micro.txt
