#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + three separate PMC passes of the bench
# command, into gpurun_out/prof_<tag>/{trace,pmc_sq,pmc_fetch,pmc_write}.  PMC passes never share a
# run with each other's TCC counters (FETCH_SIZE 3 slots + WRITE_SIZE 2 slots > 4).
# usage: tools/run_profile.sh <tag> [fp16]   (fp16: the same passes over `bench.py --precision fp16`, into prof_<tag>_fp16)
TAG=${1:-r1}
PREC=""
if [ "$2" = "fp16" ]; then PREC="--precision fp16"; TAG=${TAG}_fp16; fi
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
# one chain per pipeline: the per-kernel durations and counters are then properties of the kernel (bench.py's roofline
# leg runs in the same mode)
export OCR_PIPE_PHASES=1
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-latency --no-two-workers --no-host-input --no-fp16 $PREC"
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B > $O.trace.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -- $B > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $B > /dev/null 2>&1
# issue-slot accounting of the dominant kernel (round 4): instructions by class and the cycles the SQ spent issuing them
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_issue -- $B > /dev/null 2>&1
ls $O
