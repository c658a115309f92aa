#!/bin/bash
# Runs on the GPU box (via gpurun): PMC passes (each its own run, --kernel-trace only) over ONE network
# forward of tools/net_bench.py, into gpurun_out/pmc_<tag>/<pass>.   usage: pmc_net.sh tag kind N H W
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
B="python3 $R/tools/net_bench.py $* 1"
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -- $B > $O/sq.log 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr TCP_TCC_READ_REQ_LATENCY_sum --kernel-trace --output-format csv -d $O/tcp -- $B > $O/tcp.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum TCC_BUSY_avr --kernel-trace --output-format csv -d $O/tcc -- $B > $O/tcc.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $O/inst -- $B > $O/inst.log 2>&1
find $O -name "*counter_collection.csv" | head
