#!/bin/bash
# PMC passes over tools/micro/conv_probe (a handful of dispatches), each bounded by its own timeout.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc_probe
rm -rf $O; mkdir -p $O
run() { name=$1; shift; timeout 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -- $R/tools/micro/conv_probe > $O/$name.log 2>&1; echo "$name rc=$?"; }
run ta TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE
run tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_avr TCC_TAG_STALL_sum
run sq SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAVE_CYCLES SQ_BUSY_CYCLES
find $O -name "*counter_collection.csv" | wc -l
