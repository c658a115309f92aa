#!/bin/bash
# stall / TA / TCP counters of the big 1x1 conv kernel (conv_mfma_mt_kernel<3,2,*>) ALONE, on the recognizer's
# 480 -> 480 gated shape (tools/micro/conv_time; the program directly behind `rocprofv3 ... --`).  -> gpurun_out/pmc_conv1x1/summary.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
make -s -C $R/tools/micro conv_time || exit 1
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc_conv1x1
rm -rf $O; mkdir -p $O
SH="983040 480 480 1 3 2"
run() { name=$1; shift; CONV_TIME_ITERS=10 timeout 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -- $R/tools/micro/conv_time $SH > $O/$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VALU
run sq2 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVES
run ta TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE
run tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
python3 - <<PY > $O/summary.txt
import csv, glob, collections
print("conv_mfma_mt_kernel<3,2,gate> alone on 983040 x 480 -> 480 (tools/micro/conv_time $SH), counters per dispatch:")
for d in sorted(glob.glob("$O/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            if "conv_mfma" not in row.get("Kernel_Name", ""):
                continue
            a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
        for k, (v, n) in sorted(acc.items()):
            print("  %-40s %.5g   (%d dispatches)" % (k, v / max(n, 1), n))
PY
cat $O/summary.txt
