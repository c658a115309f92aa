#!/bin/bash
# PMC passes over tools/micro/srv_gemm_probe on one shape and one tile configuration (PROBE_CFG): tools/pmc_gemm.sh "M K N res" cfg tag
R=${GRAFT_REPO_ROOT:-$(pwd)}
SHAPE=${1:-"983040 192 768 0"}; CFG=${2:-3}; TAG=${3:-gemm}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc_$TAG
rm -rf $O; mkdir -p $O
export PROBE_CFG=$CFG
run() { name=$1; shift; timeout 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -- $R/tools/micro/srv_gemm_probe $SHAPE 1 > $O/$name.log 2>&1; echo "$name rc=$?"; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_sum
run tcc2 TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            if "srv_gemm" not in row.get("Kernel_Name", "") and "pgemm" not in row.get("Kernel_Name", ""):
                continue
            a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
        for k, (v, n) in sorted(acc.items()):
            print("%-28s per dispatch %.4g (%d dispatches)" % (k, v / max(n, 1), n))
PY
