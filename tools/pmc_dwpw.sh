#!/bin/bash
# PMC passes over tools/micro/dwpw_probe_base for one shape (args after the script name), each bounded by its own timeout.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc_dwpw
rm -rf $O; mkdir -p $O
run() { name=$1; shift; timeout -k 5 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -- $R/tools/micro/dwpw_probe_base $ARGS > $O/$name.log 2>&1; echo "$name rc=$?"; }
ARGS="$*"
run ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run wait SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
run mem SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run misc SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_MFMA
python3 - <<PY
import csv, glob, collections
for d in ("ic","wait","mem","misc"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(float); n = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        for k in acc: print(d, k, "%.4g" % (acc[k] / max(1, n[k])), "per dispatch over", n[k])
PY
