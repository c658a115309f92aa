#!/bin/bash
# On the GPU box: rocprofv3 kernel trace (+ stats) and the FETCH / WRITE PMC passes of `bench.py --config cfg5` (BASELINE
# configs[4]: server det + SVTR-large rec, fp16, batch 32), one chain per pipeline so that a launch's duration is the kernel's.
# usage: tools/run_profile_cfg5.sh <tag>   ->  gpurun_out/prof_<tag>_cfg5/{trace,pmc_fetch,pmc_write} and a summary
TAG=${1:-r6}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export OCR_PIPE_PHASES=1
B="python3 $R/bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu-baseline"
O=$R/gpurun_out/prof_${TAG}_cfg5
rm -rf $O; mkdir -p $O
# the tile configurations are found once, un-profiled, and kept (OCR_SRV_TUNE_FILE): the profiled runs launch the networks only
export OCR_SRV_TUNE_FILE=$O/tune.txt
timeout -k 10 400 $B > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$? ($(wc -l < $O/tune.txt) tuned layers)"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B > $O/trace.log 2>&1; echo "trace rc=$?"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B > /dev/null 2>&1; echo "fetch rc=$?"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $B > /dev/null 2>&1; echo "write rc=$?"
python3 - <<PY
import csv, glob, collections, json
O = "$O"
st = glob.glob(O + "/trace/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(st[0]))) if st else []
out = {"kernel_stats_top": [{k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage")} for r in rows[:14]]}
def pmc(name, key):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(O + "/" + name + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == key:
                a = acc[r["Kernel_Name"][:120]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    return acc
fe, wr = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
top = rows[0]["Name"][:120] if rows else None
if top:
    f, w = fe.get(top, [0, 1]), wr.get(top, [0, 1])
    # FETCH_SIZE / WRITE_SIZE are in KB; gfx950: FETCH_SIZE counts 128-byte requests as 64 (MI355X_MICROARCH.md, HBM): doubled here
    out["dominant"] = {"kernel": top, "avg_ns": float(rows[0]["AverageNs"]), "fetch_bytes_per_launch_corrected": 2 * 1024 * f[0] / max(f[1], 1),
                       "write_bytes_per_launch": 1024 * w[0] / max(w[1], 1), "launches_fetch_pass": f[1], "launches_write_pass": w[1]}
    out["dominant"]["traffic_bytes_per_launch"] = out["dominant"]["fetch_bytes_per_launch_corrected"] + out["dominant"]["write_bytes_per_launch"]
# the bench line's own dominant (kernel, shape) group, matched to its rocprof kernel: fused MLP launches by their template
# argument, anything else by the only kernel whose average duration agrees within 10 %
try:
    bl = json.loads([l for l in open(O + "/bench_line.json") if l.startswith("{")][-1])
    bk, bms = bl["roofline"]["kernel"], bl["roofline"]["ms_per_launch"]
    import re
    m = re.search(r"\.mlp(?:_ln)?_(\d+)_", bk)
    lnv = "true" if ".mlp_ln_" in bk else "false"
    cands = [r for r in rows if ((("srv_mlp_kernel<%s>" % m.group(1)) in r["Name"] or ("srv_mlp_kernel<%s, %s>" % (m.group(1), lnv)) in r["Name"]) if m
                                 else abs(float(r["AverageNs"]) / 1e6 - bms) <= 0.1 * bms)]
    if len(cands) == 1:
        r = cands[0]; nm = r["Name"][:120]
        f, w = fe.get(nm, [0, 1]), wr.get(nm, [0, 1])
        out["by_bench_kernel"] = {bk: {"rocprof_kernel": nm, "avg_ns": float(r["AverageNs"]), "bench_ms_per_launch": bms,
                                       "fetch_bytes_per_launch_corrected": 2 * 1024 * f[0] / max(f[1], 1), "write_bytes_per_launch": 1024 * w[0] / max(w[1], 1),
                                       "traffic_bytes_per_launch": 2 * 1024 * f[0] / max(f[1], 1) + 1024 * w[0] / max(w[1], 1)}}
except Exception as e:
    out["by_bench_kernel_error"] = repr(e)
json.dump(out, open(O + "/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
