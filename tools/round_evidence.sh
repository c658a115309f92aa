#!/bin/bash
# On the GPU box: the round's evidence, first call - full GPU suite, the three bench lines, kernel table, service load.
# (second call: tools/round_profiles.sh <tag> - rocprofv3 kernel trace + PMC passes, fp32 and fp16, and the micro-benchmarks;
# one gpurun call is limited to 20 minutes).  tools/round_evidence.sh <tag>  ->  gpurun_out/ev_<tag>/
T=${1:-r3}
O=gpurun_out/ev_$T
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
OCR_BENCH_KERNEL_TABLE=$O/kernel_table.txt python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err || exit 1
python bench.py --config cfg3 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err || exit 1
python bench.py --config cfg4 --images 2560 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err || exit 1
python tools/service_load.py jpeg 64 12 > $O/service_load.jsonl 2> $O/service_load.err
for c in cfg2 cfg3 cfg4; do python - $O/bench_$c.json <<'P'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1], round(d['value'],1), d.get('stage_ms_last_step'), (d.get('roofline') or {}).get('frac'))
P
done
