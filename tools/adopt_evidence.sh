#!/bin/bash
# Copies the judged summaries of one tools/round_evidence.sh run into profiles/ (tracked): tools/adopt_evidence.sh <tag> [round]
T=${1:?tag}; R=${2:-r5}; O=gpurun_out/ev_$T
tail -3 $O/pytest.log > profiles/${R}_gpu_tests.txt
cp $O/bench_cfg2.json profiles/${R}_bench_line.json
cp $O/bench_cfg3.json profiles/${R}_bench_cfg3.json
cp $O/bench_cfg4.json profiles/${R}_bench_cfg4.json
cp $O/kernel_table.txt profiles/${R}_kernel_table.txt
cp "$(ls -t $O/*kernel_stats.csv | head -1)" profiles/${R}_rocprofv3_kernel_stats.csv
cp $O/pmc_traffic.json profiles/${R}_pmc_traffic.json
cp $O/kernel_hbm_table.txt profiles/${R}_kernel_hbm_table.txt
cp $O/prof_summary.txt profiles/${R}_prof_summary_bench_steps2.txt
python -c "import json,sys; json.dump([json.loads(l) for l in open(sys.argv[1]) if l.strip()], open(sys.argv[2], 'w'), indent=1)" $O/service_load.jsonl profiles/${R}_service_load.json
# (round 5) the fp16 passes, the micro-benchmarks and the cv_compat blast radius
[ -f $O/bench_fp16.json ] && cp $O/bench_fp16.json profiles/${R}_bench_fp16.json
[ -f $O/pmc_traffic_fp16.json ] && cp $O/pmc_traffic_fp16.json profiles/${R}_pmc_traffic_fp16.json
[ -f $O/kernel_hbm_table_fp16.txt ] && cp $O/kernel_hbm_table_fp16.txt profiles/${R}_kernel_hbm_table_fp16.txt
ls $O/fp16/*kernel_stats.csv > /dev/null 2>&1 && cp "$(ls -t $O/fp16/*kernel_stats.csv | head -1)" profiles/${R}_rocprofv3_kernel_stats_fp16.csv
[ -f $O/micro.txt ] && cp $O/micro.txt profiles/${R}_micro.txt
[ -f gpurun_out/r5_cv_compat_blast_radius.json ] && cp gpurun_out/r5_cv_compat_blast_radius.json profiles/${R}_cv_compat_blast_radius.json
[ -f $O/fp16_check.txt ] && cp $O/fp16_check.txt profiles/${R}_fp16_check.txt
