#!/bin/bash
# rocprofv3 evidence for the default bench.py command (BASELINE configs[2]); run on the GPU box:
#   bash tools/collect_profiles.sh <tag>        -> gpurun_out/prof_<tag>/...
# Counters go in their own runs (kernel-trace only), one --pmc set per run, as the guide prescribes.
tag=${1:-r02}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
B="python3 bench.py --steps 2 --warmup 1 --no-e2e --no-device-leg --no-cpu-baseline ${BENCH_ARGS}"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $B > $out/bench_line_under_trace.json 2> $out/trace.err
B1="python3 bench.py --steps 1 --warmup 0 --no-e2e --no-device-leg --no-cpu-baseline --no-packed-leg ${BENCH_ARGS}"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fetch -- $B1 > $out/bench_line_fetch.json 2> $out/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/write -- $B1 > $out/bench_line_write.json 2> $out/write.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $out/sq -- $B1 > $out/bench_line_sq.json 2> $out/sq.err
find $out -name "*.db" -o -name "*stats*.csv" | head -20
f=$(find $out/fetch -name "*.db" | head -1); w=$(find $out/write -name "*.db" | head -1); q=$(find $out/sq -name "*.db" | head -1)
python3 profiles/pmc_traffic.py "$f" "$w" $out/bench_line_fetch.json $out/pmc_traffic.json > $out/pmc_traffic.txt 2>&1
python3 profiles/summarize_pmc.py "$q" $out/pmc_sq.csv > /dev/null 2>&1
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv 2>/dev/null
# the databases are large: keep the summaries only
find $out -name "*.db" -delete; find $out -name "*kernel_trace.csv" -delete
ls -la $out
