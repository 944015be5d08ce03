# SQ counters of every kernel of one configs[2] step: bash tools/prof_sq.sh  -> gpurun_out/prof_sq/sq.csv
cd /tmp; export TMPDIR=/tmp; cd - >/dev/null
out=gpurun_out/prof_sq; rm -rf $out; mkdir -p $out
B="python3 bench.py --steps 1 --warmup 0 --no-e2e --no-device-leg --no-cpu-baseline ${BENCH_ARGS}"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $out/a -- $B > $out/line.json 2> $out/a.err
f=$(find $out/a -name "*.db" | head -1); python3 profiles/summarize_pmc.py "$f" $out/sq.csv > /dev/null 2>&1
find $out -name "*.db" -delete
python3 - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/prof_sq/sq.csv')):
    busy=float(r['SQ_BUSY_CYCLES'])/32.0
    if busy<=0: continue
    valu=float(r['SQ_ACTIVE_INST_VALU'])*4/(1024*busy)
    print("%-60s calls %4s busy %8.2f Mcyc/call  VALU util %4.0f%%  VALU %.3g LDS %.3g" % (r['kernel'][:60], r['calls'], busy/1e6/int(r['calls']), 100*valu, float(r['SQ_INSTS_VALU']), float(r['SQ_INSTS_LDS'])))
PY
