#!/bin/bash
# Where k_ag_count spends its time: rebuilds the library with -DFK_ABLATION on the GPU box (the snapshot there is
# scratch) and prints the per-phase cycle sums of thread 0 of every workgroup (FK_AG_TIMING, fk_aggr.hip AG_T).
make -C fastk_amd/csrc ABLATION=1 -B -j16 > gpurun_out/ag_build.log 2>&1 || { tail gpurun_out/ag_build.log; exit 1; }
for dbg in ${VARIANTS:-"x=0" "aggr_variant=2" "aggr_variant=4" "aggr_variant=1"}; do
  echo "== $dbg"
  d="--debug $dbg"; [ "$dbg" = "x=0" ] && d=""
  FK_AG_TIMING=1 python bench.py ${SCALE:---scale 0.1} --steps 1 --warmup 1 --no-e2e --no-device-leg --no-cpu-baseline $d 2> gpurun_out/ag_phases.err > gpurun_out/ag_phases.json
  python -c "import json,sys; o=json.loads(open('gpurun_out/ag_phases.json').read()); print('aggregate %.1f ms, step %.1f ms' % (o['stage_ms'].get('count', -1), o['ms_per_step'])); print(o['stage_ms'])"
  grep -A1 "ag phases" gpurun_out/ag_phases.err | tail -4
done
