#!/bin/bash
# split time vs bucket count / passes at 1/10 of configs[2] (15 Gbases): how much of a split pass is
# the per-tile, per-bucket cursor traffic
for cfg in "1 1" "2 2" "8 2" "24 2" "48 2" "48 1" "96 2" "192 2"; do
  set -- $cfg
  python bench.py --scale 0.1 --steps 2 --warmup 1 --no-e2e --no-device-leg --no-cpu-baseline --stream-buckets $1 --split-passes $2 2>/dev/null \
   | python -c "import json,sys; o=json.loads(sys.stdin.read()); print('buckets $1 passes $2: split %.1f ms, step %.1f ms' % (o['stage_ms']['split'], o['ms_per_step']))"
done
