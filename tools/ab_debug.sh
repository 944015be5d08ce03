#!/bin/bash
# A/B of a fk_debug_set knob on one box: tools/ab_debug.sh <key> <value a> <value b> [reps]  (run on the GPU box)
key=$1; a=$2; b=$3; reps=${4:-2}
for r in $(seq $reps); do
  for v in $a $b; do
    python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-device-leg --no-e2e --debug $key=$v 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); s = d['stage_ms']
print('$key=$v', round(d['ms_per_step'], 1), 'count', s['count'], 'table_sort', s['table_sort'], 'sort_kmer', s['sort_kmer'], 'split', s['split'], 'sort_super', s['sort_super'])"
  done
done
