#!/bin/bash
# usage: tools/stage_ms.sh "<bench args>" knob=value ...   -- prints stage_ms of bench.py for each knob setting
for kv in "${@:2}"; do
  python bench.py $1 --no-cpu-baseline --debug $kv 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('$kv', round(d['ms_per_step'], 2), d['stage_ms'])
"
done
