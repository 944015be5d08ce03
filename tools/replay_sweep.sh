for cfg in "48 2 1" "48 2 0" "48 3 1" "48 4 1" "48 3 0"; do
  set -- $cfg
  python bench.py --scale 0.1 --steps 3 --warmup 1 --no-e2e --no-device-leg --no-cpu-baseline --stream-buckets $1 --split-passes $2 --debug split_replay=$3 2>/dev/null \
   | python -c "import json,sys; o=json.loads(sys.stdin.read()); print('buckets $1 passes $2 replay $3: split %.1f ms, step %.1f ms' % (o['stage_ms']['split'], o['ms_per_step']))"
done
