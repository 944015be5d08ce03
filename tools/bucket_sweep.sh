# bucket count at full configs[2] scale (3 split passes)
for nb in 40 56 64; do
  python bench.py --steps 2 --warmup 1 --no-e2e --no-device-leg --no-cpu-baseline --stream-buckets $nb --split-passes 3 2>/dev/null \
   | python -c "import json,sys; o=json.loads(sys.stdin.read()); print('buckets $nb: step %.1f ms' % o['ms_per_step'], o['stage_ms'])"
done
