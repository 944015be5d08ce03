# k_ag_count at full configs[2] scale: bins per table fill and fill limit
for dbg in "aggr_gshift=1" "aggr_gshift=1 --debug aggr_limit=4500" "aggr_gshift=2" ; do
  python bench.py --steps 2 --warmup 1 --no-e2e --no-device-leg --no-cpu-baseline --debug $dbg 2>/dev/null \
   | python -c "import json,sys; o=json.loads(sys.stdin.read()); print('$dbg: step %.1f ms' % o['ms_per_step'], o['stage_ms'])"
done
