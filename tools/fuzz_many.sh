#!/bin/bash
# N processes of tests/fuzz_parity.py side by side on one GPU (its iterations are small: launch latency and the CPU oracle
# bound them, not the device), each with its own seed, the reads of every iteration in a guarded read-only mapping,
# glibc's allocator filling what it hands out and takes back (MALLOC_PERTURB_) and checking its chunks (MALLOC_CHECK_).
#   [FUZZ_ENV="NAME=value ..."] tools/fuzz_many.sh <processes> <iterations each> <first seed> <seconds> [outdir]
# Prints one line per process and the total; exit code 1 when any process failed.
N=${1:-16}; IT=${2:-1000}; SEED=${3:-20261100}; SECS=${4:-1500}; OUT=${5:-gpurun_out/fuzz_many}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
pids=()
for i in $(seq 0 $((N - 1))); do
  MALLOC_PERTURB_=$((165 + i % 64)) MALLOC_CHECK_=3 PYTHONFAULTHANDLER=1 env ${FUZZ_ENV} timeout "$SECS" \
    python tests/fuzz_parity.py "$IT" $((SEED + i)) > "$OUT/fuzz_$((SEED + i)).log" 2>&1 &
  pids+=($!)
done
fail=0
for p in "${pids[@]}"; do wait "$p" || fail=1; done
total=0
for i in $(seq 0 $((N - 1))); do
  f="$OUT/fuzz_$((SEED + i)).log"
  last=$(grep -o "^iteration [0-9]* ok" "$f" | tail -1 | awk '{print $2}')
  if grep -q "^all .* iterations equal" "$f"; then last=$IT; fi
  echo "seed $((SEED + i)): ${last:-0} iterations; $(tail -1 "$f" | cut -c1-200)"
  total=$((total + ${last:-0}))
done
echo "TOTAL iterations clean: $total; any process failed or timed out: $fail"
grep -l "FAILED\|Fatal Python\|Segmentation\|CHANGED" "$OUT"/fuzz_*.log 2>/dev/null | sed 's/^/  see /'
exit $fail
