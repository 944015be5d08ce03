#!/bin/bash
# N copies of tools/probe/runtime_churn_probe side by side on one GPU:  tools/churn_probe_many.sh <processes> <seconds> <mode>
N=${1:-32}; SECS=${2:-60}; MODE=${3:-0}
cd "$(dirname "$0")/probe"
[ -x runtime_churn_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o runtime_churn_probe runtime_churn_probe.cpp 2>/dev/null
pids=()
for i in $(seq 0 $((N - 1))); do ./runtime_churn_probe "$SECS" "$i" "$MODE" > /tmp/rcp_$i.txt 2>&1 & pids+=($!); done
for p in "${pids[@]}"; do wait "$p"; done
cat /tmp/rcp_*.txt | grep -v " rounds, " | head -40
cat /tmp/rcp_*.txt | grep " rounds, " | awk '{r += $5; b += $7} END {print "TOTAL mode '"$MODE"': " r " rounds; canary bytes changed " b}'
