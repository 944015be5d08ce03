#!/bin/bash
# N copies of tools/probe/pageable_copy_probe side by side on one GPU:  tools/pageable_probe_many.sh <processes> <seconds>
N=${1:-32}; SECS=${2:-60}
cd "$(dirname "$0")/probe"
[ -x pageable_copy_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o pageable_copy_probe pageable_copy_probe.cpp 2>/dev/null
pids=()
for i in $(seq 0 $((N - 1))); do ./pageable_copy_probe "$SECS" "$i" > /tmp/pcp_$i.txt 2>&1 & pids+=($!); done
for p in "${pids[@]}"; do wait "$p"; done
cat /tmp/pcp_*.txt | grep -v "iterations," | head -40
cat /tmp/pcp_*.txt | grep "iterations," | awk '{it += $3; d += $5; h += $(NF-5)} END {print "TOTAL: " it " iterations; D2H wrong " d "; H2D wrong " h}'
