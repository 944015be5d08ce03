#!/bin/bash
# kernel statistics of the exact splitter against the default one (tools/exact_split_probe.py under rocprofv3); GPU box:
#   tools/xs_profile.sh [gbases=1.0]   -> gpurun_out/xs_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/xsprof
rm -rf $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/exact_split_probe.py ${1:-1.0} > gpurun_out/xs.log 2>&1 < /dev/null
f=$(find $out -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then cp $f gpurun_out/xs_kernel_stats.csv; head -16 gpurun_out/xs_kernel_stats.csv | cut -c1-150; fi
grep hifi gpurun_out/xs.log
rm -rf $out
