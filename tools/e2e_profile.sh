#!/bin/bash
# rocprofv3 kernel statistics of one end-to-end FastK_amd run (configs[2]-sized FASTA in /dev/shm unless SCALE is set);
# run on the GPU box:  bash tools/e2e_profile.sh <tag>   -> gpurun_out/prof_<tag>/e2e_kernel_stats.csv
tag=${1:-r03}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
d=$(mktemp -d /dev/shm/fke2eprof.XXXXXX)
python3 - "$d" "${SCALE:-1.0}" <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import bench, fastk_amd
d, scale = sys.argv[1], float(sys.argv[2])
glen = int(3000e6 * scale); L = 15000; nreads = int(50 * glen / L)
ctx = fastk_amd.Context(kmer=40)
bench.write_synth_file(ctx, os.path.join(d, "reads.fasta"), False, 20251001, glen, L, 2000, nreads)
ctx.close()
PY
sleep 12
fastk_amd/bin/FastK_amd -v -k40 -t4 -T32 -M256 -N$d/warm $d/reads.fasta > $out/e2e_warm.log 2>&1      # (the first run after the file was written)
sleep 12
export FASTK_AMD_ATEXIT=1        # (FastK_amd leaves with _exit: the profiler's exit handlers would not run)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/e2e -- fastk_amd/bin/FastK_amd -v -k40 -t4 -T32 -M256 -N$d/out $d/reads.fasta > $out/e2e_under_trace.log 2>&1
cp $(find $out/e2e -name "*kernel_stats.csv" | head -1) $out/e2e_kernel_stats.csv 2>/dev/null
find $out/e2e -name "*.db" -delete; find $out/e2e -name "*kernel_trace.csv" -delete
rm -rf "$d"
tail -6 $out/e2e_under_trace.log; head -14 $out/e2e_kernel_stats.csv | cut -c1-160
