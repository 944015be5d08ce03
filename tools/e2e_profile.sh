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
if [ -n "$E2E_PMC" ]; then
  # counters in runs of their own (kernel-trace only), the binary itself behind `--`
  sleep 12
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $out/e2e_sq -- fastk_amd/bin/FastK_amd -k40 -t4 -T32 -M256 -N$d/out $d/reads.fasta > $out/e2e_sq.log 2>&1
  q=$(find $out/e2e_sq -name "*.db" | head -1); python3 profiles/summarize_pmc.py "$q" $out/e2e_pmc_sq.csv > /dev/null 2>&1
  sleep 12
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/e2e_fetch -- fastk_amd/bin/FastK_amd -k40 -t4 -T32 -M256 -N$d/out $d/reads.fasta > $out/e2e_fetch.log 2>&1
  sleep 12
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/e2e_write -- fastk_amd/bin/FastK_amd -k40 -t4 -T32 -M256 -N$d/out $d/reads.fasta > $out/e2e_write.log 2>&1
  f=$(find $out/e2e_fetch -name "*.db" | head -1); w=$(find $out/e2e_write -name "*.db" | head -1)
  python3 profiles/e2e_traffic.py "$f" "$w" $out/e2e_pmc_traffic.json > $out/e2e_pmc_traffic.txt 2>&1
  find $out -name "*.db" -delete
fi
rm -rf "$d"
tail -6 $out/e2e_under_trace.log; head -14 $out/e2e_kernel_stats.csv | cut -c1-160
