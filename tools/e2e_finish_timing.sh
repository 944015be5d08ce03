#!/bin/bash
# Where FastK_amd's "count + table fetch" phase goes at configs[2] (or SCALE of it): FK_FINISH_TIMING prints the
# gather / count / table sort / preparation shares of fk_finish_device.   bash tools/e2e_finish_timing.sh
d=$(mktemp -d /dev/shm/fke2eft.XXXXXX)
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
fastk_amd/bin/FastK_amd -k40 -t4 -T32 -M256 -N$d/warm $d/reads.fasta > /dev/null 2>&1      # (the first run after the file was written)
for i in 1 2; do
  sleep 12
  FK_FINISH_TIMING=1 fastk_amd/bin/FastK_amd -v -k40 -t4 -T32 -M256 -N$d/out $d/reads.fasta 2>&1 | grep -E "finish timing|Wall s|Device ms|pieces packed"
done
rm -rf $d
